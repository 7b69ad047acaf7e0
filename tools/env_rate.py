"""Env-stepper throughput (fused rollout_step with random actions), per model.  ENV_RATE_N = number of envs (default 4096)."""
import os, sys, time
sys.path.insert(0, ".")
import ilswiss_amd as ia
from ilswiss_amd.envs.vecenv import HipVectorEnv
ctx = ia.Context()
for name in sys.argv[1:] or ["hopper", "walker", "halfcheetah"]:
    n = int(os.environ.get("ENV_RATE_N", "4096"))
    env = HipVectorEnv(name, n, seed=1, ctx=ctx)
    rb = ia.SimpleReplayBuffer(64 * n, env.obs_dim, env.act_dim, ctx=ctx)
    env.reset()
    for _ in range(20):
        env.rollout_step(replay=rb, random_actions=True, max_path_length=1000)
    ctx.sync()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        env.rollout_step(replay=rb, random_actions=True, max_path_length=1000)
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"{name}: {n * K / dt / 1e6:.2f} M env-steps/s ({dt / K * 1e6:.0f} us per vec step)")
    env.close()
