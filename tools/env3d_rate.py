"""The 3-D stepper alone (fused rollout_step with random actions, no policy): env-steps/s at a given width — the workload of
tools/pmc_env3d.sh's counter passes (k_env3dw_step<23> = Humanoid-v2, <14> = Ant-v2).
    python tools/env3d_rate.py [humanoid|ant] [n_env] [vec_steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia  # noqa: E402
from ilswiss_amd.envs.vecenv import HipVectorEnv  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ctx = ia.Context(0, seed=0)
env = HipVectorEnv(name, n, seed=1, ctx=ctx)
rb = ia.SimpleReplayBuffer(64 * n, env.obs_dim, env.act_dim, ctx=ctx)
env.reset()
for _ in range(5):
    env.rollout_step(replay=rb, random_actions=True, max_path_length=1000)
ctx.sync()
t0 = time.perf_counter()
for _ in range(K):
    env.rollout_step(replay=rb, random_actions=True, max_path_length=1000)
ctx.sync()
dt = time.perf_counter() - t0
print(f"{name}: {n} envs, {n * K / dt / 1e6:.3f} M env-steps/s ({dt / K * 1e6:.0f} us per vec step)")
