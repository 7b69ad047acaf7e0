"""Learning-curve probe for the on-device PPO loop (exploration returns per iteration)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
from ilswiss_amd.envs.vecenv import HipVectorEnv
from ilswiss_amd.ppo import PPO, ReparamMultivariateGaussianPolicy
from ilswiss_amd.samplers import VecPathSampler, get_average_returns

n_env, T, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mb, boot = int(sys.argv[4]), int(sys.argv[5])
np.random.seed(0)
ctx = ia.Context(0, seed=0)
env = HipVectorEnv("hopper", n_env, seed=0, ctx=ctx, norm_obs=True)
ev = HipVectorEnv("hopper", 16, seed=77, ctx=ctx, norm_obs=True, obs_rms=env.obs_rms, update_obs_rms=False)
pol = ReparamMultivariateGaussianPolicy([64, 64], 11, 3, conditioned_std=False, hidden_activation="tanh", ctx=ctx)
vf = ia.FlattenMlp([64, 64], 1, 11, hidden_activation="tanh", ctx=ctx)
tr = PPO(pol, vf, mini_batch_size=mb, update_epoch=10, gae_tau=0.95, max_samples=n_env * T)
sampler = VecPathSampler(ev, ia.MakeDeterministic(pol), num_steps=2000, max_path_length=1000)
env.rollout_stats(reset=True)
t0 = time.time()
for it in range(iters):
    tr.eval_statistics = None
    tr.train_from_rollout(env, T, max_path_length=1000, bootstrap=bool(boot))
    ep, rs = env.rollout_stats(reset=True)
    line = f"it {it:3d} samples {(it+1)*n_env*T:9d} expl_eps {ep:7.0f} expl_ret {rs/max(ep,1):8.2f} segs {tr.eval_statistics['PPO Segments']:.0f}"
    if it % 5 == 4:
        ev.sync_obs_rms()
        line += f"  eval_det {get_average_returns(sampler.obtain_samples()):8.2f}  log_std {tr.get_flat_params(0)[-3:]}"
    print(line, f"t={time.time()-t0:.1f}s", flush=True)
