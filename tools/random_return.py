"""Random-policy known answers of the device steppers (uniform[-1,1] actions through the fused rollout):  python tools/random_return.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilswiss_amd as ia
from ilswiss_amd.envs.vecenv import HipVectorEnv
ctx = ia.Context(0, seed=0)
for name, n, steps in (("hopper", 4096, 300), ("walker", 4096, 300), ("halfcheetah", 1024, 1000), ("ant", 1024, 1000), ("humanoid", 1024, 300)):
    env = HipVectorEnv(name, n, seed=1, ctx=ctx)
    for t in range(steps):
        env.rollout_step(policy=None, replay=None, max_path_length=1000, random_actions=True)
    ep, ret = env.rollout_stats()
    print(f"{name}: {int(ep)} episodes, mean return {ret / max(ep, 1):.2f}, mean length {n * steps / max(ep, 1):.1f}")
    env.close()
