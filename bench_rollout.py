"""Rollout leg of bench.py: one vec-env step for N_ENV parallel Hopper envs, entirely on the device:
policy inference (ReparamTanhMultivariateGaussianPolicy.get_actions, policies.py:245-246) -> planar-engine
physics (4 RK4 substeps) -> reward/termination -> transition record into the HBM replay ring -> auto-reset
(base_algorithm.py:183-263).  One ilsx_rollout_step call = 2 kernel launches."""
from ilswiss_amd.envs.vecenv import HipVectorEnv


class Rollout:
    def __init__(self, ctx, policy, replay, n_env, seed=0):
        self.env = HipVectorEnv("hopper", n_env, seed=seed, ctx=ctx)
        self.pol, self.rb = policy, replay

    def describe(self):
        return ("Hopper-v2 model on the HIP planar articulated-body stepper (k_envg_step: 16 lanes per env, fp64, RK4, frame_skip 4, "
                "soft contacts + joint limits via PGS), fused replay insert, auto-reset, max_path_length 1000")

    def vec_step(self):
        self.env.rollout_step(policy=self.pol, replay=self.rb, max_path_length=1000)
