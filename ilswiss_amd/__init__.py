"""ilswiss_amd — MI355X-native engine for the ILSwiss hot path (vec-env rollout -> HBM replay ->
SAC/TD3/PPO actor-critic update -> adv-IRL discriminator step).

All arithmetic is hand-written HIP for gfx950 in libilsx.so (ilswiss_amd/csrc, C ABI in include/ilsx.h);
the Python classes below keep the reference's Trainer / ReplayBuffer / policy names and only move data.
Importing the classes is cheap; touching a GPU object without the built library raises.
"""
from .device import Context, DevArray, get_context, set_gpu_mode  # noqa: F401
from .networks import (FlattenMlp, MakeDeterministic, Mlp,  # noqa: F401
                       ReparamTanhMultivariateGaussianPolicy)
from .replay import EnvReplayBuffer, SimpleReplayBuffer  # noqa: F401
from .sac import SoftActorCritic, SoftActorCriticGroup, Trainer  # noqa: F401
