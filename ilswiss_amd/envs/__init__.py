"""Vectorised environments on the device: `vecenv.HipVectorEnv` / `get_envs` (the BaseVectorEnv protocol over libilsx's batched
planar stepper) and `models` (the articulated-body descriptions of Hopper, Walker2d and HalfCheetah), `envpool` (the reference's EnvpoolEnv surface)
and `terminals` (batched terminal predicates)."""
from .vecenv import HipVectorEnv, MinmaxEnv, ProxyEnv, ScaledEnv, get_envs  # noqa: F401
from .envpool import EnvpoolEnv, HipEnvPool  # noqa: F401,E402
from .terminals import get_terminal_func  # noqa: F401,E402
