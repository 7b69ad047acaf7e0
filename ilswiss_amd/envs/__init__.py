"""Vectorised environments on the device: `vecenv.HipVectorEnv` / `get_envs` (the BaseVectorEnv protocol over libilsx's batched
planar stepper) and `models` (the articulated-body descriptions of Hopper and Walker2d)."""
from .vecenv import HipVectorEnv, MinmaxEnv, ProxyEnv, ScaledEnv, get_envs  # noqa: F401
