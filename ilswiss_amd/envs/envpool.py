"""The envpool-shaped surface of the reference (rlkit/envs/envpool.py:4-33, `use_envpool: true` in
exp_specs/sac/sac_hopper_envpool.yaml:49-57) over the HIP stepper.

`HipEnvPool` plays the part of the object `envpool.make(name, env_type=..., num_envs=..., seed=...)` returns: `step(actions,
env_id)` answers with `info` as a dict of arrays (`env_id`, `elapsed_step`), which `EnvpoolEnv.step` turns into the list of
dicts the training loop wants — the adapter's whole job in the reference.  Everything else (`reset(id)`, spaces, `__len__`, the
fused `rollout_step`) is HipVectorEnv's, reached through `__getattr__` exactly as the reference forwards to the pool."""
import numpy as np

from .vecenv import HipVectorEnv

# every MuJoCo task the steppers model: planar (csrc/env2d_group.h) and 3-D (csrc/env3d_wave.h)
ENVPOOL_NAMES = {"Hopper": "hopper", "Walker2d": "walker2d", "HalfCheetah": "halfcheetah", "Ant": "ant", "Humanoid": "humanoid"}


def _model_name(envpool_name):
    base = envpool_name.split("-")[0]
    if base not in ENVPOOL_NAMES:
        raise KeyError(f"envpool task {envpool_name!r}: the HIP stepper has {sorted(ENVPOOL_NAMES)} (-v2/-v3/-v4 all map to the same model)")
    return ENVPOOL_NAMES[base]


class HipEnvPool(HipVectorEnv):
    def __init__(self, task_id, env_type="gym", num_envs=1, seed=0, ctx=None, **kwargs):
        if env_type != "gym":
            raise NotImplementedError(f"env_type={env_type!r}: only the gym API shape is provided")
        super().__init__(_model_name(task_id), num_envs, seed=seed, ctx=ctx, **kwargs)
        self.task_id = task_id
        self._elapsed = np.zeros(self.env_num, np.int32)

    def reset(self, id=None):
        self._elapsed[slice(None) if id is None else np.atleast_1d(id)] = 0
        return super().reset(id)

    def step(self, actions, env_id=None):
        obs, rew, done, infos = super().step(actions, env_id)
        ids = np.array([i["env_id"] for i in infos], np.int32)
        self._elapsed[ids] += 1
        return obs, rew, done, dict(env_id=ids, elapsed_step=self._elapsed[ids].copy(), players=dict(env_id=ids))


class EnvpoolEnv:
    def __init__(self, env_specs, ctx=None, **kwargs):
        self._envs = HipEnvPool(env_specs["envpool_name"], env_type=env_specs.get("env_type", "gym"), num_envs=env_specs["env_num"],
                                seed=env_specs["training_env_seed"], ctx=ctx, **kwargs)

    @property
    def envs(self):
        return self._envs

    def step(self, actions, *args, **kwargs):
        obs, rew, done, info_dict = self.envs.step(actions, *args, **kwargs)
        # dict of arrays -> list of dicts; the nested "players" entry is dropped (envpool.py:19-27)
        info_list = [{k: v[idx] for k, v in info_dict.items() if k != "players"} for idx in range(len(obs))]
        return obs, rew, done, info_list

    def __getattr__(self, attrname):
        return getattr(self.envs, attrname)

    def __len__(self):
        return len(self.envs)
