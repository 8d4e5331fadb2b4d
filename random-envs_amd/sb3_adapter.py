"""stable-baselines3 ``VecEnv`` adapter (README.md:68: the reference's downstream, sb3-gym-interface, drives envs through
``reset() -> obs``, ``step_async(actions)``, ``step_wait() -> (obs, rewards, dones, infos)`` with numpy arrays, auto-reset and
``infos[i]["terminal_observation"]`` -- the semantics VecRandomEnv already has on device).

When ``stable_baselines3`` is importable the adapter IS a ``stable_baselines3.common.vec_env.VecEnv`` (real callers check
``isinstance``), and its spaces are real gymnasium / gym spaces (vec_env.make_spaces); otherwise it is the same class over
``object`` -- neither package is a dependency of the hot path."""
import numpy as np


def _vecenv_base():
    try:
        from stable_baselines3.common.vec_env import VecEnv
        return VecEnv
    except Exception:
        return object


class _AdapterBody:
    """Everything the adapter does; mixed with the base class chosen at import time (adapter_class)."""

    def __init__(self, env):
        self.env = env
        base = _vecenv_base()
        if base is not object:
            base.__init__(self, env.batch, env.observation_space, env.action_space)
        else:
            self.num_envs = env.batch
            self.observation_space = env.observation_space
            self.action_space = env.action_space
        self._actions = None

    def reset(self):
        return self.env.reset().cpu().numpy()

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        obs, rew, done, info = self.env.step(np.asarray(self._actions))
        d = done.cpu().numpy()
        trunc = info["TimeLimit.truncated"].cpu().numpy()
        infos = [{} for _ in range(self.num_envs)]
        if d.any():
            term = info["terminal_observation"].cpu().numpy()
            for i in np.nonzero(d)[0]:
                infos[i] = {"terminal_observation": term[i], "TimeLimit.truncated": bool(trunc[i])}
        return obs.cpu().numpy(), rew.cpu().numpy(), d, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def seed(self, seed=None):
        return self.env.seed(seed)

    def _n(self, indices):
        if indices is None:
            return self.num_envs
        return 1 if isinstance(indices, int) else len(list(indices))

    def env_method(self, method_name, *args, indices=None, **kwargs):
        """SB3 expects one result per (selected) env; the batch is ONE object, so the call runs once and its result
        is repeated.  Batched results ([num_envs, ...] tensors) are split per env instead."""
        out = getattr(self.env, method_name)(*args, **kwargs)
        n = self._n(indices)
        if hasattr(out, "shape") and len(out.shape) >= 1 and out.shape[0] == self.num_envs:
            idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
            return [out[i] for i in idx]
        return [out] * n

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)] * self._n(indices)

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self._n(indices)

    def get_images(self):
        return [None] * self.num_envs

    def export_lane(self, k=0):
        """Host snapshot (qpos, qvel, xi) of env k for an external viewer (rendering stays off the GPU path)."""
        return self.env.export_lane(k)


def adapter_class():
    """The adapter class over ``stable_baselines3.common.vec_env.VecEnv`` when that is importable, over ``object`` otherwise."""
    base = _vecenv_base()
    return type("SB3VecEnvAdapter", (_AdapterBody,) if base is object else (_AdapterBody, base), {"__doc__": _AdapterBody.__doc__})


SB3VecEnvAdapter = adapter_class()
