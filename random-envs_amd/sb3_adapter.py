"""Minimal stable-baselines3 ``VecEnv``-protocol adapter (duck-typed: stable-baselines3 itself is not a
dependency).  The reference's downstream (README.md:68, sb3-gym-interface) drives envs through
``reset() -> obs``, ``step_async(actions)``, ``step_wait() -> (obs, rewards, dones, infos)`` with numpy arrays
and auto-reset + ``infos[i]["terminal_observation"]`` -- the semantics VecRandomEnv already has on device."""
import numpy as np


class SB3VecEnvAdapter:
    def __init__(self, env):
        self.env = env
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

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self.env, method_name)(*args, **kwargs)]

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)
