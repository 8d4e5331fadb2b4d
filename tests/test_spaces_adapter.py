"""Spaces and the VecEnv adapter pick up the REAL gymnasium / gym / stable-baselines3 classes when those are importable
(jinja_mujoco_env.py:23-36,99-103; README.md:68).  None of them is installed in this image, so minimal fake modules pin
the branch; without them the duck-typed stand-ins are used."""
import sys
import types

import numpy as np


class _Dims:
    obs_dim, act_dim, discrete_action, act_low, act_high = 11, 3, 0, -1.0, 1.0


def _fake_gymnasium():
    gymn = types.ModuleType("gymnasium"); spaces = types.ModuleType("gymnasium.spaces")

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.full(shape, low, dtype=dtype); self.high = np.full(shape, high, dtype=dtype)
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    class Discrete:
        def __init__(self, n):
            self.n = n
    spaces.Box, spaces.Discrete = Box, Discrete
    gymn.spaces = spaces
    return gymn, spaces


def test_stand_ins_without_gym(monkeypatch):
    from random_envs_amd import vec_env
    for name in ("gymnasium", "gymnasium.spaces", "gym", "gym.spaces"):
        monkeypatch.setitem(sys.modules, name, None)        # import raises ImportError
    assert vec_env.spaces_module() is None
    obs, act = vec_env.make_spaces(_Dims)
    assert isinstance(obs, vec_env._Box) and isinstance(act, vec_env._Box)
    assert obs.shape == (11,) and obs.dtype == np.float32 and np.isinf(obs.low).all()
    assert act.shape == (3,) and act.low.min() == -1.0 and act.high.max() == 1.0 and act.contains(act.sample())


def test_real_spaces_when_gymnasium_is_importable(monkeypatch):
    from random_envs_amd import vec_env
    gymn, spaces = _fake_gymnasium()
    monkeypatch.setitem(sys.modules, "gymnasium", gymn); monkeypatch.setitem(sys.modules, "gymnasium.spaces", spaces)
    assert vec_env.spaces_module() is spaces
    obs, act = vec_env.make_spaces(_Dims)
    assert isinstance(obs, spaces.Box) and isinstance(act, spaces.Box)          # what SB3's isinstance checks see
    assert obs.dtype == np.float32 and obs.shape == (11,) and act.low[0] == np.float32(-1.0) and act.high[0] == np.float32(1.0)

    class Cart(_Dims):
        obs_dim, act_dim, discrete_action, act_low, act_high = 4, 1, 1, 0.0, 1.0
    obs, act = vec_env.make_spaces(Cart)
    assert isinstance(act, spaces.Discrete) and act.n == 2                        # random_cartpole.py:96


def test_adapter_subclasses_sb3_vecenv_when_importable(monkeypatch):
    from random_envs_amd import sb3_adapter
    assert sb3_adapter.adapter_class().__mro__[-1] is object and len(sb3_adapter.adapter_class().__mro__) == 3   # no SB3 here
    sb3 = types.ModuleType("stable_baselines3"); common = types.ModuleType("stable_baselines3.common")
    ve = types.ModuleType("stable_baselines3.common.vec_env")

    class VecEnv:
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
    ve.VecEnv = VecEnv; common.vec_env = ve; sb3.common = common
    for name, mod in (("stable_baselines3", sb3), ("stable_baselines3.common", common), ("stable_baselines3.common.vec_env", ve)):
        monkeypatch.setitem(sys.modules, name, mod)
    cls = sb3_adapter.adapter_class()
    assert issubclass(cls, VecEnv)

    class Env:      # the attributes the adapter reads at construction
        batch, observation_space, action_space = 8, "obs-space", "act-space"
    a = cls(Env())
    assert isinstance(a, VecEnv) and a.num_envs == 8 and a.observation_space == "obs-space" and a.get_attr("batch") == [8] * 8
