"""CartPole HIP kernel (through the C-ABI) vs golden vectors from the reference and vs the oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _obs_to_qv(state):
    state = np.asarray(state)
    return state[:, [0, 2]], state[:, [1, 3]]          # qpos = (x, theta), qvel = (x_dot, theta_dot)


def test_step_matches_reference_golden(torch_mod, golden_dir):
    import random_envs_amd as rex
    cases = json.load(open(os.path.join(golden_dir, "cartpole_step.json")))["cases"]
    n = len(cases)
    env = rex.make("RandomCartPole-v0", batch=n, autoreset=False)
    st = np.array([c["state"] for c in cases]); q, v = _obs_to_qv(st)
    env.set_task(np.array([c["xi"] for c in cases], dtype=np.float32))
    env.set_state(q, v)
    a = torch_mod.tensor([c["action"] for c in cases], dtype=torch_mod.int32)
    obs, r, d, _ = env.step(a)
    ns = np.array([c["next_state"] for c in cases])
    assert np.abs(obs.cpu().numpy() - ns).max() < 2e-5                  # fp32 vs the reference's fp64
    assert np.array_equal(r.cpu().numpy(), np.array([c["reward"] for c in cases], dtype=np.float32))
    # done: identical except within fp32 rounding of a threshold
    thr_x, thr_t = 2.4, 12 * 2 * np.pi / 360
    near = (np.abs(np.abs(ns[:, 0]) - thr_x) < 1e-5) | (np.abs(np.abs(ns[:, 2]) - thr_t) < 1e-5)
    gd = np.array([c["done"] for c in cases])
    assert np.array_equal(d.cpu().numpy()[~near], gd[~near])
    # second step from the new state: reward 0 after done (random_cartpole.py:214-222)
    a2 = torch_mod.tensor([c["action2"] for c in cases], dtype=torch_mod.int32)
    obs2, r2, d2, _ = env.step(a2)
    ok = ~near
    assert np.array_equal(r2.cpu().numpy()[ok], np.array([c["reward2"] for c in cases], dtype=np.float32)[ok])
    assert np.abs(obs2.cpu().numpy() - np.array([c["next_state2"] for c in cases])).max() < 5e-5
    env.close()


def test_rollout_matches_reference_golden(torch_mod, golden_dir):
    import random_envs_amd as rex
    ros = json.load(open(os.path.join(golden_dir, "cartpole_rollout.json")))["rollouts"]
    env = rex.make("RandomCartPole-v0", batch=len(ros), autoreset=False, time_limit=False)
    env.set_task(np.array([ro["xi"] for ro in ros], dtype=np.float32))
    q, v = _obs_to_qv(np.array([ro["states"][0] for ro in ros]))
    env.set_state(q, v)
    for t in range(60):   # fp32 drift stays small over 60 steps
        a = torch_mod.tensor([ro["actions"][t] for ro in ros], dtype=torch_mod.int32)
        obs, r, d, _ = env.step(a)
        ref = np.array([ro["states"][t + 1] for ro in ros])
        assert np.abs(obs.cpu().numpy() - ref).max() < 1e-3 * (1 + np.abs(ref).max())
    env.close()


def test_large_batch_vs_oracle_and_reset(torch_mod):
    import random_envs_amd as rex
    from oracle_bindings import oracle_cartpole_step
    B = 32768
    env = rex.make("RandomCartPole-v0", batch=B, seed=3, autoreset=False)
    env.set_dr_distribution("uniform", [5, 15, 0.8, 1.2, 0.08, 0.12, 0.4, 0.6])
    t0 = env.get_task().cpu().numpy().copy()
    env.set_dr_training(True)
    obs = env.reset().cpu().numpy()
    assert obs.min() >= -0.05 and obs.max() <= 0.05 and abs(obs.mean()) < 1e-3     # random_cartpole.py:227
    assert abs(obs.std() - 0.1 / np.sqrt(12)) < 1e-3
    assert np.array_equal(env.get_task().cpu().numpy(), t0)       # reset() never resamples
    env.set_random_task()
    xi = env.get_task().cpu().numpy()
    lo = np.array([5, .8, .08, .4]); hi = np.array([15, 1.2, .12, .6])
    assert (xi >= lo).all() and (xi <= hi).all() and np.abs(xi.mean(0) - (lo + hi) / 2).max() < 0.02 * (hi - lo).max()
    q, v = env.get_state()
    st = np.stack([q[:, 0].cpu(), v[:, 0].cpu(), q[:, 1].cpu(), v[:, 1].cpu()], 1).astype(np.float64)
    a = torch_mod.randint(0, 2, (B,), dtype=torch_mod.int32)
    o, r, d, _ = env.step(a)
    ns, rr, dd = oracle_cartpole_step(st, a.numpy(), xi.astype(np.float64))
    assert np.abs(o.cpu().numpy() - ns).max() < 1e-5
    env.close()


def test_autoreset_and_time_limit(torch_mod):
    import random_envs_amd as rex
    B = 512
    env = rex.make("RandomCartPole-v0", batch=B, seed=1)
    env.reset()
    ndone = 0
    for t in range(520):
        a = torch_mod.ones(B, dtype=torch_mod.int32)     # always push right: falls over quickly
        obs, r, d, info = env.step(a)
        ndone += int(d.sum())
        if d.any():
            ob = obs.cpu().numpy()[d.cpu().numpy()]
            assert np.abs(ob).max() <= 0.05 + 1e-7       # done lanes already hold the reset observation
            term = info["terminal_observation"].cpu().numpy()[d.cpu().numpy()]
            assert (np.abs(term[:, 0]) > 2.4 - 1e-4).any() or (np.abs(term[:, 2]) > 0.2).any()
    assert ndone > B
    assert env.step_count() == 520 * B
    env.close()
