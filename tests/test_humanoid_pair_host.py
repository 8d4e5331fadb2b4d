"""The two-lanes-per-env humanoid engine (random-envs_amd/csrc/humanoid_pair.hpp: the step kernel's math) compiled for the host
-- the two lanes of a pair as two lock-stepped threads, the env's LDS column as shared memory -- against the independent
fp64 oracle: a whole env step (20 forward evaluations, RK4), the 376-dim observation, reward, done; standing, crouched and
piled-up states (all three solver paths); fp64 pins the algorithm, fp32 sets the GPU tolerance."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle_bindings import _p, oracle_humanoid_step
from random_envs_amd.specs import SPECS

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "host_harness", "_build_humanoid_pair_host.so")
SRC = os.path.join(HERE, "host_harness", "humanoid_pair_host.cpp")
DEPS = [SRC] + [os.path.join(os.path.dirname(HERE), "random-envs_amd", "csrc", f) for f in
                ("humanoid_pair.hpp", "humanoid_engine.hpp", "humanoid_model.hpp", "planar_spec.hpp")]
UB = ctypes.POINTER(ctypes.c_ubyte); I = ctypes.POINTER(ctypes.c_int)


@pytest.fixture(scope="module")
def hp():
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-o", SO, SRC])
    return ctypes.CDLL(SO)


def _step(hp, f32, q, v, a, xi, xprev=None):
    n = q.shape[0]
    qs, vs, as_, xs = [np.ascontiguousarray(x.T) for x in (q, v, a, xi)]
    xp = None if xprev is None else np.ascontiguousarray(xprev.T)
    qo = np.zeros_like(qs); vo = np.zeros_like(vs); obs = np.zeros((376, n)); r = np.zeros(n); d = np.zeros(n, dtype=np.uint8)
    xo = np.zeros((14, n)); ov = np.zeros(n, dtype=np.int32); nr = np.zeros(n, dtype=np.int32)
    hp.hp_step(f32, n, _p(qs), _p(vs), _p(as_), _p(xs), _p(xp), _p(qo), _p(vo), _p(obs), _p(r), d.ctypes.data_as(UB), _p(xo),
               ov.ctypes.data_as(I), nr.ctypes.data_as(I))
    return dict(qpos=qo.T, qvel=vo.T, obs=obs.T, reward=r, done=d.astype(bool), xipos_x=xo.T, overflow=ov, nrows=nr)


def _states(n, seed, spread=0.3):
    rng = np.random.RandomState(seed)
    nom = np.array(SPECS["humanoid"].nominal_task)
    q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1)) + rng.uniform(-.01, .01, (n, 24))
    q[:, 7:] += rng.uniform(-spread, spread, (n, 17)); q[:, 2] = rng.uniform(1.0, 1.45, n)
    v = rng.uniform(-1, 1, (n, 23)); a = rng.uniform(-.5, .5, (n, 17)); xi = nom * rng.uniform(.8, 1.2, (n, 30))
    return q, v, a, xi


def test_side_bodies_have_no_orientation_offset(hp):
    assert hp.hp_check_model() == 1


def test_env_step_vs_oracle(hp):
    n = 96
    q, v, a, xi = _states(n, 1)
    ref = oracle_humanoid_step(q, v, a, xi)
    for f32, tv, to in ((0, 1e-10, 1e-10), (1, 1e-4, 2e-5)):
        out = _step(hp, f32, q, v, a, xi)
        ev = np.abs(out["qvel"] - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
        eo = np.abs(out["obs"] - ref["obs"]).max(1) / (1 + np.abs(ref["obs"]).max(1))
        assert ev.max() < tv and eo.max() < to and out["overflow"].sum() == 0, (f32, ev.max(), eo.max())
        assert np.abs(out["reward"] - ref["reward"]).max() < (1e-9 if not f32 else 1e-4)
        assert np.array_equal(out["done"], ref["done"])
        assert np.abs(out["xipos_x"] - ref["xipos_x"]).max() < (1e-12 if not f32 else 1e-5)
    # second step: mass_center() "before" from the xipos the first step left behind
    a2 = np.random.RandomState(3).uniform(-.4, .4, (n, 17))
    out1 = _step(hp, 0, q, v, a, xi)
    ref2 = oracle_humanoid_step(ref["qpos"], ref["qvel"], a2, xi, xipos_x_prev=ref["xipos_x"])
    out2 = _step(hp, 0, out1["qpos"], out1["qvel"], a2, xi, xprev=out1["xipos_x"])
    assert np.abs(out2["reward"] - ref2["reward"]).max() < 1e-8


def test_pile_ups_all_solver_paths(hp):
    """lying / crumpled humanoids with hinges past their limits: <= 16 rows, 17..21 rows and the scratch-row PGS (> 21)"""
    n = 64; rng = np.random.RandomState(5)
    nom = np.array(SPECS["humanoid"].nominal_task)
    q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1))
    q[:, 7:] += rng.uniform(-1.2, 1.2, (n, 17)); q[:, 2] = rng.uniform(0.05, 0.6, n)
    qq = np.array([1, 0, 0, 0]) + rng.uniform(-1, 1, (n, 4)); q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    v = rng.uniform(-1, 1, (n, 23)); a = rng.uniform(-.4, .4, (n, 17)); xi = nom * rng.uniform(.9, 1.1, (n, 30))
    ref = oracle_humanoid_step(q, v, a, xi)
    out = _step(hp, 0, q, v, a, xi)
    ok = np.isfinite(ref["qvel"]).all(1)
    ev = np.abs(out["qvel"] - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
    assert out["overflow"].sum() == 0 and ok.sum() > n // 2
    assert (out["nrows"] > 21).sum() >= 3 and (out["nrows"] <= 16).sum() >= 1, np.sort(out["nrows"])
    assert ev[ok].max() < 1e-7, ev[ok].max()

