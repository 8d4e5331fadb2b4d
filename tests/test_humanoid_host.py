"""Humanoid: the kernels' math (random-envs_amd/csrc/humanoid_engine.hpp: MuJoCo-style com-based CRB / RNE,
incremental PGS) compiled for the host, against the independent 3-D oracle (world-frame Jacobian sums,
dual PGS with an explicit A matrix)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle_bindings import _p, lib, oracle_energy_drift, oracle_humanoid_step
from random_envs_amd.specs import SPECS

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "host_harness", "_build_humanoid_host.so")
SRC = os.path.join(HERE, "host_harness", "humanoid_host.cpp")
DEPS = [SRC] + [os.path.join(os.path.dirname(HERE), "random-envs_amd", "csrc", f) for f in
                ("humanoid_engine.hpp", "humanoid_model.hpp", "planar_spec.hpp")]


@pytest.fixture(scope="module")
def hh():
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO, SRC])
    return ctypes.CDLL(SO)


def _states(n, seed, spread=0.3):
    rng = np.random.RandomState(seed)
    nom = np.array(SPECS["humanoid"].nominal_task)
    q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1)) + rng.uniform(-.01, .01, (n, 24))
    q[:, 7:] += rng.uniform(-spread, spread, (n, 17)); q[:, 2] = rng.uniform(1.0, 1.45, n)
    v = rng.uniform(-1, 1, (n, 23)); a = rng.uniform(-.5, .5, (n, 17)); xi = nom * rng.uniform(.8, 1.2, (n, 30))
    return q, v, a, xi


def test_published_masses_and_model_compiler(hh):
    """12 of the 13 body masses equal the public mujoco-py-era Humanoid-v2 constants (2.1.0 capsule volume);
    the pelvis value recalled for that table (6.61619413) is the exact-capsule one and is not asserted."""
    mass = np.zeros(14); ib = np.zeros(28); idf = np.zeros(23); ipos = np.zeros(42); inr = np.zeros(84); npair = ctypes.c_int()
    hh.hh_constants(_p(mass), _p(ib), _p(idf), _p(ipos), _p(inr), ctypes.byref(npair))
    pub = [8.32207894, 2.03575204, None, 4.52555626, 2.63249442, 1.76714587, 4.52555626, 2.63249442, 1.76714587,
           1.59405984, 1.19834313, 1.59405984, 1.19834313]
    for k, p in enumerate(pub):
        if p is not None:
            assert abs(mass[1 + k] - p) < 5e-9
    assert np.allclose(mass[1:], SPECS["humanoid"].nominal_task[:13], rtol=1e-13)
    bm = np.zeros(14); nc = ctypes.c_int()
    n_or = lib().mjo_humanoid_probe(None, None, None, None, _p(bm), None, None, ctypes.byref(nc), ctypes.byref(nc), ctypes.byref(nc), None, None, 0)
    assert n_or == npair.value == 126 and np.allclose(bm, mass, rtol=1e-13)


def test_compile_time_dof_tree_matches_model_tables(hh):
    """the static mass-matrix code is generated from constexpr tables; they must agree with the XML transcription"""
    assert hh.hh_check_topology() == 1


def test_oracle_energy_conservation_3d():
    rng = np.random.RandomState(0)
    q = np.array([0, 0, 3.0, 1, 0, 0, 0] + [0] * 17, dtype=float); q[7:] += rng.uniform(-.3, .3, 17)
    qq = rng.randn(4); q[3:7] = qq / np.linalg.norm(qq)
    e0, e1 = oracle_energy_drift("humanoid", 100, q, rng.uniform(-1, 1, 23))
    assert abs(e1 - e0) < 1e-6 * abs(e0)


def test_forward_dynamics_fp64(hh):
    """qacc incl. un-converged PGS iterates (same row order => same sweep sequence) to rounding."""
    O = lib(); rng = np.random.RandomState(0)
    nom = np.array(SPECS["humanoid"].nominal_task); worst = 0; ncon_seen = 0
    for _ in range(120):
        q = np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float); q[7:] += rng.uniform(-.4, .4, 17); q[2] = rng.uniform(0.9, 1.5)
        qq = np.array([1, 0, 0, 0]) + rng.uniform(-.3, .3, 4); q[3:7] = qq / np.linalg.norm(qq)
        v = rng.uniform(-2, 2, 23); a = rng.uniform(-.5, .5, 17); xi = nom * rng.uniform(.7, 1.3, 30)
        qa_o = np.zeros(23); M_o = np.zeros((23, 23)); nc, ne, it = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        O.mjo_humanoid_probe(_p(q), _p(v), _p(a), _p(xi), None, _p(qa_o), _p(M_o), ctypes.byref(nc), ctypes.byref(ne), ctypes.byref(it), None, None, 0)
        qa_h = np.zeros(23); M_h = np.zeros((23, 23)); info = np.zeros(4, dtype=np.int32)
        hh.hh_forward(0, _p(q), _p(v), _p(a), _p(xi), _p(qa_h), _p(M_h), info.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert info[0] == nc.value and info[1] == ne.value and info[3] == 0
        assert np.abs(M_o - M_h).max() < 1e-12
        worst = max(worst, np.abs(qa_o - qa_h).max() / (1 + np.abs(qa_o).max())); ncon_seen += nc.value
    assert worst < 1e-9 and ncon_seen > 50


def test_forward_dynamics_pile_ups(hh):
    """Row counts across the solver's three paths: <= 16 rows, 17..21 rows (largest in-LDS sweep size) and > 21 rows
    (scratch-row fallback): a humanoid lying / crouching on the floor with joints past their limits."""
    O = lib(); rng = np.random.RandomState(5)
    nom = np.array(SPECS["humanoid"].nominal_task); seen = {"le16": 0, "17_21": 0, "gt21": 0}; worst = 0
    for k in range(160):
        q = np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float)
        q[7:] += rng.uniform(-1.2, 1.2, 17)                               # many hinges beyond their range -> limit rows
        q[2] = rng.uniform(0.05, 0.6)                                     # low: torso / limbs on the floor
        qq = np.array([1, 0, 0, 0]) + rng.uniform(-1, 1, 4); q[3:7] = qq / np.linalg.norm(qq)
        v = rng.uniform(-1, 1, 23); a = rng.uniform(-.4, .4, 17); xi = nom * rng.uniform(.9, 1.1, 30)
        qa_o = np.zeros(23); M_o = np.zeros((23, 23)); nc, ne, it = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        O.mjo_humanoid_probe(_p(q), _p(v), _p(a), _p(xi), None, _p(qa_o), _p(M_o), ctypes.byref(nc), ctypes.byref(ne), ctypes.byref(it), None, None, 0)
        qa_h = np.zeros(23); M_h = np.zeros((23, 23)); info = np.zeros(4, dtype=np.int32)
        hh.hh_forward(0, _p(q), _p(v), _p(a), _p(xi), _p(qa_h), _p(M_h), info.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert info[3] == 0, (k, "rows dropped", nc.value, ne.value)       # storage is sized to the model: nothing is ever dropped
        assert info[0] == nc.value and info[1] == ne.value, (k, info, nc.value, ne.value)
        seen["le16" if ne.value <= 16 else ("17_21" if ne.value <= 21 else "gt21")] += 1
        seen["gt64"] = seen.get("gt64", 0) + (ne.value > 64)
        worst = max(worst, np.abs(qa_o - qa_h).max() / (1 + np.abs(qa_o).max()))
    assert min(seen.values()) >= 3, seen                                    # incl. states beyond round 1's 64-row cap
    assert worst < 1e-8, worst


def test_env_step_obs_reward(hh):
    n = 200
    q, v, a, xi = _states(n, 1)
    ref = oracle_humanoid_step(q, v, a, xi)
    UB = ctypes.POINTER(ctypes.c_ubyte); I = ctypes.POINTER(ctypes.c_int)
    qs, vs, as_, xs = [np.ascontiguousarray(x.T) for x in (q, v, a, xi)]
    for f32, tv, to in ((0, 1e-11, 1e-11), (1, 1e-4, 2e-5)):
        qo = np.zeros_like(qs); vo = np.zeros_like(vs); obs = np.zeros((376, n)); r = np.zeros(n); d = np.zeros(n, dtype=np.uint8)
        xo = np.zeros((14, n)); ov = np.zeros(n, dtype=np.int32)
        hh.hh_step(f32, n, _p(qs), _p(vs), _p(as_), _p(xs), None, _p(qo), _p(vo), _p(obs), _p(r), d.ctypes.data_as(UB), _p(xo), ov.ctypes.data_as(I))
        ev = np.abs(vo.T - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
        eo = np.abs(obs.T - ref["obs"]).max(1) / (1 + np.abs(ref["obs"]).max(1))
        assert ev.max() < tv and eo.max() < to and ov.sum() == 0, (f32, ev.max(), eo.max())
        assert np.abs(r - ref["reward"]).max() < (1e-10 if not f32 else 1e-4)
        assert np.array_equal(d.astype(bool), ref["done"])
    # the obs layout: 22 + 23 + 140 + 84 + 23 + 84 (random_humanoid.py:207-216); cfrc_ext block is zero
    assert ref["obs"].shape[1] == 376 and np.all(ref["obs"][:, 292:] == 0) and np.all(ref["obs"][:, 45:55] == 0)


def test_second_broad_phase_test_never_changes_a_result(hh):
    """The projected segment-distance bounds of the broad phase (humanoid_engine.hpp::collide) only ever drop pairs the
    narrow phase would find apart: a harness built without them (-DREX_NO_SECOND_CULL) gives the same contacts, the same
    rows and bit-identical accelerations -- standing, crouched, lying and crumpled bodies, random orientations."""
    so2 = SO.replace(".so", "_nocull.so")
    if not os.path.exists(so2) or any(os.path.getmtime(d) > os.path.getmtime(so2) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DREX_NO_SECOND_CULL", "-o", so2, SRC])
    ref = ctypes.CDLL(so2)
    rng = np.random.RandomState(11)
    n = 3000
    q, v, a, xi = _states(n, 12, spread=0.3)
    # a third crouched / contorted (joint angles far out), a third low and tilted arbitrarily (lying, crumpled)
    q[n // 3:, 7:] = rng.uniform(-1.6, 1.6, (n - n // 3, 17))
    quat = rng.normal(size=(n - 2 * (n // 3), 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    q[2 * (n // 3):, 3:7] = quat; q[2 * (n // 3):, 2] = rng.uniform(0.05, 0.6, n - 2 * (n // 3))
    ncon = 0
    for f32 in (0, 1):
        for i in range(n):
            qa1 = np.zeros(23); qa2 = np.zeros(23); M = np.zeros(23 * 23); i1 = np.zeros(4, dtype=np.int32); i2 = np.zeros(4, dtype=np.int32)
            hh.hh_forward(f32, _p(q[i]), _p(v[i]), _p(a[i]), _p(xi[i]), _p(qa1), _p(M), i1.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
            ref.hh_forward(f32, _p(q[i]), _p(v[i]), _p(a[i]), _p(xi[i]), _p(qa2), _p(M), i2.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
            assert (i1[:2] == i2[:2]).all(), (i, i1, i2)
            assert np.array_equal(qa1, qa2), i
            ncon += int(i1[0])
    assert ncon > 4 * n      # the sample is contact-rich (capsule-capsule pairs included)


def test_fp32_host_engine_at_the_device_sweep_schedule(hh):
    """on the device the PGS stopping rule is evaluated every REX_PGS_CHECK-th (10th) sweep, in the host harness every sweep.  A harness
    built with -DREX_PGS_CHECK_HOST runs the fp32 engine at the DEVICE's schedule: sweep counts are then the every-sweep counts rounded up to
    the next multiple of 10 (never past MuJoCo's cap of 50), and one env step still agrees with the fp64 oracle within the fp32 tolerance the
    GPU tests use -- the extra sweeps act on a system already converged to the tolerance."""
    so3 = os.path.join(HERE, "host_harness", "_build_humanoid_host_devsched.so")
    if not os.path.exists(so3) or any(os.path.getmtime(d) > os.path.getmtime(so3) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DREX_PGS_CHECK_HOST", "-o", so3, SRC])
    hd = ctypes.CDLL(so3)
    rng = np.random.RandomState(3)
    nom = np.array(SPECS["humanoid"].nominal_task)
    I = ctypes.POINTER(ctypes.c_int)
    seen_rows = 0; rounded = 0
    for k in range(80):
        q = np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float); q[7:] += rng.uniform(-.6, .6, 17); q[2] = rng.uniform(0.7, 1.35)
        qq = np.array([1, 0, 0, 0]) + rng.uniform(-.3, .3, 4); q[3:7] = qq / np.linalg.norm(qq)
        v = rng.uniform(-2, 2, 23); a = rng.uniform(-.4, .4, 17); xi = nom * rng.uniform(.8, 1.2, 30)
        out = []
        for h in (hh, hd):
            qa = np.zeros(23); M = np.zeros((23, 23)); info = np.zeros(4, dtype=np.int32)
            h.hh_forward(1, _p(q), _p(v), _p(a), _p(xi), _p(qa), _p(M), info.ctypes.data_as(I))
            out.append((qa, info.copy()))
        (qa1, i1), (qa2, i2) = out
        assert i1[0] == i2[0] and i1[1] == i2[1] and i2[3] == 0
        if i1[1] == 0 or i1[1] > 21:      # no rows / the scratch-row path (which checks every sweep on the device too)
            continue
        seen_rows += 1
        assert i2[2] == min(50, -(-i1[2] // 10) * 10), (k, i1, i2)      # ceil to the next multiple of 10, capped at 50
        rounded += int(i2[2] != i1[2])
        assert np.abs(qa1 - qa2).max() <= 2e-5 * (1 + np.abs(qa1).max()), (k, np.abs(qa1 - qa2).max())
    assert seen_rows >= 20 and rounded >= 5, (seen_rows, rounded)
    # a whole env step at the device schedule against the fp64 oracle: the GPU tests' fp32 tolerances
    n = 200
    q, v, a, xi = _states(n, 1)
    ref = oracle_humanoid_step(q, v, a, xi)
    UB = ctypes.POINTER(ctypes.c_ubyte)
    qs, vs, as_, xs = [np.ascontiguousarray(x.T) for x in (q, v, a, xi)]
    qo = np.zeros_like(qs); vo = np.zeros_like(vs); obs = np.zeros((376, n)); r = np.zeros(n); d = np.zeros(n, dtype=np.uint8)
    xo = np.zeros((14, n)); ov = np.zeros(n, dtype=np.int32)
    hd.hh_step(1, n, _p(qs), _p(vs), _p(as_), _p(xs), None, _p(qo), _p(vo), _p(obs), _p(r), d.ctypes.data_as(UB), _p(xo), ov.ctypes.data_as(I))
    ev = np.abs(vo.T - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
    eo = np.abs(obs.T - ref["obs"]).max(1) / (1 + np.abs(ref["obs"]).max(1))
    assert ev.max() < 5e-4 and eo.max() < 2e-4 and ov.sum() == 0, (ev.max(), eo.max())
    assert np.abs(r - ref["reward"]).max() < 2e-3
