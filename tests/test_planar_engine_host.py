"""The kernels' math (random-envs_amd/csrc/planar_engine.hpp, the exact code the GPU runs) compiled
for the host and compared with the independent 3-D oracle: fp64 instantiation pins the planar
reduction itself, fp32 instantiation gives the tolerance the GPU tests use."""
import numpy as np
import pytest

from host_harness.build import host_constants, host_forward, host_step, last_mode, set_fast, set_line_search, set_rolled
from oracle_bindings import (DIMS, oracle_batch_step, oracle_constants, oracle_contacts, oracle_forward,
                             rollout_states)
from random_envs_amd.specs import SPECS

KINDS = ["hopper", "walker2d", "halfcheetah"]


@pytest.mark.parametrize("kind", KINDS)
def test_model_compiler_matches_oracle(kind):
    """The product's own model derivation (planar_model.hpp) vs the oracle's 3-D compile."""
    h = host_constants(kind); o = oracle_constants(kind)
    assert np.allclose(h["mass"], o["body_mass"][1:], rtol=1e-13)
    assert np.allclose(h["iyy"], o["body_inertia"][1:, 4], rtol=1e-12)
    assert np.allclose(h["tran_invw"], o["body_invweight0"][1:, 0], rtol=1e-10)
    assert np.allclose(h["dof_invw"][1:], o["dof_invweight0"][3:], rtol=1e-10)


def test_walker_model_compiler_tracks_lengths():
    size = [0.5, 0.3, 0.7, 0.25]
    h = host_constants("walker2d", size=size); o = oracle_constants("walker2d", size=size)
    assert np.allclose(h["mass"], o["body_mass"][1:], rtol=1e-13)
    assert np.allclose(h["tran_invw"], o["body_invweight0"][1:, 0], rtol=1e-10)
    assert not np.allclose(h["mass"], host_constants("walker2d")["mass"])


@pytest.mark.parametrize("kind", KINDS)
def test_forward_dynamics_fp64_matches_oracle(kind):
    d = DIMS[kind]; rng = np.random.RandomState(1)
    nom = np.array(SPECS[kind].nominal_task)
    worst = 0
    for _ in range(150):
        xi = nom * rng.uniform(0.6, 1.4, d["nx"])
        q = rng.uniform(-0.4, 0.4, d["nq"]); v = rng.uniform(-3, 3, d["nv"]); a = rng.uniform(-1.2, 1.2, d["nu"])
        q[1] = rng.uniform(-0.2, 0.3) if kind == "halfcheetah" else rng.uniform(1.05, 1.4)
        o = oracle_forward(kind, q, v, a, xi); qa, M, _ = host_forward(kind, False, q, v, a, xi)
        assert np.allclose(M, o["M"], rtol=0, atol=1e-11)
        worst = max(worst, np.abs(qa - o["qacc"]).max() / (1 + np.abs(o["qacc"]).max()))
    assert worst < 1e-6, worst


@pytest.mark.parametrize("kind", KINDS)
def test_env_step_on_rollout_states(kind):
    """one env.step (frame_skip mj_steps) from states reached by random-action rollouts"""
    d = DIMS[kind]
    q, v, xi = rollout_states(kind, 600, steps_max=50, seed=3)
    rng = np.random.RandomState(4); a = rng.uniform(-1, 1, (600, d["nu"]))
    ref = oracle_batch_step(kind, q, v, a, xi)
    q64, v64, cap = host_step(kind, False, q, v, a, xi, d["frame_skip"])
    e64 = np.abs(v64 - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
    assert np.percentile(e64, 99) < 1e-7 and cap.sum() == 0
    q32, v32, cap = host_step(kind, True, q, v, a, xi, d["frame_skip"])
    eq = np.abs(q32 - ref["qpos"]).max(1); ev = np.abs(v32 - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
    # fp32 tolerance used by the GPU parity tests
    assert np.percentile(eq, 99) < 2e-5 and np.percentile(ev, 99) < 2e-4, (eq.max(), ev.max())
    assert np.median(ev) < 1e-5


def test_hopper_self_collision_rows():
    """states where the foot folds onto the thigh / torso: capsule-capsule rows are active"""
    xi = np.array(SPECS["hopper"].nominal_task)
    hits = 0
    rng = np.random.RandomState(0)
    for _ in range(3000):
        q = np.array([0, rng.uniform(1.2, 1.6), rng.uniform(-.3, .3), rng.uniform(-2.6, -1.5), rng.uniform(-2.6, -1.5), rng.uniform(-.8, .8)])
        v = rng.uniform(-1, 1, 6); a = rng.uniform(-1, 1, 3)
        o = oracle_forward("hopper", q, v, a, xi)
        con = oracle_contacts("hopper", q, v, xi)
        self_con = con[con[:, 0] > 0]
        if len(self_con) == 0:
            continue
        # capsule axes that CROSS in the plane (dist = -(r1+r2)) have a rounding-noise normal in any
        # implementation, MuJoCo's included: ill-posed, skip
        if self_con[:, 2].min() < -0.07:
            continue
        hits += 1
        qa, _, _ = host_forward("hopper", False, q, v, a, xi)
        assert np.abs(qa - o["qacc"]).max() / (1 + np.abs(o["qacc"]).max()) < 1e-7
    assert hits > 20


def test_walker_compact_geometry_table():
    """the 25-slot compact per-env geometry (planar_model.hpp kWalkerMap) expands back to all 105 PlanarGeom
    fields for arbitrary xi lengths"""
    import ctypes
    from host_harness.build import lib as hlib
    L = hlib(); D = ctypes.POINTER(ctypes.c_double)
    rng = np.random.RandomState(0); nom = np.array([.4, .45, .6, .2])
    for _ in range(50):
        size = np.array([rng.uniform(.15, 1), rng.uniform(.15, 1), rng.uniform(.15, 1), rng.uniform(.15, 1)])
        err = ctypes.c_double()
        worst = L.ph_walker_compact_check(size.ctypes.data_as(D), nom.ctypes.data_as(D), ctypes.byref(err))
        assert err.value < 1e-15, (worst, err.value, size)   # rounding of zero offsets


@pytest.mark.parametrize("kind", KINDS)
def test_fast_and_general_solver_instantiations_agree(kind):
    """The feet-only straight-line solver instantiation (forward(): mode 3) and the general one are the same Newton
    method over the same rows: on states where only the feet touch they agree to rounding in fp64 and to the fp32
    tolerance in fp32; states with other contacts never enter the fast path (identical results)."""
    d = DIMS[kind]
    q, v, xi = rollout_states(kind, 400, steps_max=50, seed=8)
    a = np.random.RandomState(2).uniform(-1, 1, (400, d["nu"]))
    try:
        set_fast(1); qf, vf, _ = host_step(kind, False, q, v, a, xi, d["frame_skip"]); q32f, v32f, _ = host_step(kind, True, q, v, a, xi, d["frame_skip"])
        set_fast(0); qg, vg, _ = host_step(kind, False, q, v, a, xi, d["frame_skip"]); q32g, v32g, _ = host_step(kind, True, q, v, a, xi, d["frame_skip"])
    finally:
        set_fast(1)
    e = np.abs(vf - vg).max(1) / (1 + np.abs(vg).max(1))
    assert e.max() < 1e-9 and np.abs(qf - qg).max() < 1e-11, (e.max(), np.abs(qf - qg).max())
    # the fast path IS what ran: single forward evaluations report the instantiation they entered
    modes = {0: 0, 1: 0, 2: 0, 3: 0}
    for i in range(0, 400, 4):
        qa, _, _ = host_forward(kind, False, q[i], v[i], a[i], xi[i]); m = last_mode(); modes[m] += 1
        set_fast(0)
        qb, _, _ = host_forward(kind, False, q[i], v[i], a[i], xi[i]); assert last_mode() in ((0, 2) if m in (0, 2) else (1,))
        set_fast(1)
        assert np.abs(qa - qb).max() <= 1e-9 * (1 + np.abs(qb).max())
    assert modes[3] > 30, modes
    e32 = np.abs(v32f - v32g).max(1) / (1 + np.abs(v32g).max(1))
    assert np.percentile(e32, 99) < 2e-4 and np.abs(q32f - q32g).max() < 2e-5


@pytest.mark.parametrize("gen", [1, 2])
@pytest.mark.parametrize("kind", KINDS)
def test_rolled_general_solver_matches_the_unrolled_one_and_the_oracle(kind, gen):
    """solve_newton_rolled (gen 1: the general path of the 256-register hopper kernel, runtime loops over a ROW list) and solve_newton_list (gen 2:
    the general path of the two-lanes-per-env kernels, runtime loops over the list of contact units with their data in a column of memory, one-group
    correction included) are the same Newton method on the same rows as the unrolled per-slot solver: same minimiser in fp64 (both against the
    oracle too), fp32 tolerance in fp32.  Run with the feet-only path switched off so that EVERY evaluation goes through it, and on the
    self-collision states with it on."""
    d = DIMS[kind]
    q, v, xi = rollout_states(kind, 300, steps_max=50, seed=11)
    a = np.random.RandomState(5).uniform(-1, 1, (300, d["nu"]))
    ref = oracle_batch_step(kind, q, v, a, xi)
    try:
        set_fast(0)
        set_rolled(0); qu, vu, _ = host_step(kind, False, q, v, a, xi, d["frame_skip"])
        set_rolled(gen); qr, vr, cap = host_step(kind, False, q, v, a, xi, d["frame_skip"]); q32, v32, cap32 = host_step(kind, True, q, v, a, xi, d["frame_skip"])
        set_fast(1); qf, vf, _ = host_step(kind, False, q, v, a, xi, d["frame_skip"])
        modes = set()
        for i in range(0, 300, 5):
            set_fast(0); qa, _, _ = host_forward(kind, False, q[i], v[i], a[i], xi[i]); modes.add(last_mode())
            o = oracle_forward(kind, q[i], v[i], a[i], xi[i])
            assert np.abs(qa - o["qacc"]).max() / (1 + np.abs(o["qacc"]).max()) < 1e-6
    finally:
        set_rolled(0); set_fast(1)
    assert cap.sum() == 0 and cap32.sum() == 0
    assert modes <= {0, 1, 2} and (1 in modes or 2 in modes), modes
    e = np.abs(vr - vu).max(1) / (1 + np.abs(vu).max(1))
    assert e.max() < 1e-9 and np.abs(qr - qu).max() < 1e-11, (e.max(), np.abs(qr - qu).max())
    assert np.abs(qf - qr).max() < 1e-11
    e64 = np.abs(vr - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
    assert np.percentile(e64, 99) < 1e-7
    eq = np.abs(q32 - ref["qpos"]).max(1); ev = np.abs(v32 - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
    assert np.percentile(eq, 99) < 2e-5 and np.percentile(ev, 99) < 2e-4, (eq.max(), ev.max())


@pytest.mark.parametrize("gen", [1, 2])
def test_rolled_solver_on_hopper_self_collision_rows(gen):
    """the row list / the unit list carries the capsule-capsule rows (mode 2) as well"""
    xi = np.array(SPECS["hopper"].nominal_task); rng = np.random.RandomState(3); hits = 0
    try:
        set_rolled(gen)
        for _ in range(1500):
            q = np.array([0, rng.uniform(1.2, 1.6), rng.uniform(-.3, .3), rng.uniform(-2.6, -1.5), rng.uniform(-2.6, -1.5), rng.uniform(-.8, .8)])
            v = rng.uniform(-1, 1, 6); a = rng.uniform(-1, 1, 3)
            con = oracle_contacts("hopper", q, v, xi); self_con = con[con[:, 0] > 0]
            if len(self_con) == 0 or self_con[:, 2].min() < -0.07:
                continue
            o = oracle_forward("hopper", q, v, a, xi); qa, _, _ = host_forward("hopper", False, q, v, a, xi)
            assert last_mode() == 2
            assert np.abs(qa - o["qacc"]).max() / (1 + np.abs(o["qacc"]).max()) < 1e-7
            hits += 1
    finally:
        set_rolled(0)
    assert hits > 10


@pytest.mark.parametrize("kind", KINDS)
def test_line_search_schedule_does_not_change_the_solution(kind):
    """The solver's line-search schedule (SolParams.ls_free full Newton steps, then the safeguarded exact search) only
    changes HOW the unique minimiser is reached: every schedule gives the oracle's result and none hits the iteration cap."""
    d = DIMS[kind]
    q, v, xi = rollout_states(kind, 500, steps_max=70, seed=12)
    a = np.random.RandomState(6).uniform(-1, 1, (500, d["nu"]))
    ref = oracle_batch_step(kind, q, v, a, xi)
    try:
        for ls_max, ls_free in ((3, 0), (1, 0), (3, 2), (3, 4), (3, 24)):
            set_line_search(ls_max, ls_free)
            q64, v64, cap = host_step(kind, False, q, v, a, xi, d["frame_skip"])
            e = np.abs(v64 - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
            assert e.max() < 1e-6 and cap.sum() == 0, (ls_max, ls_free, e.max(), cap.sum())
    finally:
        set_line_search(-1, -1)

