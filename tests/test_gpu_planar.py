"""Parity of the planar HIP kernels (hopper / walker2d / half-cheetah), called through the C-ABI,
against the fp64 oracle on identical (qpos, qvel, action, xi); plus size-independent properties at
BASELINE.json's full batch sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IDS = {"hopper": "RandomHopper-v0", "walker2d": "RandomWalker2d-v0", "halfcheetah": "RandomHalfCheetah-v0"}
# fp32 tolerances (stated per quantity): established with the host fp32 instantiation of the same
# code in tests/test_planar_engine_host.py
TOL_QPOS_P99, TOL_QVEL_REL_P99, TOL_QVEL_REL_MED = 2e-5, 2e-4, 1e-5


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _step_from(env, torch, q, v, xi, a):
    env.set_task(np.asarray(xi, dtype=np.float32))
    env.set_state(q, v)
    obs, r, d, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
    qq, vv = env.get_state()
    return obs.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), qq.cpu().numpy(), vv.cpu().numpy()


def _done_margin(kind, qpos):
    """distance of the reference end state to the nearest threshold of the done rule (random_hopper.py:92,
    random_walker2d.py:124-125; the half-cheetah never terminates)"""
    z, th = qpos[:, 1], qpos[:, 2]
    if kind == "hopper":
        return np.minimum(np.abs(z - 0.7), np.abs(np.abs(th) - 0.2))
    if kind == "walker2d":
        return np.minimum.reduce([np.abs(z - 0.8), np.abs(z - 2.0), np.abs(th - 1.0), np.abs(th + 1.0)])
    return np.full(len(z), np.inf)


# per-lane gates (parity_util.assert_lanes_explained): stated tolerance, cap on explained outliers
TOL_QPOS, CAP_QPOS = 2e-5, 5e-4
TOL_QVEL_REL, CAP_QVEL_REL = 2e-4, 2e-2
TOL_REWARD, CAP_REWARD = 5e-3, 1e-1


@pytest.mark.parametrize("lanes", [None, 64])
@pytest.mark.parametrize("kind", ["hopper", "walker2d", "halfcheetah"])
def test_step_parity_on_rollout_states(torch_mod, kind, lanes):
    """Every lane within the stated fp32 tolerance of the oracle, or explained by the oracle's own ill-conditioning;
    once with the default launch shape and once with 64-lane blocks (the shape rex uses past 32 768 envs)."""
    import random_envs_amd as rex
    from oracle_bindings import DIMS, oracle_batch_step, oracle_sensitivity, rollout_states
    from parity_util import assert_done_explained, assert_lanes_explained, lanes_per_block
    n = 2048; d = DIMS[kind]
    q, v, xi = rollout_states(kind, n, steps_max=60, seed=11)
    # the kernel sees fp32 inputs: give the oracle the same rounded values
    q = q.astype(np.float32).astype(np.float64); v = v.astype(np.float32).astype(np.float64)
    xi = xi.astype(np.float32).astype(np.float64)
    a = np.random.RandomState(5).uniform(-1.2, 1.2, (n, d["nu"])).astype(np.float32).astype(np.float64)
    with lanes_per_block(lanes):
        env = rex.make(IDS[kind], batch=n, autoreset=False)
    obs, r, dn, qq, vv = _step_from(env, torch_mod, q, v, xi, a)
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_batch_step(kind, q_, v_, a_, x_), [q, v, a, xi],
                                   ["qpos", "qvel", "reward"])
    eq = np.abs(qq - ref["qpos"]).max(1)
    vs = 1 + np.abs(ref["qvel"]).max(1)
    ev = np.abs(vv - ref["qvel"]).max(1) / vs
    assert np.isfinite(qq).all() and np.isfinite(vv).all()
    assert np.percentile(eq, 99) < TOL_QPOS_P99, (np.percentile(eq, 99), eq.max())
    assert np.percentile(ev, 99) < TOL_QVEL_REL_P99, (np.percentile(ev, 99), ev.max())
    assert np.median(ev) < TOL_QVEL_REL_MED
    tag = "%s lanes=%s" % (kind, lanes)
    assert_lanes_explained(eq, sens["qpos"], TOL_QPOS, CAP_QPOS, label=tag + " |dqpos|")
    assert_lanes_explained(ev, sens["qvel"] / vs, TOL_QVEL_REL, CAP_QVEL_REL, label=tag + " |dqvel|rel")
    # obs = concat(qpos[1:], qvel); reward; done
    assert np.array_equal(obs, np.concatenate([qq[:, 1:], vv], 1))
    er = np.abs(r - ref["reward"])
    assert np.percentile(er, 99) < 5e-3 and np.median(er) < 2e-4, (np.percentile(er, 99), er.max())
    assert_lanes_explained(er, sens["reward"], TOL_REWARD, CAP_REWARD, label=tag + " |dreward|")
    # done differs only where the reference end state is within fp32 rounding of a threshold
    assert_done_explained(dn, ref["done"], _done_margin(kind, ref["qpos"]), 2e-5, label=tag)
    assert env.counters()["solver_capped"] == 0
    env.close()


@pytest.mark.parametrize("knobs", [dict(REX_PAIR=0), dict(REX_CORR=0), dict(REX_CORR=1), dict(REX_FAST=0), dict(REX_FAST=0, REX_PAIR=1),
                                   dict(REX_FAST=0, REX_PAIR=1, REX_CORR=0), dict(REX_PAIR=0, REX_CORR=0, REX_LS_FREE=0, REX_LS_MAX=3),
                                   dict(REX_ROLLED=1, REX_PAIR=0), dict(REX_ROLLED=1, REX_FAST=0)])
@pytest.mark.parametrize("kind", ["hopper", "walker2d", "halfcheetah"])
def test_every_solver_configuration_matches_the_oracle(torch_mod, kind, knobs):
    """The solver's machinery is switchable at create time (REX_PAIR: two lanes per env, REX_CORR: Woodbury correction off / one group / two
    groups (the default), REX_FAST: feet-only instantiation + qacc_smooth skip -- with REX_PAIR=1 beside REX_FAST=0 EVERY evaluation of every lane
    goes through the LIST solver of the two-lanes-per-env kernels (solve_newton_list: the general path of every BASELINE configuration) --,
    REX_LS_FREE / REX_LS_MAX: line-search schedule, REX_ROLLED (with one lane per env): the
    hopper's 256-register / two-waves-per-SIMD step kernel with the rolled general solver, the default past 65 536 envs -- with REX_FAST=0 every
    evaluation of every lane goes through the rolled solver).  Every
    configuration reaches the same unique minimiser: each one against the oracle, every lane, same tolerances."""
    if "REX_ROLLED" in knobs and kind != "hopper":
        pytest.skip("the rolled step kernel is the hopper's")
    import random_envs_amd as rex
    from oracle_bindings import DIMS, oracle_batch_step, rollout_states
    from parity_util import create_knobs
    n = 1024; d = DIMS[kind]
    q, v, xi = rollout_states(kind, n, steps_max=60, seed=13)
    q, v, xi = [x.astype(np.float32).astype(np.float64) for x in (q, v, xi)]
    a = np.random.RandomState(8).uniform(-1.2, 1.2, (n, d["nu"])).astype(np.float32).astype(np.float64)
    with create_knobs(**knobs):
        env = rex.make(IDS[kind], batch=n, autoreset=False)
    obs, r, dn, qq, vv = _step_from(env, torch_mod, q, v, xi, a)
    ref = oracle_batch_step(kind, q, v, a, xi)
    eq = np.abs(qq - ref["qpos"]).max(1); ev = np.abs(vv - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
    print(kind, knobs, "max |dqpos| %.2e max |dqvel|rel %.2e" % (eq.max(), ev.max()))
    assert eq.max() < TOL_QPOS and ev.max() < TOL_QVEL_REL, (kind, knobs, eq.max(), ev.max())
    assert env.counters()["solver_capped"] == 0
    env.close()


@pytest.mark.parametrize("shape", ["pair", "one_lane", "rolled"])
def test_hopper_contact_rich_and_limit_states(torch_mod, shape):
    """random (not rollout) states: deeper penetrations, joint limits violated, large velocities -- practically every wave leaves the
    feet-only path, so this is the test of the general solver in its three forms: the LIST solver of the two-lanes-per-env kernels (the default up
    to 32 768 envs: floor units and capsule-capsule rows in the LDS column, two-group correction), the unrolled per-slot one of the
    one-lane-per-env kernels, and the rolled row-list one of the two-waves-per-SIMD kernel"""
    rolled = 1 if shape == "rolled" else 0
    import random_envs_amd as rex
    from parity_util import create_knobs
    from oracle_bindings import oracle_batch_step
    from random_envs_amd.specs import SPECS
    n = 4096; rng = np.random.RandomState(7)
    xi = np.array(SPECS["hopper"].nominal_task) * rng.uniform(0.5, 1.5, (n, 4))
    q = rng.uniform(-0.4, 0.4, (n, 6)); q[:, 1] = rng.uniform(1.05, 1.4, n); q[:, 5] = rng.uniform(-0.9, 0.9, n)
    v = rng.uniform(-3, 3, (n, 6)); a = rng.uniform(-1.2, 1.2, (n, 3))
    q, v, xi, a = [x.astype(np.float32).astype(np.float64) for x in (q, v, xi, a)]
    with create_knobs(REX_ROLLED=rolled, REX_PAIR=(None if shape == "pair" else 0)):
        env = rex.make("RandomHopper-v0", batch=n, autoreset=False)
    assert env.launch_shape()["pair"] == (shape == "pair") and env.launch_shape()["rolled"] == bool(rolled)
    obs, r, dn, qq, vv = _step_from(env, torch_mod, q, v, xi, a)
    ref = oracle_batch_step("hopper", q, v, a, xi)
    vs = 1 + np.abs(ref["qvel"]).max(1)
    ev = np.abs(vv - ref["qvel"]).max(1) / vs
    assert np.percentile(ev, 99) < 5e-4 and np.median(ev) < 1e-5, (np.percentile(ev, 99), ev.max())
    # qpos, reward and done as well, every lane (these states exercise the capsule-capsule self-collision rows)
    from oracle_bindings import oracle_sensitivity
    from parity_util import assert_done_explained, assert_lanes_explained
    _, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_batch_step("hopper", q_, v_, a_, x_), [q, v, a, xi],
                                 ["qpos", "qvel", "reward"])
    assert_lanes_explained(np.abs(qq - ref["qpos"]).max(1), sens["qpos"], 5e-5, 2e-3, label="hopper contact-rich |dqpos|")
    assert_lanes_explained(ev, sens["qvel"] / vs, 5e-4, 5e-2, label="hopper contact-rich |dqvel|rel")
    assert_lanes_explained(np.abs(r - ref["reward"]), sens["reward"], 1e-2, 5e-1, label="hopper contact-rich |dreward|")
    assert_done_explained(dn, ref["done"], _done_margin("hopper", ref["qpos"]), 5e-5, label="hopper contact-rich")
    env.close()


def test_full_batch_properties_hopper(torch_mod):
    """B = 32768 (the north-star batch): determinism, x-translation invariance, shard independence."""
    import random_envs_amd as rex
    torch = torch_mod
    B = 32768
    def run(env_offset=0, batch=B, shift=0.0, steps=3):
        env = rex.make("RandomHopper-v0", batch=batch, seed=42, env_offset=env_offset, autoreset=False)
        env.set_dr_distribution("uniform", [3.0, 4.0, 3.5, 4.5, 2.2, 3.2, 4.5, 5.5])
        env.set_dr_training(True)
        env.reset()
        if shift:
            q, v = env.get_state(); q = q.clone(); q[:, 0] += shift; env.set_state(q, v)
        g = torch.Generator().manual_seed(1)
        acts = (torch.rand(steps, B, 3, generator=g) * 2 - 1)[:, env_offset:env_offset + batch]
        outs = []
        for t in range(steps):
            obs, r, d, _ = env.step(acts[t])
            outs.append((obs.clone(), r.clone(), d.clone()))
        xi = env.get_task().clone(); env.close()
        return outs, xi
    full, xi = run()
    again, _ = run()
    for (o1, r1, d1), (o2, r2, d2) in zip(full, again):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)      # bitwise deterministic
    lo = torch.tensor([3.0, 3.5, 2.2, 4.5], device=xi.device); hi = torch.tensor([4.0, 4.5, 3.2, 5.5], device=xi.device)
    assert (xi >= lo).all() and (xi <= hi).all()
    assert torch.allclose(xi.mean(0), (lo + hi) / 2, atol=0.01)
    shifted, _ = run(shift=100.0)
    for (o1, r1, d1), (o2, r2, d2) in zip(full, shifted):
        assert torch.equal(o1, o2) and torch.equal(r1, r2)         # obs excludes x; reward uses the in-kernel dx
    half, xih = run(env_offset=B // 2, batch=B // 2)
    for (o1, r1, d1), (o2, r2, d2) in zip(full, half):
        assert torch.equal(o1[B // 2:], o2) and torch.equal(r1[B // 2:], r2)            # sharding does not change results
    assert torch.equal(xi[B // 2:], xih)


@pytest.mark.parametrize("kind,B", [("halfcheetah", 16384), ("walker2d", 8192)])
def test_full_batch_finite_and_deterministic(torch_mod, kind, B):
    import random_envs_amd as rex
    from random_envs_amd.specs import SPECS
    torch = torch_mod
    nom = np.array(SPECS[kind].nominal_task)
    def run():
        env = rex.make(IDS[kind] if kind != "halfcheetah" else "RandomHalfCheetahNoisy-v0", batch=B, seed=9)
        env.set_dr_distribution("uniform", np.stack([0.8 * nom, 1.2 * nom], 1).ravel().tolist())
        env.set_dr_training(True)
        env.reset()
        g = torch.Generator().manual_seed(2)
        tot = 0
        for t in range(20):
            obs, r, d, _ = env.step(torch.rand(B, env.dims.act_dim, generator=g) * 2 - 1)
            assert torch.isfinite(obs).all() and torch.isfinite(r).all()
            tot += r.sum().item()
        c = env.counters(); env.close()
        return obs.clone(), tot, c
    o1, t1, c1 = run(); o2, t2, c2 = run()
    assert torch.equal(o1, o2) and t1 == t2
    assert c1["nonfinite"] == 0


@pytest.mark.parametrize("kind,eid", [("hopper", "RandomHopperUnmodeled-v0"), ("halfcheetah", "RandomHalfCheetahUnmodeled-v0"),
                                      ("walker2d", "RandomWalker2dUnmodeled-v0")])
def test_unmodeled_ids(torch_mod, kind, eid):
    """same kernels, reduced task vector, frozen 0.8x prefix (test.py's source env)."""
    import random_envs_amd as rex
    from oracle_bindings import DIMS, UNMODELED_NX, oracle_batch_step, oracle_sensitivity
    from parity_util import assert_lanes_explained
    torch = torch_mod
    n = 512; d = DIMS[kind]; nx = UNMODELED_NX[kind]
    env = rex.make(eid, batch=n, seed=3, autoreset=False)
    assert env.task_dim == nx and len(env.dyn_ind_to_name) == nx
    nom = np.array(env.original_task)
    assert np.allclose(env.get_task().cpu().numpy(), nom[None], rtol=1e-6)
    rng = np.random.RandomState(1)
    q = rng.uniform(-.005, .005, (n, d["nq"])); v = rng.uniform(-.3, .3, (n, d["nv"]))
    if kind != "halfcheetah":
        q[:, 1] += 1.25
    a = rng.uniform(-1, 1, (n, d["nu"]))
    q, v, a = [x.astype(np.float32).astype(np.float64) for x in (q, v, a)]
    # (1) freshly constructed env (no set_task yet): frozen prefix at 0.8x nominal
    env.set_state(q, v)
    obs, r, dn, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
    step1 = lambda q_, v_, a_, x_: oracle_batch_step(kind, q_, v_, a_, x_, variant=1)
    ref, sens = oracle_sensitivity(step1, [q, v, a, np.full((n, nx), np.nan)], ["obs"])
    os_ = 1 + np.abs(ref["obs"]).max(1)
    e0 = np.abs(obs.cpu().numpy() - ref["obs"]).max(1) / os_
    assert_lanes_explained(e0, sens["obs"] / os_, 2e-4, 2e-2, label=eid + " fresh |dobs|rel")      # every lane
    # (2) after set_task with a random reduced task
    lo = np.array([b[0] for b in env.spec.search_bounds]); hi = np.array([b[1] for b in env.spec.search_bounds])
    xi = (nom * rng.uniform(0.8, 1.2, (n, nx))).clip(lo, hi).astype(np.float32).astype(np.float64)
    env.set_task(xi.astype(np.float32)); env.set_state(q, v)
    assert np.allclose(env.get_task().cpu().numpy(), xi, rtol=1e-6)
    obs, r, dn, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
    ref, sens = oracle_sensitivity(step1, [q, v, a, xi], ["obs"])
    os_ = 1 + np.abs(ref["obs"]).max(1)
    e1 = np.abs(obs.cpu().numpy() - ref["obs"]).max(1) / os_
    assert_lanes_explained(e1, sens["obs"] / os_, 2e-4, 2e-2, label=eid + " set_task |dobs|rel")
    # (3) test.py's scenario: uniform DR on the source env, then reset resamples only the reduced task
    env.set_dr_distribution("uniform", np.stack([lo * 1.1, np.minimum(hi, lo * 1.1 + 1.0)], 1).ravel().tolist())
    env.set_dr_training(True); env.reset()
    t = env.get_task().cpu().numpy()
    assert (t >= lo * 1.1 - 1e-6).all() and (t <= np.minimum(hi, lo * 1.1 + 1.0) + 1e-6).all()
    obs, r, dn, _ = env.step(torch.zeros(n, d["nu"]))
    assert torch.isfinite(obs).all()
    env.close()


def test_episode_statistics_match_oracle(torch_mod):
    """End-to-end sanity beyond single steps (which is where parity is defined): under the same random policy
    the distribution of episode lengths / returns of the GPU env matches the fp64 oracle's (chaotic divergence
    makes trajectories differ, statistics must not)."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_batch_step
    from random_envs_amd.specs import SPECS
    torch = torch_mod
    n = 2048; rng = np.random.RandomState(0)
    nom = np.array(SPECS["hopper"].nominal_task)
    # GPU: auto-reset rollouts, collect finished-episode lengths
    env = rex.make("RandomHopper-v0", batch=n, seed=123)
    env.reset()
    ep_len = torch.zeros(n, device="cuda"); lens = []; rets = []; ep_ret = torch.zeros(n, device="cuda")
    g = torch.Generator().manual_seed(5)
    for t in range(150):
        obs, r, d, _ = env.step(torch.rand(n, 3, generator=g) * 2 - 1)
        ep_len += 1; ep_ret += r
        if d.any():
            lens.append(ep_len[d].cpu().numpy()); rets.append(ep_ret[d].cpu().numpy())
            ep_len[d] = 0; ep_ret[d] = 0
    gl = np.concatenate(lens); gr = np.concatenate(rets); env.close()
    # oracle: first episode of n envs from the same reset distribution
    q = rng.uniform(-.005, .005, (n, 6)); q[:, 1] += 1.25; v = rng.uniform(-.005, .005, (n, 6))
    xi = np.tile(nom, (n, 1)); alive = np.ones(n, bool); ol = np.zeros(n); orr = np.zeros(n)
    for t in range(150):
        idx = np.where(alive)[0]
        if len(idx) == 0:
            break
        out = oracle_batch_step("hopper", q[idx], v[idx], rng.uniform(-1, 1, (len(idx), 3)), xi[idx], tolerance=0.0)
        q[idx] = out["qpos"]; v[idx] = out["qvel"]; ol[idx] += 1; orr[idx] += out["reward"]
        alive[idx[out["done"]]] = False
    ol = ol[~alive]; orr = orr[~alive]
    assert len(gl) > 3000 and len(ol) > 1500
    assert abs(gl.mean() - ol.mean()) < 0.06 * ol.mean(), (gl.mean(), ol.mean())
    assert abs(np.median(gl) - np.median(ol)) <= 2
    assert abs(gr.mean() - orr.mean()) < 0.08 * abs(orr.mean()) + 0.5, (gr.mean(), orr.mean())
