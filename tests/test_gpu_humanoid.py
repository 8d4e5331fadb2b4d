"""Humanoid HIP kernels (through the C-ABI) vs the fp64 oracle: 376-dim observation, reward with
mass_center(), done; reset semantics; BASELINE config 5 shape (uniform 30-dim xi inside the search bounds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _states(n, seed):
    from random_envs_amd.specs import SPECS
    rng = np.random.RandomState(seed)
    nom = np.array(SPECS["humanoid"].nominal_task)
    q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1)) + rng.uniform(-.01, .01, (n, 24))
    q[:, 7:] += rng.uniform(-.3, .3, (n, 17)); q[:, 2] = rng.uniform(1.0, 1.45, n)
    v = rng.uniform(-1, 1, (n, 23)); a = rng.uniform(-.5, .5, (n, 17)); xi = nom * rng.uniform(.8, 1.2, (n, 30))
    return [x.astype(np.float32).astype(np.float64) for x in (q, v, a, xi)]


# per-lane gates (parity_util): stated fp32 tolerance for the humanoid (PGS capped at 50 sweeps amplifies rounding), cap
TOL_OBS, CAP_OBS = 2e-4, 2e-2
TOL_QVEL, CAP_QVEL = 5e-4, 5e-2
TOL_REW, CAP_REW = 2e-3, 2e-1


@pytest.mark.parametrize("lanes", [None, 64])
def test_step_parity_and_second_step(torch_mod, lanes):
    """Every lane within tolerance or explained by the oracle's own conditioning; default launch shape and 64-lane blocks
    (the > 64 KB dynamic-LDS configuration rex uses past 32 768 envs)."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_humanoid_step, oracle_sensitivity
    from parity_util import assert_done_explained, assert_lanes_explained, lanes_per_block
    torch = torch_mod
    n = 1024
    q, v, a, xi = _states(n, 3)
    with lanes_per_block(lanes):
        env = rex.make("RandomHumanoid-v0", batch=n, autoreset=False)
    assert env.task_dim == 30 and env.dims.obs_dim == 376 and env.dims.act_dim == 17
    env.set_task(xi.astype(np.float32)); env.set_state(q, v)
    obs, r, d, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_humanoid_step(q_, v_, a_, x_), [q, v, a, xi],
                                   ["obs", "qvel", "reward"], trials=2)
    o = obs.cpu().numpy().astype(np.float64)
    os_ = 1 + np.abs(ref["obs"]).max(1)
    eo = np.abs(o - ref["obs"]).max(1) / os_
    qq, vv = env.get_state()
    vs = 1 + np.abs(ref["qvel"]).max(1)
    ev = np.abs(vv.cpu().numpy() - ref["qvel"]).max(1) / vs
    # stated fp32 tolerance for the humanoid (PGS capped at 50 sweeps amplifies rounding): p99 < 5e-4, median < 2e-5
    assert np.percentile(ev, 99) < 5e-4 and np.median(ev) < 2e-5, (np.percentile(ev, 99), ev.max())
    assert np.percentile(eo, 99) < 2e-4, (np.percentile(eo, 99), eo.max())
    tag = "humanoid lanes=%s" % lanes
    assert_lanes_explained(eo, sens["obs"] / os_, TOL_OBS, CAP_OBS, label=tag + " |dobs|rel")
    assert_lanes_explained(ev, sens["qvel"] / vs, TOL_QVEL, CAP_QVEL, label=tag + " |dqvel|rel")
    er = np.abs(r.cpu().numpy() - ref["reward"])
    assert np.percentile(er, 99) < 2e-3
    assert_lanes_explained(er, sens["reward"], TOL_REW, CAP_REW, label=tag + " |dreward|")
    z = ref["qpos"][:, 2]
    assert_done_explained(d.cpu().numpy(), ref["done"], np.minimum(np.abs(z - 1.0), np.abs(z - 2.0)), 2e-5, label=tag)
    assert np.all(o[:, 292:] == 0)                                   # cfrc_ext block
    # second step: mass_center() "before" comes from the xipos the previous step left behind
    a2 = np.random.RandomState(9).uniform(-.4, .4, (n, 17)).astype(np.float32).astype(np.float64)
    q1, v1 = qq.cpu().numpy().astype(np.float64), vv.cpu().numpy().astype(np.float64)
    obs2, r2, d2, _ = env.step(torch.as_tensor(a2, dtype=torch.float32))
    ref2, sens2 = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_humanoid_step(q_, v_, a_, x_, xipos_x_prev=ref["xipos_x"]),
                                     [q1, v1, a2, xi], ["reward", "obs"], trials=2)
    er2 = np.abs(r2.cpu().numpy() - ref2["reward"])
    assert_lanes_explained(er2, sens2["reward"], 5e-3, 2e-1, label=tag + " second step |dreward|")      # every lane
    os2 = 1 + np.abs(ref2["obs"]).max(1)
    assert_lanes_explained(np.abs(obs2.cpu().numpy() - ref2["obs"]).max(1) / os2, sens2["obs"] / os2, TOL_OBS, CAP_OBS, label=tag + " second step |dobs|rel")
    c = env.counters(); assert c["nonfinite"] == 0 and c["overflow"] == 0
    env.close()


@pytest.mark.parametrize("lanes", [None, 64])
def test_reset_and_set_state_observation_vs_oracle(torch_mod, lanes):
    """a9 / SURVEY Q10: reset_model = set_state (sim.forward() with the masses in force, i.e. the PREVIOUS episode's)
    -> set_random_task -> _get_obs (random_humanoid.py:219-234).  humanoid_reset_kernel and humanoid_forward_kernel
    against oracle/mjo_humanoid.c::mjo_humanoid_batch_reset_obs, the cinert block with the OLD masses."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_humanoid_reset_obs
    from parity_util import lanes_per_block
    torch = torch_mod
    n = 2048
    with lanes_per_block(lanes):
        env = rex.make("RandomHumanoid-v0", batch=n, seed=21, autoreset=False)
    lo, hi = env.get_task_search_bounds()
    env.set_dr_distribution("uniform", np.stack([lo, hi], 1).ravel().tolist()); env.set_dr_training(True)
    env.reset()                                                       # episode 1: xi_1 drawn
    xi_old = env.get_task().cpu().numpy().astype(np.float64)
    obs = env.reset().cpu().numpy().astype(np.float64)                # episode 2: obs built with xi_1, THEN xi_2 drawn
    xi_new = env.get_task().cpu().numpy().astype(np.float64)
    assert np.abs(xi_new - xi_old).max() > 0.1
    q, v = env.get_state()
    q = q.cpu().numpy().astype(np.float64); v = v.cpu().numpy().astype(np.float64)
    ref_old, _ = oracle_humanoid_reset_obs(q, v, xi_old)
    ref_new, _ = oracle_humanoid_reset_obs(q, v, xi_new)
    sc = 1 + np.abs(ref_old).max(1)
    e_old = np.abs(obs - ref_old).max(1) / sc
    e_new = np.abs(obs - ref_new).max(1) / sc
    print("reset obs vs oracle(old masses): max %.2e | vs oracle(new masses): median %.2e" % (e_old.max(), np.median(e_new)))
    assert e_old.max() < 2e-5, e_old.max()                            # one forward, no solver: every lane, tight
    assert np.median(e_new) > 1e-2                                    # ... and it is NOT the new masses' observation
    assert np.array_equal(obs[:, :22], q[:, 2:].astype(np.float32).astype(np.float64))
    # set_state -> sim.forward() -> _get_obs with the CURRENT task (humanoid_forward_kernel via rex_get_obs)
    rng = np.random.RandomState(4)
    q2 = q.copy(); q2[:, 7:] += rng.uniform(-.4, .4, (n, 17)); v2 = rng.uniform(-2, 2, (n, 23))
    q2, v2 = q2.astype(np.float32).astype(np.float64), v2.astype(np.float32).astype(np.float64)
    env.set_state(q2, v2)
    o2 = torch.empty(376, n, device="cuda")
    import ctypes
    from random_envs_amd import _native
    _native.check(_native.lib().rex_get_obs(env._h, ctypes.c_void_p(o2.data_ptr()), env._stream()))
    o2 = o2.t().cpu().numpy().astype(np.float64)
    ref2, xip = oracle_humanoid_reset_obs(q2, v2, xi_new)
    e2 = np.abs(o2 - ref2).max(1) / (1 + np.abs(ref2).max(1))
    assert e2.max() < 2e-5, e2.max()
    # the xipos that forward left behind is what the next step's mass_center() "before" reads
    st = env.get_full_state()
    assert np.abs(st["aux"].t().cpu().numpy() - xip).max() < 1e-5
    env.close()


def test_reset_dr_and_rollout_config5(torch_mod):
    """BASELINE config 5 shape on one GPU shard: uniform 30-dim xi inside the search bounds, U(-0.4,0.4) actions."""
    import random_envs_amd as rex
    torch = torch_mod
    B = 4096
    env = rex.make("RandomHumanoid-v0", batch=B, seed=11)
    lo, hi = env.get_task_search_bounds()
    env.set_dr_distribution("uniform", np.stack([lo, hi], 1).ravel().tolist())
    env.set_dr_training(True)
    obs = env.reset()
    assert obs.shape == (B, 376) and torch.isfinite(obs).all()
    xi = env.get_task().cpu().numpy()
    assert (xi >= lo - 1e-5).all() and (xi <= hi + 1e-5).all() and np.abs(xi.mean(0) - (lo + hi) / 2).max() < 0.15 * (hi - lo).max()
    q, v = env.get_state()
    q0 = np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17)
    assert np.abs(q.cpu().numpy() - q0).max() <= 0.01 + 1e-6 and np.abs(v.cpu().numpy()).max() <= 0.01 + 1e-7   # random_humanoid.py:220-229
    g = torch.Generator().manual_seed(0); ndone = 0
    for t in range(30):
        obs, r, d, info = env.step(torch.rand(B, 17, generator=g) * 0.8 - 0.4)
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
        ndone += int(d.sum())
        if d.any():
            z = info["terminal_observation"][d][:, 0]
            assert ((z < 1.0) | (z > 2.0)).all()                     # done = z < 1 or z > 2 (random_humanoid.py:173)
            assert (obs[d][:, 0] - 1.4).abs().max() <= 0.01 + 1e-5   # auto-reset lanes restart near qpos0
    assert ndone > 0
    c = env.counters(); assert c["nonfinite"] == 0
    env.close()


def test_noisy_humanoid_only_noises_qpos_qvel(torch_mod):
    import random_envs_amd as rex
    torch = torch_mod
    B = 2048
    clean = rex.make("RandomHumanoid-v0", batch=B, seed=5, autoreset=False)
    noisy = rex.make("RandomHumanoidNoisy-v0", batch=B, seed=5, autoreset=False)
    clean.reset(); noisy.reset()
    q, v = clean.get_state(); noisy.set_state(q, v)
    a = torch.zeros(B, 17)
    oc = clean.step(a)[0]; on = noisy.step(a)[0]
    diff = (on - oc).cpu().numpy()
    assert abs(diff[:, :45].std() - np.sqrt(1e-3)) < 2e-3            # sigma = sqrt(1e-3) (random_humanoid.py:39,193-204)
    assert np.abs(diff[:, 45:]).max() == 0
    clean.close(); noisy.close()


def test_humanoid_unmodeled_id(torch_mod):
    """RandomHumanoidUnmodeled-v0: masses 1..4 and dampings 6..8 frozen at 0.8x, 23-dim task
    (random_humanoid_unmodeled.py:40-53)."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_batch_step, oracle_sensitivity
    from parity_util import assert_lanes_explained
    torch = torch_mod
    n = 256
    env = rex.make("RandomHumanoidUnmodeled-v0", batch=n, autoreset=False)
    assert env.task_dim == 23 and env.dyn_index_to_name(0) == "mass4" and env.dyn_index_to_name(22) == "damp17"
    q, v, a, _ = _states(n, 5)
    nom = np.array(env.original_task)
    assert np.allclose(env.get_task().cpu().numpy(), nom[None], rtol=1e-6)
    xi = (nom * np.random.RandomState(2).uniform(.8, 1.2, (n, 23))).astype(np.float32).astype(np.float64)
    env.set_task(xi.astype(np.float32)); env.set_state(q, v)
    obs, r, d, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_batch_step("humanoid", q_, v_, a_, x_, variant=1, tolerance=0.0),
                                   [q, v, a, xi], ["obs", "reward"], trials=2)
    os_ = 1 + np.abs(ref["obs"]).max(1)
    eo = np.abs(obs.cpu().numpy() - ref["obs"]).max(1) / os_
    assert_lanes_explained(eo, sens["obs"] / os_, TOL_OBS, CAP_OBS, label="humanoid unmodeled |dobs|rel")        # every lane
    assert_lanes_explained(np.abs(r.cpu().numpy() - ref["reward"]), sens["reward"], TOL_REW, CAP_REW, label="humanoid unmodeled |dreward|")
    env.close()


def test_pile_up_states_all_solver_paths(torch_mod):
    """Humanoids lying / crouching on the floor with hinges past their limits: row counts beyond the in-LDS sweep sizes
    (> 16, > 21 -> scratch-row fallback) and beyond round 1's 64-row cap.  One env step vs the oracle, every lane."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_humanoid_step
    from random_envs_amd.specs import SPECS
    torch = torch_mod
    n = 512
    rng = np.random.RandomState(5)
    nom = np.array(SPECS["humanoid"].nominal_task)
    q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1))
    q[:, 7:] += rng.uniform(-1.2, 1.2, (n, 17)); q[:, 2] = rng.uniform(0.05, 0.6, n)
    qq = np.array([1, 0, 0, 0]) + rng.uniform(-1, 1, (n, 4)); q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    v = rng.uniform(-1, 1, (n, 23)); a = rng.uniform(-.4, .4, (n, 17)); xi = nom * rng.uniform(.9, 1.1, (n, 30))
    q, v, a, xi = [x.astype(np.float32).astype(np.float64) for x in (q, v, a, xi)]
    env = rex.make("RandomHumanoid-v0", batch=n, autoreset=False)
    env.set_task(xi.astype(np.float32)); env.set_state(q, v)
    obs, r, d, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
    ref = oracle_humanoid_step(q, v, a, xi)
    _, vv = env.get_state()
    vv = vv.cpu().numpy().astype(np.float64)
    ok = np.isfinite(ref["qvel"]).all(1)
    vs = 1 + np.abs(ref["qvel"]).max(1)
    ev = np.abs(vv - ref["qvel"]).max(1) / vs
    c = env.counters()
    # Row storage is sized to the model (MAXCON 64 / MAXEFC 192; these states reach 26 contacts / 108 rows): nothing is
    # dropped, so EVERY lane the oracle itself keeps finite has to agree -- to fp32 rounding, or as far as the oracle's own
    # conditioning explains (deep penetrations under 50 capped PGS sweeps amplify input rounding by orders of magnitude).
    assert c["overflow"] == 0, c
    assert np.isfinite(vv[ok]).all() and c["nonfinite"] <= (~ok).sum()
    from oracle_bindings import oracle_sensitivity
    from parity_util import assert_lanes_explained
    _, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_humanoid_step(q_, v_, a_, x_), [q, v, a, xi], ["qvel"], trials=2)
    assert np.median(ev[ok]) < 1e-5, np.median(ev[ok])
    assert_lanes_explained(ev[ok], (sens["qvel"] / vs)[ok], 1e-4, 5e-1, K=256.0, label="humanoid pile-ups |dqvel|rel")
    env.close()


@pytest.mark.gpu
def test_crouched_states_every_sweep_layout(torch_mod):
    """Crouching / half-fallen humanoids: 1 .. 30 constraint rows per evaluation, so every layout of the dual matrix is used --
    the square sizes, the packed triangle (17 .. 21 rows: both feet flat + joint limits) and the scratch-row fallback.  One env
    step vs the oracle, every lane."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_humanoid_step, oracle_sensitivity
    from parity_util import assert_lanes_explained
    from random_envs_amd.specs import SPECS
    torch = torch_mod
    n = 768; rng = np.random.RandomState(11)
    nom = np.array(SPECS["humanoid"].nominal_task)
    q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1))
    q[:, 7:] += rng.uniform(-0.9, 0.9, (n, 17)); q[:, 2] = rng.uniform(0.3, 1.3, n)
    qq = np.array([1, 0, 0, 0]) + rng.uniform(-.5, .5, (n, 4)); q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    v = rng.uniform(-1, 1, (n, 23)); a = rng.uniform(-.4, .4, (n, 17)); xi = nom * rng.uniform(.9, 1.1, (n, 30))
    q, v, a, xi = [x.astype(np.float32).astype(np.float64) for x in (q, v, a, xi)]
    env = rex.make("RandomHumanoid-v0", batch=n, autoreset=False)
    env.set_task(xi.astype(np.float32)); env.set_state(q, v)
    env.step(torch.as_tensor(a, dtype=torch.float32))
    ref = oracle_humanoid_step(q, v, a, xi)
    _, vv = env.get_state()
    vv = vv.cpu().numpy().astype(np.float64)
    ok = np.isfinite(ref["qvel"]).all(1)
    vs = 1 + np.abs(ref["qvel"]).max(1)
    ev = np.abs(vv - ref["qvel"]).max(1) / vs
    c = env.counters()
    assert c["overflow"] == 0 and ok.sum() > n * 0.9, c
    _, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_humanoid_step(q_, v_, a_, x_), [q, v, a, xi], ["qvel"], trials=2)
    assert np.median(ev[ok]) < 1e-5, np.median(ev[ok])
    assert_lanes_explained(ev[ok], (sens["qvel"] / vs)[ok], 1e-4, 5e-1, K=256.0, label="humanoid crouched states |dqvel|rel")
    env.close()


@pytest.mark.parametrize("fused", [1, 0])
def test_autoreset_observation_vs_oracle(torch_mod, fused):
    """a9 / SURVEY Q10 on the AUTO-reset path: a finished env restarts inside the step launch (humanoid_pair_step_kernel; the masked
    humanoid_reset_kernel launch with REX_HUM_FUSED_RESET=0).  The observation a finished lane returns is reset_model()'s: the
    new state's sim.forward() with the masses the episode that just ended had, against oracle/mjo_humanoid.c; its task is redrawn
    afterwards, the other lanes' tasks stay; data.xipos of the new state is what the next mass_center() reads."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_humanoid_reset_obs
    from parity_util import create_knobs
    torch = torch_mod
    B = 4096
    with create_knobs(REX_HUM_FUSED_RESET=fused):
        env = rex.make("RandomHumanoid-v0", batch=B, seed=13)
    lo, hi = env.get_task_search_bounds()
    env.set_dr_distribution("uniform", np.stack([lo, hi], 1).ravel().tolist()); env.set_dr_training(True)
    env.reset()
    g = torch.Generator().manual_seed(1); checked = 0
    for t in range(60):
        xi_before = env.get_task().cpu().numpy().astype(np.float64)
        obs, r, d, info = env.step(torch.rand(B, 17, generator=g) * 0.8 - 0.4)
        idx = np.where(d.cpu().numpy())[0]
        if len(idx) < 16:
            continue
        q, v = env.get_state()
        q = q.cpu().numpy().astype(np.float64)[idx]; v = v.cpu().numpy().astype(np.float64)[idx]
        assert np.abs(q - np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17)).max() <= 0.01 + 1e-6 and np.abs(v).max() <= 0.01 + 1e-7
        ref, xip = oracle_humanoid_reset_obs(q, v, xi_before[idx])
        o = obs.cpu().numpy().astype(np.float64)[idx]
        e = np.abs(o - ref).max(1) / (1 + np.abs(ref).max(1))
        assert e.max() < 2e-5, (fused, e.max())
        xi_after = env.get_task().cpu().numpy().astype(np.float64)
        keep = np.ones(B, bool); keep[idx] = False
        assert np.array_equal(xi_after[keep], xi_before[keep]) and np.abs(xi_after[idx] - xi_before[idx]).max(1).min() > 1e-3
        aux = env.get_full_state()["aux"].t().cpu().numpy()[idx]
        assert np.abs(aux - xip).max() < 1e-5
        checked += len(idx)
        if checked >= 256:
            break
    assert checked >= 64, checked
    c = env.counters(); assert c["nonfinite"] == 0 and c["overflow"] == 0
    env.close()


def test_one_lane_step_kernel_matches_the_oracle(torch_mod):
    """REX_HUM_PAIR=0: the one-env-per-lane step kernel (humanoid_step_kernel over humanoid_engine.hpp), kept behind the knob as
    the A/B partner of the pair kernel: same oracle, same per-lane gates; and the two kernels agree with each other."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_humanoid_step, oracle_sensitivity
    from parity_util import assert_lanes_explained, create_knobs
    torch = torch_mod
    n = 512
    q, v, a, xi = _states(n, 17)
    outs = []
    for pair in (0, 1):
        with create_knobs(REX_HUM_PAIR=pair):
            env = rex.make("RandomHumanoid-v0", batch=n, autoreset=False)
        env.set_task(xi.astype(np.float32)); env.set_state(q, v)
        obs, r, d, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
        outs.append((obs.cpu().numpy().astype(np.float64), r.cpu().numpy().astype(np.float64), env.counters()))
        env.close()
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_humanoid_step(q_, v_, a_, x_), [q, v, a, xi], ["obs", "reward"], trials=2)
    os_ = 1 + np.abs(ref["obs"]).max(1)
    for pair, (o, r, c) in zip((0, 1), outs):
        assert c["nonfinite"] == 0 and c["overflow"] == 0
        assert_lanes_explained(np.abs(o - ref["obs"]).max(1) / os_, sens["obs"] / os_, TOL_OBS, CAP_OBS, label="humanoid REX_HUM_PAIR=%d |dobs|rel" % pair)
        assert_lanes_explained(np.abs(r - ref["reward"]), sens["reward"], TOL_REW, CAP_REW, label="humanoid REX_HUM_PAIR=%d |dreward|" % pair)
    assert_lanes_explained(np.abs(outs[0][0] - outs[1][0]).max(1) / os_, sens["obs"] / os_, TOL_OBS, CAP_OBS, label="one-lane vs pair kernel |dobs|rel")
