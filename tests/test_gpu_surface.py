"""The rest of the drop-in surface on the GPU, each item against the oracle or a reference-stated rule:
the launch shapes rex picks past 32 768 envs at real batches of that size, device xi draws vs oracle/dr_sampler.py (two-sample KS, all four
dr_types), observation noise of the Noisy ids, per-term reward `info`, RNG-exact / time-limit-exact resume, the
offline-replay helpers for every chain, lane export for a viewer, `dt`, side-effect-free sample_tasks."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _dr_oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dr_sampler
    return dr_sampler


# ------------------------------------------------------------------------------------------------- 64-lane blocks
@pytest.mark.parametrize("kind,eid,B", [("hopper", "RandomHopper-v0", 65536), ("walker2d", "RandomWalker2d-v0", 65536), ("halfcheetah", "RandomHalfCheetah-v0", 40960),
                                        ("hopper", "RandomHopper-v0", 131072)])
def test_step_parity_at_65536_envs(torch_mod, kind, eid, B):
    """The launch shapes rex picks by itself past 32 768 envs (one lane per env in 64-lane blocks; hopper past 65 536 envs: the 256-register
    kernel, two waves per SIMD): 2 048 oracle-checked states tiled over the batch, every copy bit-identical to the first (lane position must
    not matter) and the first within tolerance."""
    import random_envs_amd as rex
    from oracle_bindings import DIMS, oracle_batch_step, oracle_sensitivity, rollout_states
    from parity_util import assert_lanes_explained
    torch = torch_mod
    n = 2048; d = DIMS[kind]
    q, v, xi = rollout_states(kind, n, steps_max=60, seed=21)
    q, v, xi = [x.astype(np.float32).astype(np.float64) for x in (q, v, xi)]
    a = np.random.RandomState(6).uniform(-1, 1, (n, d["nu"])).astype(np.float32).astype(np.float64)
    env = rex.make(eid, batch=B, autoreset=False)
    rep = B // n
    env.set_task(np.tile(xi, (rep, 1)).astype(np.float32)); env.set_state(np.tile(q, (rep, 1)), np.tile(v, (rep, 1)))
    obs, r, dn, _ = env.step(torch.as_tensor(np.tile(a, (rep, 1)), dtype=torch.float32))
    qq, vv = env.get_state()
    qq = qq.cpu().numpy(); vv = vv.cpu().numpy()
    for k in range(1, rep):
        assert np.array_equal(qq[k * n:(k + 1) * n], qq[:n]) and np.array_equal(vv[k * n:(k + 1) * n], vv[:n])
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_batch_step(kind, q_, v_, a_, x_), [q, v, a, xi], ["qpos", "qvel"])
    vs = 1 + np.abs(ref["qvel"]).max(1)
    assert_lanes_explained(np.abs(qq[:n] - ref["qpos"]).max(1), sens["qpos"], 2e-5, 5e-4, label=kind + " B=%d |dqpos|" % B)
    assert_lanes_explained(np.abs(vv[:n] - ref["qvel"]).max(1) / vs, sens["qvel"] / vs, 2e-4, 2e-2, label=kind + " B=%d |dqvel|rel" % B)
    env.close()


# ------------------------------------------------------------------------------------------------- DR draws vs the pinned sampler
def _ks2(a, b):
    from scipy.stats import ks_2samp
    return ks_2samp(a, b).statistic


@pytest.mark.parametrize("dr_type", ["uniform", "truncnorm", "gaussian", "fullgaussian"])
def test_device_xi_vs_dr_sampler_oracle(torch_mod, dr_type):
    """Two-sample KS of the device draws against draws of oracle/dr_sampler.py (bit-pinned to the reference-generated
    goldens for uniform / gaussian / fullgaussian), per task dimension; RNG bits cannot match numpy."""
    import random_envs_amd as rex
    ds = _dr_oracle()
    B, N = 65536, 3000
    env = rex.make("RandomHalfCheetah-v0", batch=B, seed=17)
    spec = env.spec; d = env.task_dim
    mean = np.array(spec.nominal_task); std = 0.15 * mean
    lower = np.array(spec.lower_bounds)
    np.random.seed(123)
    if dr_type == "uniform":
        lo, hi = 0.7 * mean, 1.4 * mean
        env.set_dr_distribution("uniform", np.stack([lo, hi], 1).ravel().tolist())
        ref = np.array([ds.sample_task("uniform", min_task=lo, max_task=hi) for _ in range(N)])
    elif dr_type == "truncnorm":
        m2 = mean.copy(); s2 = std.copy(); m2[7] = 0.05; s2[7] = 0.04        # friction close to its lower bound 0.02: redraw rule
        env.set_dr_distribution("truncnorm", np.stack([m2, s2], 1).ravel().tolist())
        ref = np.array([ds.sample_task("truncnorm", mean_task=m2, stdev_task=s2, lower_bounds=lower) for _ in range(N)])
    elif dr_type == "gaussian":
        env.set_dr_distribution("gaussian", np.stack([mean, std], 1).ravel().tolist())
        ref = np.array([ds.sample_task("gaussian", mean_task=mean, stdev_task=std) for _ in range(N)])
    else:
        m = np.full(d, 2.0); m[0] = 0.3                                      # dim 0 piles up at the clip
        A = np.random.RandomState(1).randn(d, d) * 0.2; cov = A @ A.T + 0.1 * np.eye(d)
        env.set_dr_distribution("fullgaussian", {"mean": m, "cov": cov})
        sb = env.get_task_search_bounds()
        ref = np.array([ds.sample_task("fullgaussian", mean_task=m, cov_task=cov, search_bounds=sb) for _ in range(N)])
    xi = env.sample_task().cpu().numpy().astype(np.float64)                 # [B, d], side-effect free
    crit = 1.95 * np.sqrt(1.0 / N + 1.0 / B)                                # alpha = 0.001
    stats = [_ks2(xi[:, k], ref[:, k]) for k in range(d)]
    print(dr_type, "KS per dim", np.round(stats, 4), "critical", round(crit, 4))
    assert max(stats) < crit, (dr_type, stats)
    if dr_type == "fullgaussian":                                            # correlations too
        assert np.abs(np.corrcoef(xi.T) - np.corrcoef(ref.T)).max() < 0.08
    if dr_type == "truncnorm":                                               # the clamp mass at the lower bound
        assert abs((xi[:, 7] <= lower[7] + 1e-7).mean() - (ref[:, 7] <= lower[7] + 1e-12).mean()) < 0.02
    env.close()


def test_sample_tasks_has_no_side_effects(torch_mod):
    import random_envs_amd as rex
    torch = torch_mod
    env = rex.make("RandomWalker2dUnmodeled-v0", batch=512, seed=3)
    lo, hi = env.get_task_search_bounds()
    env.set_dr_distribution("uniform", np.stack([lo, hi], 1).ravel().tolist())
    before = env.get_full_state()
    full_before = env.get_task().clone()
    t1 = env.sample_tasks(3)
    assert t1.shape == (3, 512, env.task_dim) and not torch.equal(t1[0], t1[1])
    after = env.get_full_state()
    for k in before:
        assert torch.equal(before[k], after[k]), k                            # task, counters, state untouched
    assert torch.equal(env.get_task(), full_before)
    t = t1.cpu().numpy()
    assert (t >= lo - 1e-5).all() and (t <= hi + 1e-5).all()
    env.close()


# ------------------------------------------------------------------------------------------------- observation noise
@pytest.mark.parametrize("eid,sigma", [("RandomHopperNoisy-v0", 1e-2), ("RandomWalker2dNoisy-v0", np.sqrt(1e-3))])
def test_noisy_obs_sigma(torch_mod, eid, sigma):
    """obs += sqrt(noise_level) * randn (random_hopper.py:28,108: 1e-4; random_walker2d.py:30,140: 1e-3), reset obs too."""
    import random_envs_amd as rex
    torch = torch_mod
    B = 8192
    env = rex.make(eid, batch=B, seed=2, autoreset=False)
    obs0 = env.reset().clone()
    q, v = env.get_state()
    res0 = (obs0 - torch.cat([q[:, 1:], v], 1)).cpu().numpy()
    assert abs(res0.std() - sigma) < 0.03 * sigma and abs(res0.mean()) < 0.05 * sigma
    obs, r, d, _ = env.step(torch.zeros(B, env.dims.act_dim))
    q, v = env.get_state()
    res = (obs - torch.cat([q[:, 1:], v], 1)).cpu().numpy()
    assert abs(res.std() - sigma) < 0.03 * sigma and abs(res.mean()) < 0.05 * sigma
    assert abs(np.corrcoef(res0[:, 0], res[:, 0])[0, 1]) < 0.05                # fresh noise every step
    from scipy.stats import kstest
    assert kstest(res[:, 3] / sigma, "norm").statistic < 0.02
    env.close()


# ------------------------------------------------------------------------------------------------- info / dt
def test_info_terms_and_dt(torch_mod):
    import random_envs_amd as rex
    torch = torch_mod
    env = rex.make("RandomHalfCheetah-v0", batch=256, seed=1)
    env.reset()
    assert abs(env.dt - 0.05) < 1e-7                                           # 0.01 * frame_skip 5
    a = torch.rand(256, 6) * 2 - 1
    q0, _ = env.get_state()
    obs, r, d, info = env.step(a)
    q1, _ = env.get_state()
    # random_half_cheetah.py:105-110: reward_ctrl = -0.1 * sum(a^2), reward_run = (x' - x) / dt, reward = their sum
    assert torch.allclose(info["reward_ctrl"], -0.1 * (a.cuda() ** 2).sum(1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(info["reward_run"], (q1[:, 0] - q0[:, 0]) / env.dt, rtol=1e-3, atol=1e-4)
    assert torch.allclose(info["reward_run"] + info["reward_ctrl"], r, rtol=1e-5, atol=1e-5)
    env.close()
    env = rex.make("RandomHumanoid-v0", batch=128, seed=1)
    env.reset()
    assert abs(env.dt - 0.015) < 1e-7
    a = torch.rand(128, 17) * 0.8 - 0.4
    obs, r, d, info = env.step(a)
    # random_humanoid.py:176-187
    assert torch.allclose(info["reward_quadctrl"], -0.1 * (a.cuda() ** 2).sum(1), rtol=1e-5, atol=1e-6)
    assert (info["reward_alive"] == 5).all() and (info["reward_impact"] == 0).all()
    assert torch.allclose(info["reward_linvel"] + info["reward_quadctrl"] + info["reward_alive"] + info["reward_impact"], r, atol=1e-4)
    env.close()
    assert abs(rex.make("RandomHopper-v0", batch=8).dt - 0.008) < 1e-7


# ------------------------------------------------------------------------------------------------- exact resume
@pytest.mark.parametrize("eid", ["RandomHopperNoisy-v0", "RandomHumanoid-v0"])
def test_checkpoint_resume_is_bit_exact(torch_mod, eid):
    """get_full_state / set_full_state carry qpos, qvel, xi AND the per-lane step / episode counters that key the Philox
    streams: a resumed env reproduces the original bit for bit -- observation noise, auto-resets, xi resamples and
    TimeLimit truncation included (the reference's get_sim_state returns the whole MjSimState, random_hopper.py:148-152)."""
    import random_envs_amd as rex
    torch = torch_mod
    B = 1024
    def mk():
        env = rex.make(eid, batch=B, seed=77)
        nom = torch.tensor(env.original_task)
        env.set_dr_distribution("uniform", torch.stack([0.8 * nom, 1.2 * nom], 1).flatten().tolist()); env.set_dr_training(True)
        return env
    amp = 1.0 if "Hopper" in eid else 0.4
    g = torch.Generator().manual_seed(3)
    acts = (torch.rand(60, B, 3 if "Hopper" in eid else 17, generator=g) * 2 - 1) * amp
    a = mk(); a.reset()
    for t in range(30):
        a.step(acts[t])
    snap = a.get_full_state()
    ref = [tuple(x.clone() for x in a.step(acts[t])[:3]) for t in range(30, 60)]
    assert sum(int(r[2].sum()) for r in ref) > 0                               # the window contains auto-resets
    b = mk(); b.reset()                                                        # a different history ...
    for t in range(5):
        b.step(acts[59 - t])
    b.set_full_state(snap)                                                     # ... then the snapshot
    for t in range(30, 60):
        o, r, d, _ = b.step(acts[t])
        o0, r0, d0 = ref[t - 30]
        assert torch.equal(d, d0) and torch.equal(r, r0) and torch.equal(o, o0), "step %d after resume" % t
    a.close(); b.close()


def test_time_limit_survives_resume_and_long_episodes(torch_mod):
    import random_envs_amd as rex
    torch = torch_mod
    B = 64
    env = rex.make("RandomHalfCheetahNoisy-v0", batch=B, seed=0)
    env.reset()
    st = env.get_full_state(); st["t"] = torch.full_like(st["t"], 497)
    env.set_full_state(st)
    flags = []
    for t in range(3):
        obs, r, d, info = env.step(torch.zeros(B, 6)); flags.append(bool(d.all()))
    assert flags == [False, False, True]                                       # truncation at step 500 of the RESUMED count
    env.close()
    # time_limit off: the observation-noise streams stay distinct far past step 1015 (no overlap with the next episode's)
    env = rex.make("RandomHalfCheetahNoisy-v0", batch=B, seed=0, time_limit=False, autoreset=False)
    env.reset()
    st = env.get_full_state(); st["t"] = torch.full_like(st["t"], 1200); env.set_full_state(st)
    o1 = env.step(torch.zeros(B, 6))[0].clone()
    st2 = env.get_full_state(); st2["episode"] = st2["episode"] + 1; st2["t"] = torch.full_like(st["t"], 176)
    st2["qpos"], st2["qvel"] = st["qpos"], st["qvel"]
    env.set_full_state(st2)                                                     # (ep+1, t=177): collided with (ep, 1201) before
    o2 = env.step(torch.zeros(B, 6))[0]
    assert (o1 - o2).abs().max() > 1e-4
    env.close()


# ------------------------------------------------------------------------------------------------- replay helpers (f2)
@pytest.mark.parametrize("eid", ["RandomHalfCheetah-v0", "RandomWalker2d-v0", "RandomHumanoid-v0"])
def test_replay_transitions_every_chain(torch_mod, eid):
    """get_full_mjstate (obs -> qpos with root x -- and y for the humanoid -- zeroed, random_half_cheetah.py:136-146,
    random_walker2d.py:161-171, random_humanoid.py:244-264) + set_sim_state + step under candidate xi == stepping the
    original state: the dynamics do not depend on the dropped coordinates."""
    import random_envs_amd as rex
    torch = torch_mod
    B = 512
    hum = "Humanoid" in eid
    env = rex.make(eid, batch=B, seed=9, autoreset=False)
    env.reset()
    amp = 0.4 if hum else 1.0
    g = torch.Generator().manual_seed(4)
    for t in range(10):
        obs, _, _, _ = env.step((torch.rand(B, env.dims.act_dim, generator=g) * 2 - 1) * amp)
    obs = obs.clone()
    q, v = env.get_state()
    fq, fv = env.get_full_mjstate(obs)
    skip = 2 if hum else 1
    assert torch.equal(fq[:, skip:], q[:, skip:]) and torch.equal(fv, v) and (fq[:, :skip] == 0).all()
    a = (torch.rand(B, env.dims.act_dim, generator=g) * 2 - 1) * amp
    xi = env.get_task() * (1 + 0.1 * (torch.rand(B, env.task_dim, generator=g).cuda() - 0.5))     # candidate xi, on device
    env2 = rex.make(eid, batch=B, seed=1)
    nxt, r, d = env2.replay_transitions(obs, a, xi)
    assert env2.autoreset                                                     # restored
    env.set_task(xi); env.set_state(q, v)
    ref, rr, dd, _ = env.step(a)
    if hum:
        # The dynamics do not depend on the dropped root x, y: qpos[2:], qvel and qfrc_actuator agree to rounding.  The
        # cinert / cvel blocks DO move with the absolute position once the masses are randomised -- MuJoCo refers them to
        # the subtree COM, which divides sum(m_i x_i) by the compile-time subtree mass that set_task never refreshes
        # (SURVEY Q4); the oracle restates the same quirk (oracle/mjo_core.c::mjo_com_quantities).
        inv = list(range(45)) + list(range(269, 292))
        err = ((nxt[:, inv] - ref[:, inv]).abs() / (1 + ref[:, inv].abs())).max().item()
        assert err < 2e-4, err
        assert (r - rr).abs().max().item() < 2e-2
        zz = ref[:, 0]                                                          # done = z < 1 or z > 2: may differ only at a threshold
        far = ((zz - 1.0).abs() > 1e-4) & ((zz - 2.0).abs() > 1e-4)
        assert torch.equal(d[far], dd[far])
    else:     # the planar kernels integrate from x = 0 anyway: bit-identical
        assert torch.equal(nxt, ref) and torch.equal(r, rr) and torch.equal(d, dd)
    env.close(); env2.close()


def test_fused_replay_matches_the_three_call_path_and_leaves_the_env_alone(torch_mod):
    """rex_replay (hopper, half-cheetah): one launch from the caller's buffers == set_task + set_sim_state + step bit for bit,
    and the handle's own state, task and counters are untouched."""
    import random_envs_amd as rex
    torch = torch_mod
    for eid in ("RandomHopper-v0", "RandomHalfCheetah-v0"):
        B = 2048
        env = rex.make(eid, batch=B, seed=11)
        nom = torch.tensor(env.original_task)
        env.set_dr_distribution("uniform", torch.stack([0.8 * nom, 1.2 * nom], 1).flatten().tolist()); env.set_dr_training(True)
        env.reset()
        g = torch.Generator().manual_seed(2)
        for _ in range(12):
            obs, _, _, _ = env.step(torch.rand(B, env.dims.act_dim, generator=g) * 2 - 1)
        obs = obs.clone(); a = torch.rand(B, env.dims.act_dim, generator=g) * 2 - 1
        xi = env.sample_task()
        before = env.get_full_state()
        nxt, r, d = env.replay_transitions(obs, a, xi)
        after = env.get_full_state()
        for k in before:
            assert torch.equal(before[k], after[k]), (eid, k)
        ref_env = rex.make(eid, batch=B, seed=5, autoreset=False)
        q, v = ref_env.get_full_mjstate(obs)
        ref_env.set_task(xi); ref_env.set_state(q, v)
        o2, r2, d2, _ = ref_env.step(a)
        assert torch.equal(nxt, o2) and torch.equal(r, r2) and torch.equal(d, d2), eid
        env.close(); ref_env.close()


# ------------------------------------------------------------------------------------------------- lane export / adapter
def test_export_lane_and_sb3_adapter(torch_mod):
    import random_envs_amd as rex
    from random_envs_amd.sb3_adapter import SB3VecEnvAdapter
    B = 96
    env = rex.make("RandomWalker2dUnmodeled-v0", batch=B, seed=5)
    lo, hi = env.get_task_search_bounds()
    env.set_dr_distribution("uniform", np.stack([lo, hi], 1).ravel().tolist()); env.set_dr_training(True); env.reset()
    q, v = env.get_state(); xi = env.get_task()
    for k in (0, 37, B - 1):
        e = env.export_lane(k)
        assert np.array_equal(e["qpos"], q[k].cpu().numpy()) and np.array_equal(e["qvel"], v[k].cpu().numpy())
        assert np.array_equal(e["task"], xi[k].cpu().numpy()) and len(e["task_names"]) == env.task_dim
    with pytest.raises(ValueError):
        env.export_lane(B)
    ad = SB3VecEnvAdapter(env)
    assert len(ad.get_attr("task_dim")) == B and ad.get_attr("task_dim", indices=[1, 2]) == [env.task_dim] * 2
    tasks = ad.env_method("get_task")
    assert len(tasks) == B and tasks[3].shape == (env.task_dim,)
    assert ad.env_is_wrapped(object) == [False] * B
    obs, rew, done, infos = ad.step(np.zeros((B, 6), dtype=np.float32))
    assert obs.shape == (B, 17) and len(infos) == B
    ad.close()
