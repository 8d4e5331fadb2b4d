"""C-ABI / VecRandomEnv behaviour: ragged and tiny batches, masks, error paths, determinism,
state helpers, the SB3-style adapter."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("batch", [1, 7, 100, 4097])
def test_ragged_batches_match_full_lanes(torch_mod, batch, fast):
    """lane i of a ragged batch computes what lane i of a big batch computes (global-index RNG, no cross-lane data flow):
    covers the partial last wavefront.  With the solver knob REX_FAST=0 the arithmetic of a lane does not depend on its
    wave at all -> bit-identical.  By default the WAVE picks the solver instantiation (feet-only straight-line / general /
    skip of the qacc_smooth factorisation on warm starts): every choice converges to the same unique minimiser, so lanes
    agree to fp32 rounding instead, and everything the RNG decides (reset states, xi) stays bit-identical."""
    import random_envs_amd as rex
    from parity_util import create_knobs
    torch = torch_mod
    with create_knobs(REX_FAST=fast):
        big = rex.make("RandomHopper-v0", batch=8192, seed=7, autoreset=False)
        small = rex.make("RandomHopper-v0", batch=batch, seed=7, autoreset=False)
    for e in (big, small):
        e.set_dr_distribution("uniform", [3.0, 4.0, 3.5, 4.5, 2.2, 3.2, 4.5, 5.5]); e.set_dr_training(True)
        ob0 = e.reset().clone()
        if e is big:
            ref0 = ob0
    assert torch.equal(ref0[:batch], ob0) and torch.equal(big.get_task()[:batch], small.get_task())
    a = torch.rand(8192, 3, generator=torch.Generator().manual_seed(0)) * 2 - 1
    for _ in range(3):
        ob, rb, db, _ = big.step(a); os_, rs, ds, _ = small.step(a[:batch])
        if fast:
            assert torch.allclose(ob[:batch], os_, rtol=2e-5, atol=2e-6) and torch.allclose(rb[:batch], rs, rtol=1e-4, atol=1e-4)
            assert torch.equal(db[:batch], ds)
        else:
            assert torch.equal(ob[:batch], os_) and torch.equal(rb[:batch], rs) and torch.equal(db[:batch], ds)
    big.close(); small.close()


def test_error_paths(torch_mod):
    import random_envs_amd as rex
    from random_envs_amd import _native
    L = _native.lib()
    h = ctypes.c_void_p()
    assert L.rex_create(99, 0, 16, 0, 0, 0, ctypes.byref(h)) == -1 and b"unknown env kind" in L.rex_last_error()
    assert L.rex_create(1, 0, 0, 0, 0, 0, ctypes.byref(h)) == -1 and b"batch" in L.rex_last_error()
    assert L.rex_create(0, 1, 16, 0, 0, 0, ctypes.byref(h)) == -1            # CartPole has no Unmodeled id
    with pytest.raises(KeyError):
        rex.make("RandomNope-v0")
    env = rex.make("RandomHopper-v0", batch=8)
    with pytest.raises(Exception, match="Unknown dr_type"):
        env.set_dr_distribution("bogus", [])
    with pytest.raises(ValueError):
        env.set_random_task()                                                # random_env.py:201
    fp = ctypes.POINTER(ctypes.c_float)
    bad = (ctypes.c_float * 3)(1, 2, 3)
    assert L.rex_set_dr(env._h, 1, ctypes.cast(bad, fp), 3, None) == -1 and b"expected 8 params" in L.rex_last_error()
    assert L.rex_step(env._h, None, None, None, None, None, None, None) == -1
    env.close()


def test_masked_reset_and_random_task(torch_mod):
    import random_envs_amd as rex
    torch = torch_mod
    B = 256
    env = rex.make("RandomWalker2d-v0", batch=B, seed=3, autoreset=False)
    nom = np.array(env.original_task)
    env.set_dr_distribution("uniform", np.stack([0.9 * nom, 1.1 * nom], 1).ravel().tolist()); env.set_dr_training(True)
    env.reset()
    for _ in range(5):
        env.step(torch.zeros(B, 6))
    q0, v0 = [x.clone() for x in env.get_state()]; xi0 = env.get_task().clone()
    mask = torch.zeros(B, dtype=torch.uint8); mask[::4] = 1
    env.reset(mask)
    q1, v1 = env.get_state(); xi1 = env.get_task()
    m = mask.bool().cuda()
    assert torch.equal(q1[~m], q0[~m]) and torch.equal(xi1[~m], xi0[~m])      # untouched lanes
    assert (q1[m][:, 1] - 1.25).abs().max() <= 0.005 + 1e-6 and (xi1[m] != xi0[m]).any(1).all()
    env.set_random_task(mask)                                                 # xi only, state untouched
    q2, _ = env.get_state()
    assert torch.equal(q2, q1) and (env.get_task()[m] != xi1[m]).any(1).all() and torch.equal(env.get_task()[~m], xi1[~m])
    obs = torch.empty(17, B, device="cuda")
    _native = __import__("random_envs_amd")._native
    _native.check(_native.lib().rex_get_obs(env._h, ctypes.c_void_p(obs.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.equal(obs.t(), torch.cat([q2[:, 1:], env.get_state()[1]], 1))
    env.close()


def test_seed_determinism_and_reseed(torch_mod):
    import random_envs_amd as rex
    def first_obs(seed, reseed=None):
        env = rex.make("RandomHalfCheetah-v0", batch=64, seed=seed)
        if reseed is not None:
            assert env.seed(reseed) == [reseed]
        o = env.reset().clone(); env.close(); return o
    import torch
    assert torch.equal(first_obs(1), first_obs(1)) and not torch.equal(first_obs(1), first_obs(2))
    assert torch.equal(first_obs(1, reseed=2), first_obs(2))


def test_state_helpers_and_replay(torch_mod):
    """get_full_mjstate / replay_transitions: one logged transition x many candidate xi."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_batch_step
    torch = torch_mod
    B = 512
    env = rex.make("RandomHopper-v0", batch=B, seed=0, autoreset=False)
    env.reset()
    for _ in range(10):
        env.step(torch.rand(B, 3) * 2 - 1)
    obs = env.get_state(); q, v = obs
    o = torch.cat([q[:, 1:], v], 1)
    fq, fv = env.get_full_mjstate(o)
    assert torch.equal(fq[:, 1:], q[:, 1:]) and (fq[:, 0] == 0).all() and torch.equal(fv, v)     # random_hopper.py:128-136
    sq, sv = env.get_sim_state(); env.set_sim_state((sq, sv))
    # the same transition (row 0) replayed under 512 different masses
    nom = np.array(env.original_task)
    xi = (nom * np.random.RandomState(0).uniform(.7, 1.3, (B, 4))).astype(np.float32)
    a = torch.rand(1, 3) * 2 - 1
    nxt, r, d = env.replay_transitions(o[:1].expand(B, -1), a.expand(B, -1), xi)
    from oracle_bindings import oracle_sensitivity
    from parity_util import assert_lanes_explained
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_batch_step("hopper", q_, v_, a_, x_),
                                   [fq[:1].expand(B, -1).cpu().numpy().astype(np.float64), fv[:1].expand(B, -1).cpu().numpy().astype(np.float64),
                                    a.expand(B, -1).numpy().astype(np.float64), xi.astype(np.float64)], ["obs"])
    os_ = 1 + np.abs(ref["obs"]).max(1)
    err = np.abs(nxt.cpu().numpy() - ref["obs"]).max(1) / os_
    assert_lanes_explained(err, sens["obs"] / os_, 2e-4, 2e-2, label="hopper replay |dobs|rel")      # every lane
    assert nxt.std(0).max() > 1e-4          # different xi -> different next states
    env.close()


def test_sb3_style_adapter(torch_mod):
    import random_envs_amd as rex
    from random_envs_amd.sb3_adapter import SB3VecEnvAdapter
    venv = SB3VecEnvAdapter(rex.make("RandomHopper-v0", batch=128, seed=0))
    obs = venv.reset()
    assert obs.shape == (128, 11) and obs.dtype == np.float32
    seen = 0
    for _ in range(60):
        obs, rew, dones, infos = venv.step(np.random.uniform(-1, 1, (128, 3)).astype(np.float32))
        for i in np.nonzero(dones)[0]:
            assert infos[i]["terminal_observation"].shape == (11,) and "TimeLimit.truncated" in infos[i]
            seen += 1
    assert seen > 0 and venv.get_attr("task_dim") == [4] * 128 and venv.get_attr("task_dim", indices=[0]) == [4]
    venv.close()


def test_launch_shape_follows_the_batch(torch_mod):
    """rex_get_launch_shape: what rex_create picks from the per-GPU batch and the GPU's SIMD count -- planar chains two lanes
    per env while 32 envs x SIMDs hold the batch, one lane per env in 64-lane blocks past that, the hopper past 64 envs x SIMDs on the rolled
    two-waves-per-SIMD kernel; the humanoid on two lanes per env everywhere; knobs override."""
    import random_envs_amd as rex
    from parity_util import create_knobs
    simds = 4 * torch_mod.cuda.get_device_properties(0).multi_processor_count
    for eid, hopper in (("RandomHopper-v0", True), ("RandomWalker2d-v0", False), ("RandomHalfCheetah-v0", False)):
        n16, n32 = (64, 64) if hopper else (16, 32)     # walker2d / half-cheetah: narrower blocks from 8 envs per SIMD up while they all get a SIMD (pair_lanes_for)
        for B, want in ((1, (64, True, False)), (8 * simds - 1, (64, True, False)), (8 * simds, (n16, True, False)), (8 * simds + 1, (n32, True, False)),
                        (16 * simds, (n32, True, False)), (16 * simds + 1, (64, True, False)),
                        (32 * simds, (64, True, False)), (32 * simds + 1, (64, False, False)),
                        (64 * simds, (64, False, False)), (64 * simds + 1, (64, False, hopper))):
            env = rex.make(eid, batch=B, autoreset=False)
            sh = env.launch_shape()
            assert (sh["lanes"], sh["pair"], sh["rolled"]) == want and not sh["hum_pair"], (eid, B, sh)
            kind = {"RandomHopper-v0": "hopper", "RandomWalker2d-v0": "walker2d", "RandomHalfCheetah-v0": "halfcheetah"}[eid]
            from random_envs_amd import sharding
            assert sharding.shape_for_batch(kind, B, simds) == sh, (eid, B)     # the host restatement pin_global_shape relies on
            env.close()
    env = rex.make("RandomHumanoid-v0", batch=64, autoreset=False)
    assert env.launch_shape()["hum_pair"] and not env.launch_shape()["pair"]
    env.close()
    with create_knobs(REX_PAIR=0, REX_ROLLED=1, REX_LANES=64):
        env = rex.make("RandomHopper-v0", batch=256, autoreset=False)
    assert env.launch_shape() == dict(lanes=64, pair=False, rolled=True, hum_pair=False)
    env.close()
    with create_knobs(REX_FAST=0):   # the pair split lives in the feet-only instantiation
        env = rex.make("RandomHopper-v0", batch=256, autoreset=False)
    assert not env.launch_shape()["pair"]
    env.close()
    env = rex.make("RandomCartPole-v0", batch=256, autoreset=False)
    assert env.launch_shape() == dict(lanes=32, pair=False, rolled=False, hum_pair=False)
    env.close()


def test_stray_knobs_are_refused_and_shapes_are_pinned_through_the_abi(torch_mod):
    """A tuning variable set WITHOUT REX_ALLOW_TUNING=1 makes rex_create fail loudly (REX_ERR_STATE naming the variable) instead of silently
    changing the solver; with it the knob is honoured.  rex_set_launch_shape pins the shape per handle: round trip, -1 keeps a field, shapes an env
    kind has no kernel for are REX_ERR_ARG."""
    import random_envs_amd as rex
    from random_envs_amd import _native
    from parity_util import create_knobs
    for k, v in (("REX_FAST", "0"), ("REX_CORR", "0"), ("REX_PAIR", "0"), ("REX_LANES", "64"), ("REX_HUM_ITERS", "5"), ("REX_DIAG_NOCONTACT", "1")):
        os.environ.pop("REX_ALLOW_TUNING", None)
        os.environ[k] = v
        try:
            with pytest.raises(_native.RexError, match=k):
                rex.make("RandomHopper-v0", batch=64, autoreset=False)
        finally:
            os.environ.pop(k)
    with create_knobs(REX_HUM_ITERS=5):   # allowed to tune, but the product build has no such knob: still refused
        with pytest.raises(_native.RexError, match="REX_TUNING"):
            rex.make("RandomHumanoid-v0", batch=64, autoreset=False)
    env = rex.make("RandomHopper-v0", batch=256, autoreset=False)
    assert env.launch_shape() == dict(lanes=64, pair=True, rolled=False, hum_pair=False)
    assert env.set_launch_shape(lanes=64, pair=False) == dict(lanes=64, pair=False, rolled=False, hum_pair=False)
    assert env.set_launch_shape(rolled=True) == dict(lanes=64, pair=False, rolled=True, hum_pair=False)
    assert env.set_launch_shape() == dict(lanes=64, pair=False, rolled=True, hum_pair=False)       # all -1: nothing changes
    for bad in (dict(lanes=48), dict(pair=True), dict(hum_pair=True)):                              # pair + rolled / not a humanoid / lanes
        with pytest.raises(ValueError):
            env.set_launch_shape(**bad)
    env.close()
    env = rex.make("RandomWalker2d-v0", batch=64, autoreset=False)
    with pytest.raises(ValueError):
        env.set_launch_shape(rolled=True)
    env.close()


def test_pinned_shards_reproduce_a_single_gpu_batch_past_the_pair_limit(torch_mod):
    """rex_create picks the launch shape from the PER-GPU batch, so 65 536 envs on one GPU (one lane per env) and the same envs split
    over two GPUs (two lanes per env) round differently.  With sharding.pin_global_shape every shard runs the global batch's shape and the split
    reproduces the single-GPU trajectories bit for bit; un-pinned they agree to fp32 rounding (and then diverge chaotically)."""
    import random_envs_amd as rex
    from random_envs_amd import sharding
    torch = torch_mod
    simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    B, steps = 64 * simds, 6
    assert sharding.shape_for_batch("hopper", B, simds) == dict(lanes=64, pair=False, rolled=False, hum_pair=False)
    g = torch.Generator().manual_seed(3)
    acts = (torch.rand(steps, B, 3, generator=g) * 2 - 1)

    def run(off, n, pin):
        env = rex.make("RandomHopper-v0", batch=n, seed=7, env_offset=off)
        if pin:
            sharding.pin_global_shape(env, B)
        shape = env.launch_shape()
        env.set_dr_distribution("uniform", [3.0, 4.0, 3.5, 4.5, 2.2, 3.2, 4.5, 5.5]); env.set_dr_training(True)
        env.reset()
        outs = [tuple(x.clone() for x in env.step(acts[t, off:off + n].cuda())[:3]) for t in range(steps)]
        env.close()
        return shape, outs
    shape_full, full = run(0, B, False)
    off, n = sharding.shard_strong(B, 1, 2)
    shape_pin, pinned = run(off, n, True)
    shape_free, free = run(off, n, False)
    assert shape_full == shape_pin == dict(lanes=64, pair=False, rolled=False, hum_pair=False) and shape_free["pair"]
    for (o, r, d), (o1, r1, d1) in zip(full, pinned):
        assert torch.equal(o[off:off + n], o1) and torch.equal(r[off:off + n], r1) and torch.equal(d[off:off + n], d1)
    o, o2 = full[0][0][off:off + n], free[0][0]
    assert not torch.equal(o, o2) and (o - o2).abs().max() < 1e-3   # another kernel: equal to rounding after one step, not bit for bit


def test_narrow_blocks_in_xcd_order_cover_every_env(torch_mod):
    """The 32- / 16-lane two-lanes-per-env blocks take their env group in XCD-transposed order (planar_step_kernel: block b works on group
    (b % 8) * (blocks / 8) + b / 8 when the block count is a multiple of 8).  Every env must be stepped exactly once -- also with a ragged last
    group, also when the block count is no multiple of 8 and the order stays the launch order -- and agree with the same envs on 64-lane blocks
    to fp32 rounding (another grouping of envs into waves: not bit for bit)."""
    import random_envs_amd as rex
    torch = torch_mod
    simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    for eid, nact in (("RandomWalker2d-v0", 6), ("RandomHalfCheetah-v0", 6)):
        # 16 * 8 m - 5 envs: 32-lane blocks, 8 m of them, the last one ragged;  8 * simds: 16-lane blocks;  8 * simds + 9: 32-lane blocks, count odd
        for B, lanes in ((16 * (simds * 3 // 4) - 5, 32), (8 * simds, 16), (8 * simds + 9, 32)):
            g = torch.Generator().manual_seed(11)
            acts = (torch.rand(2, B, nact, generator=g) * 2 - 1).cuda()
            outs = []
            for pin in (False, True):
                env = rex.make(eid, batch=B, seed=5)
                if pin:
                    env.set_launch_shape(lanes=64)
                assert env.launch_shape()["pair"] and env.launch_shape()["lanes"] == (64 if pin else lanes), (eid, B, env.launch_shape())
                env.reset()
                o = [env.step(acts[t])[0].clone() for t in range(2)]
                q, v = env.get_state()
                assert env.counters()["nonfinite"] == 0
                outs.append((o, q.clone(), v.clone()))
                env.close()
            (o_n, q_n, v_n), (o_w, q_w, v_w) = outs
            assert torch.isfinite(o_n[1]).all()
            assert (o_n[0] - o_w[0]).abs().max() < 2e-3 and (q_n - q_w).abs().max() < 2e-3, (eid, B, float((o_n[0] - o_w[0]).abs().max()))
            assert (v_n - v_w).abs().max() < 2e-2, (eid, B, float((v_n - v_w).abs().max()))   # (an env left out of a step would be off by g * dt = 0.08)
            assert not torch.equal(q_n[-8:], torch.zeros_like(q_n[-8:]))      # (the ragged tail was stepped too: its state left the reset noise)
