"""BASELINE.json's configurations AS the configurations, every lane against the oracle:
C5 humanoid shard (xi uniform over the search bounds, 32 768 envs -- and once 65 536: the 64-lane / > 64 KB dynamic-LDS
shape rex picks by itself), C4 walker2d stepped from truncnorm-drawn xi (8 192 lanes, the device's own draws), C3
HalfCheetahNoisy with friction U(0.3, 0.5) at 16 384; rex_replay for every chain and the Unmodeled ids; bit-reproducible
index sharding; `bench.py --gpus N` starting its own ranks and the `--config` lines."""
import json
import os
import subprocess
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


def _f32(*xs):
    return [np.asarray(x).astype(np.float32).astype(np.float64) for x in xs]


# ------------------------------------------------------------------------------------------------- C5: humanoid shard
@pytest.mark.parametrize("B", [32768, 65536])
def test_humanoid_config5_shard_every_lane(torch_mod, B):
    """1 024 oracle-checked states (xi uniform over the search bounds random_humanoid.py:72-105, random joint angles and
    heights between fallen and standing) tiled over the C5 shard: every tiled copy bit-identical to the first (lane position
    must not matter), the first copy within the stated tolerance in EVERY lane, a second run bit-identical, counters 0."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_humanoid_step, oracle_sensitivity
    from parity_util import assert_done_explained, assert_lanes_explained
    from random_envs_amd.specs import SPECS
    torch = torch_mod
    n = 1024; rng = np.random.RandomState(17)
    sb = np.array(SPECS["humanoid"].search_bounds)
    xi = rng.uniform(sb[:, 0], sb[:, 1], (n, 30))
    q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1)) + rng.uniform(-.01, .01, (n, 24))
    q[:, 7:] += rng.uniform(-.4, .4, (n, 17)); q[:, 2] = rng.uniform(0.9, 1.45, n)
    v = rng.uniform(-1, 1, (n, 23)); a = rng.uniform(-.4, .4, (n, 17))
    q, v, a, xi = _f32(q, v, a, xi)
    rep = B // n

    def run():
        env = rex.make("RandomHumanoid-v0", batch=B, autoreset=False)
        env.set_task(np.tile(xi, (rep, 1)).astype(np.float32)); env.set_state(np.tile(q, (rep, 1)), np.tile(v, (rep, 1)))
        obs, r, d, _ = env.step(torch.as_tensor(np.tile(a, (rep, 1)), dtype=torch.float32))
        qq, vv = env.get_state()
        out = [z.clone() for z in (obs, r, d, qq, vv)]
        c = env.counters(); env.close()
        return out, c
    (obs, r, d, qq, vv), c = run()
    assert c["nonfinite"] == 0 and c["overflow"] == 0, c
    for k in range(1, rep):
        s = slice(k * n, (k + 1) * n)
        assert torch.equal(obs[s], obs[:n]) and torch.equal(vv[s], vv[:n]) and torch.equal(r[s], r[:n]), k
    (obs2, r2, d2, qq2, vv2), _ = run()
    assert torch.equal(obs, obs2) and torch.equal(r, r2) and torch.equal(d, d2) and torch.equal(vv, vv2)
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_humanoid_step(q_, v_, a_, x_), [q, v, a, xi],
                                   ["obs", "qvel", "reward"], trials=2)
    o = obs[:n].cpu().numpy().astype(np.float64)
    os_ = 1 + np.abs(ref["obs"]).max(1); vs = 1 + np.abs(ref["qvel"]).max(1)
    tag = "humanoid C5 B=%d" % B
    assert_lanes_explained(np.abs(o - ref["obs"]).max(1) / os_, sens["obs"] / os_, 2e-4, 2e-2, label=tag + " |dobs|rel")
    assert_lanes_explained(np.abs(vv[:n].cpu().numpy() - ref["qvel"]).max(1) / vs, sens["qvel"] / vs, 5e-4, 5e-2, label=tag + " |dqvel|rel")
    assert_lanes_explained(np.abs(r[:n].cpu().numpy() - ref["reward"]), sens["reward"], 2e-3, 2e-1, label=tag + " |dreward|")
    z = ref["qpos"][:, 2]
    assert_done_explained(d[:n].cpu().numpy(), ref["done"], np.minimum(np.abs(z - 1.0), np.abs(z - 2.0)), 2e-5, label=tag)


# ------------------------------------------------------------------------------------------------- C4: walker2d, truncnorm xi
def test_walker2d_config4_truncnorm_step_vs_oracle(torch_mod):
    """C4's shard (8 192 envs): xi drawn ON THE DEVICE from the truncnorm distribution (mean nominal, std 10 %), lengths
    included -- every env has its own compiled geometry --, 25 steps of auto-reset rollout, then one step of every lane
    against the oracle on the device's own (qpos, qvel, xi)."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_batch_step, oracle_sensitivity
    from parity_util import assert_done_explained, assert_lanes_explained
    from random_envs_amd.specs import SPECS
    torch = torch_mod
    B = 8192
    mean = np.array(SPECS["walker2d"].nominal_task)
    env = rex.make("RandomWalker2d-v0", batch=B, seed=4)
    env.set_dr_distribution("truncnorm", np.stack([mean, 0.1 * mean], 1).ravel().tolist())
    env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(3)
    for _ in range(25):
        env.step(torch.rand(B, 6, generator=g) * 2 - 1)
    q, v = env.get_state(); xi = env.get_task()
    q, v, xi = [z.cpu().double().numpy() for z in (q, v, xi)]
    z = (xi - mean) / (0.1 * mean)
    assert np.abs(z).max() <= 2 + 1e-4 and z[:, 7:11].std() > 0.5          # lengths really vary per env
    a = (torch.rand(B, 6, generator=g) * 2 - 1)
    env.autoreset = False; env._push_flags()
    obs, r, d, _ = env.step(a)
    qq, vv = env.get_state()
    a64 = a.double().numpy()
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_batch_step("walker2d", q_, v_, a_, x_), [q, v, a64, xi],
                                   ["qpos", "qvel", "reward"], trials=2)
    vs = 1 + np.abs(ref["qvel"]).max(1)
    assert_lanes_explained(np.abs(qq.cpu().numpy() - ref["qpos"]).max(1), sens["qpos"], 2e-5, 5e-4, label="walker2d C4 |dqpos|")
    assert_lanes_explained(np.abs(vv.cpu().numpy() - ref["qvel"]).max(1) / vs, sens["qvel"] / vs, 2e-4, 2e-2, label="walker2d C4 |dqvel|rel")
    assert_lanes_explained(np.abs(r.cpu().numpy() - ref["reward"]), sens["reward"], 5e-3, 1e-1, label="walker2d C4 |dreward|")
    zz, th = ref["qpos"][:, 1], ref["qpos"][:, 2]
    margin = np.minimum.reduce([np.abs(zz - 0.8), np.abs(zz - 2.0), np.abs(th - 1.0), np.abs(th + 1.0)])
    assert_done_explained(d.cpu().numpy(), ref["done"], margin, 2e-5, label="walker2d C4")
    assert env.counters()["nonfinite"] == 0
    env.close()


# ------------------------------------------------------------------------------------------------- C3: HalfCheetahNoisy, friction
def test_halfcheetah_noisy_config3_vs_oracle(torch_mod):
    """C3: RandomHalfCheetahNoisy-v0 at 16 384 envs, masses nominal +-20 %, friction U(0.3, 0.5): the state the step leaves
    (= the clean part of the observation) against the oracle in every lane, and the observation noise on top of it:
    N(0, 1e-4) per component (random_half_cheetah.py:30,112-121)."""
    import random_envs_amd as rex
    from oracle_bindings import oracle_batch_step, oracle_sensitivity
    from parity_util import assert_lanes_explained
    from random_envs_amd.specs import SPECS
    torch = torch_mod
    B = 16384
    nom = np.array(SPECS["halfcheetah"].nominal_task)
    lo, hi = 0.8 * nom, 1.2 * nom; lo[7], hi[7] = 0.3, 0.5
    env = rex.make("RandomHalfCheetahNoisy-v0", batch=B, seed=8)
    env.set_dr_distribution("uniform", np.stack([lo, hi], 1).ravel().tolist())
    env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(6)
    for _ in range(20):
        env.step(torch.rand(B, 6, generator=g) * 2 - 1)
    q, v = env.get_state(); xi = env.get_task()
    q, v, xi = [z.cpu().double().numpy() for z in (q, v, xi)]
    assert xi[:, 7].min() >= 0.3 - 1e-6 and xi[:, 7].max() <= 0.5 + 1e-6
    a = torch.rand(B, 6, generator=g) * 2 - 1
    obs, r, d, info = env.step(a)
    qq, vv = env.get_state()
    ref, sens = oracle_sensitivity(lambda q_, v_, a_, x_: oracle_batch_step("halfcheetah", q_, v_, a_, x_), [q, v, a.double().numpy(), xi],
                                   ["qpos", "qvel", "reward"], trials=2)
    vs = 1 + np.abs(ref["qvel"]).max(1)
    assert_lanes_explained(np.abs(qq.cpu().numpy() - ref["qpos"]).max(1), sens["qpos"], 2e-5, 5e-4, label="halfcheetah C3 |dqpos|")
    assert_lanes_explained(np.abs(vv.cpu().numpy() - ref["qvel"]).max(1) / vs, sens["qvel"] / vs, 2e-4, 2e-2, label="halfcheetah C3 |dqvel|rel")
    assert_lanes_explained(np.abs(r.cpu().numpy() - ref["reward"]), sens["reward"], 5e-3, 1e-1, label="halfcheetah C3 |dreward|")
    clean = torch.cat([qq[:, 1:], vv], 1)
    noise = (obs - clean).cpu().numpy()
    assert abs(noise.std() - 0.01) < 2e-4 and abs(noise.mean()) < 1e-4 and np.abs(noise.std(0) - 0.01).max() < 5e-4
    assert not d.any() and env.counters()["nonfinite"] == 0
    env.close()


# ------------------------------------------------------------------------------------------------- rex_replay: every chain / id
@pytest.mark.parametrize("eid", ["RandomWalker2d-v0", "RandomHumanoid-v0", "RandomHopperUnmodeled-v0", "RandomHalfCheetahUnmodeled-v0",
                                 "RandomWalker2dUnmodeled-v0", "RandomHumanoidUnmodeled-v0"])
def test_rex_replay_matches_the_three_call_path_and_leaves_the_env_alone(torch_mod, eid):
    """rex_replay from the caller's buffers == set_task + set_sim_state + step on a second env, bit for bit, and the
    handle's own state, task, step / episode counters, aux rows and diagnostic counters are untouched
    (random_walker2d.py:161-185, random_humanoid.py:244-270 and the Unmodeled task files)."""
    import random_envs_amd as rex
    torch = torch_mod
    B = 1024
    hum = "Humanoid" in eid
    amp = 0.4 if hum else 1.0
    env = rex.make(eid, batch=B, seed=11)
    lo, hi = env.get_task_search_bounds()
    nom = np.array(env.original_task)
    env.set_dr_distribution("uniform", np.stack([np.maximum(lo, 0.8 * nom), np.minimum(hi, 1.2 * nom)], 1).ravel().tolist())
    env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(2)
    for _ in range(12):
        obs, _, _, _ = env.step((torch.rand(B, env.dims.act_dim, generator=g) * 2 - 1) * amp)
    obs = obs.clone(); a = (torch.rand(B, env.dims.act_dim, generator=g) * 2 - 1) * amp
    xi = env.sample_task().clone()
    before = env.get_full_state(); c0 = env.counters()
    nxt, r, d = env.replay_transitions(obs, a, xi)
    after = env.get_full_state(); c1 = env.counters()
    for k in before:
        assert torch.equal(before[k], after[k]), (eid, k)
    assert c0 == c1
    ref_env = rex.make(eid, batch=B, seed=5, autoreset=False)
    o2, r2, d2 = ref_env.replay_three_call(obs, a, xi)
    assert torch.equal(nxt, o2) and torch.equal(r, r2) and torch.equal(d, d2), eid
    assert torch.isfinite(nxt).all() and nxt.std(0).max() > 1e-4
    env.close(); ref_env.close()


# ------------------------------------------------------------------------------------------------- sharding reproducibility
@pytest.mark.parametrize("knobs,split", [(dict(REX_FAST=0), 1000), (dict(), 992)])
def test_two_shards_reproduce_the_single_gpu_run_bit_for_bit(torch_mod, knobs, split):
    """Index sharding must not change a trajectory: with REX_FAST=0 a lane's arithmetic does not depend on its
    wave at all, so ANY split reproduces the one-shard run; with the default knobs (the WAVE picks the solver instantiation)
    a split on a whole wavefront (32 envs -- what sharding.shard_strong produces) does."""
    import random_envs_amd as rex
    from parity_util import create_knobs
    torch = torch_mod
    B, steps = 2048, 40

    def run(off, n):
        with create_knobs(**knobs):
            env = rex.make("RandomHopper-v0", batch=n, seed=42, env_offset=off)
        env.set_dr_distribution("uniform", [3.0, 4.0, 3.5, 4.5, 2.2, 3.2, 4.5, 5.5]); env.set_dr_training(True)
        env.reset()
        g = torch.Generator().manual_seed(1)
        acts = (torch.rand(steps, B, 3, generator=g) * 2 - 1)[:, off:off + n]
        outs = []
        for t in range(steps):
            o, r, d, _ = env.step(acts[t]); outs.append((o.clone(), r.clone(), d.clone()))
        env.close()
        return outs
    full = run(0, B); lo = run(0, split); hi = run(split, B - split)
    for (o, r, d), (o1, r1, d1), (o2, r2, d2) in zip(full, lo, hi):
        assert torch.equal(o[:split], o1) and torch.equal(o[split:], o2) and torch.equal(r[:split], r1) and torch.equal(r[split:], r2)
        assert torch.equal(d[:split], d1) and torch.equal(d[split:], d2)


# ------------------------------------------------------------------------------------------------- bench.py: self-launch, configs
def _bench(*extra, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_n_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` WITHOUT torch.distributed.run: the parent spawns the two ranks before touching the GPU
    and relays rank 0's line; n_gpus = the ranks the process group saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    d = _bench("--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "24", "--warmup", "4", "--batch", "4096",
               "--no-cpu-baseline", "--counter-every", "8", env=env)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["counter_reductions"] == 4
    assert abs(d["value"] - 8192 * 24 / (d["ms_per_step"] * 24 / 1e3)) < 1e-6 * d["value"]


@pytest.mark.parametrize("cfg,eid,B,kernel,nbytes", [
    ("C1", "RandomCartPole-v0", 1, "cartpole_step_kernel", 73), ("C2", "RandomHopper-v0", 4096, "HopperSpec", 173), ("C3", "RandomHalfCheetahNoisy-v0", 16384, "HalfCheetahSpec", 273),
    ("C4", "RandomWalker2d-v0", 8192, "Walker2dSpec", 293), ("C5", "RandomHumanoid-v0", 32768, "humanoid_pair_step_kernel", 2073)])
def test_bench_config_lines(cfg, eid, B, kernel, nbytes):
    """`bench.py --config C1..C5`: SURVEY 8(d)'s inputs, each line with the roofline of its own kernel, a `metric` that names what the
    line measured (BASELINE.json's string belongs to the north-star line alone) and (C1, C2) the CPU leg on the GPU leg's settled states."""
    extra = () if cfg in ("C1", "C2") else ("--no-cpu-baseline",)
    d = _bench("--config", cfg, "--steps", "24", "--warmup", "4", "--settle", "60", "--cpu-sample-steps", "2", *extra)
    assert d["metric"] == "env-steps/sec at batch %d, %s, 1 MI355X; %% HBM roofline" % (B, eid)
    assert d["metric"] != json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    w = d["config"]["workload"]
    assert w.startswith(cfg + ": " + eid) and "batch %d per GPU" % B in w and cfg in d["config"]["baseline_config"]
    r = d["roofline"]
    assert kernel in r["kernel"] and r["bytes_per_env_step"] == nbytes and r["kernel_launches_timed"] >= 64
    assert abs(r["achieved"] - nbytes * B / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert d["nonfinite_lanes"] == 0 and d["overflow_lanes"] == 0
    if cfg in ("C1", "C2"):
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and "settled" in c["sample"] and c["value"] > 0 and c["cores"] >= 1
    if cfg == "C1":   # one env, one core: the reference's usage shape (test_random_policy.py:25-32) beside the GPU's step latency
        assert c["cores"] == 1 and c["one_env_one_core"] > 0 and "U{0,1}" in w and "no DR" in w
