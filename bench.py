#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched RandomHopper-v0 hot path on N MI355X (one process per GPU).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches it under
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE from the env).  A "step" is one
`env.step()` of the whole per-GPU batch: the hopper step kernel (4 RK4 mj_steps = 16 forward-
dynamics solves per env) + the masked auto-reset kernel (rocRAND init noise + xi resample).
Actions, state and xi are resident in HBM before the timed region starts.  The env batch is
sharded by index (weak scaling: 32768 envs per GPU); the ONLY collective is the all-reduce of the
step counter (plus the max-over-ranks of the elapsed time required by the contract).

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel (planar_step_kernel<HopperSpec>), from
                  HIP-event durations of every launch in the timed region,
  "cpu_baseline": the fp64 oracle (a CPU port of the same step, oracle/) timed on this host's cores
                  on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ENV_ID = "RandomHopper-v0"
BATCH_PER_GPU = 32768
# algorithmic bytes per env-step = read (qpos,qvel,action,xi) + write (qpos,qvel,obs,reward,done)
BYTES_PER_ENV_STEP = {"RandomHopper-v0": 173, "RandomWalker2d-v0": 293, "RandomHalfCheetah-v0": 273,
                      "RandomHalfCheetahNoisy-v0": 273, "RandomCartPole-v0": 73, "RandomHumanoid-v0": 2073}
KERNEL_NAME = {"RandomHopper-v0": "planar_step_kernel<HopperSpec>", "RandomWalker2d-v0": "planar_step_kernel<Walker2dSpec>",
               "RandomHalfCheetah-v0": "planar_step_kernel<HalfCheetahSpec>",
               "RandomHalfCheetahNoisy-v0": "planar_step_kernel<HalfCheetahSpec>", "RandomCartPole-v0": "cartpole_step_kernel",
               "RandomHumanoid-v0": "humanoid_step_kernel"}
METRIC = "env-steps/sec at batch 32768, RandomHopper-v0, 1/2/4/8 MI355X; % HBM roofline"   # BASELINE.json
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(env_id, batch, steps, seed=0, min_seconds=12.0):
    """oracle ("port") timed on the host cores: `batch` envs x `steps` env-steps from reset states."""
    import numpy as np
    from oracle_bindings import DIMS, oracle_rollout
    from random_envs_amd.registry import spec as env_spec
    from random_envs_amd.specs import IDS
    kind = IDS[env_id][0]
    d = DIMS[kind]; nominal = np.array(env_spec(env_id).nominal_task)
    cores = len(os.sched_getaffinity(0))
    rng = np.random.RandomState(seed)
    if kind == "humanoid":
        q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (batch, 1)) + rng.uniform(-.01, .01, (batch, 24))
        v = rng.uniform(-.01, .01, (batch, 23)); amp = 0.4
    elif kind == "halfcheetah":
        q = rng.uniform(-.1, .1, (batch, 9)); v = 0.1 * rng.randn(batch, 9); amp = 1.0
    else:
        q = rng.uniform(-.005, .005, (batch, d["nq"])); q[:, 1] += 1.25
        v = rng.uniform(-.005, .005, (batch, d["nv"])); amp = 1.0
    xi = nominal * rng.uniform(0.9, 1.1, (batch, d["nx"]))
    acts = rng.uniform(-amp, amp, (steps, batch, d["nu"]))
    oracle_rollout(kind, q[:64], v[:64], acts[:2, :64], xi[:64], nthreads=cores)   # warm the library
    # bounded sample: repeat the (batch x steps) rollout from the reset states until >= min_seconds
    # of wall time has been spent (MuJoCo's own solver tolerance 1e-8)
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < min_seconds and reps < 64:
        oracle_rollout(kind, q, v, acts, xi, nthreads=cores)
        reps += 1; dt = time.perf_counter() - t0
    steps = steps * reps
    # the reference's own usage shape: ONE env stepped by ONE core; ~2 s
    n1, t1, r1 = min(steps // reps, acts.shape[0]), 0.0, 0
    t0 = time.perf_counter()
    while t1 < 2.0 and r1 < 100000:
        oracle_rollout(kind, q[:1], v[:1], acts[:n1, :1], xi[:1], nthreads=1)
        r1 += 1; t1 = time.perf_counter() - t0
    one_core = n1 * r1 / t1
    return dict(value=batch * steps / dt, unit="env-steps/s", cores=cores, kind="port", one_env_one_core=one_core,
                sample="%s: %d envs x %d env-steps (%d-step rollouts from reset states, %d repeats), U(-a,a) actions, "
                       "xi nominal+-10%%, fp64, %d threads, %.1f s" % (env_id, batch, steps, steps // reps, reps, cores, dt))


def source_digest():
    """sha256 over the kernel sources: ties a committed PMC traffic figure to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "random-envs_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(env_id):
    """HBM bytes per launch of the dominant kernel from the separate rocprofv3 --pmc passes (profiles/collect.sh writes
    profiles/hbm_traffic.json together with the digest of the sources it profiled).  Returns (bytes or None, stale)."""
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(tp):
        return None, False
    try:
        rec = json.load(open(tp))
    except Exception:
        return None, False
    rec = rec.get(env_id, rec if env_id == ENV_ID else {})
    if not rec or "bytes_per_launch" not in rec:
        return None, False
    if rec.get("source_digest") != source_digest():
        return None, True          # kernels changed since the PMC run: do not report a stale figure
    return rec["bytes_per_launch"], False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="envs per GPU (weak) / global batch (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch envs on every GPU; strong: --batch envs in total, split by index (SURVEY 8d north-star point)")
    ap.add_argument("--env", default=ENV_ID)
    ap.add_argument("--replay", action="store_true",
                    help="second workload (SURVEY 8 f2): ONE logged transition replayed under a fresh candidate xi per env and step "
                         "(set_task + set_sim_state + step, no auto-reset)")
    ap.add_argument("--counter-every", type=int, default=256, help="steps between asynchronous all-reduces of the step counter")
    ap.add_argument("--settle", type=int, default=300,
                    help="untimed steps before the warm-up: right after reset every env is in the same phase of its first episode "
                         "(in the air, no contact rows -- cheaper than the steady state where episodes end and restart all the time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: skip the per-launch HIP events")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket every n-th launch of the timed region with HIP events (the two event packets cost ~8 us of stream "
                         "time per bracketed launch, 9 %% of a hopper step: timing every launch would lower `value` by that much)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the launch path on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--cpu-sample-steps", type=int, default=8)
    args = ap.parse_args()

    import torch
    import __graft_entry__ as graft
    from random_envs_amd import sharding
    rank, local_rank, world = sharding.dist_env()
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if rank == 0 and not os.environ.get("REX_LIB"):   # REX_LIB: a tuning build chosen by hand (profiles/ab_bench.sh)
        graft.build()
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live
    sharding.init(args.backend, dev)   # "nccl" = RCCL over xGMI; no-op at N=1
    sharding.barrier()
    import random_envs_amd as rex

    if args.scaling == "strong":
        env_offset, B = sharding.shard_strong(args.batch, rank, world)   # fixed global batch split by index
    else:
        env_offset, B = sharding.shard(args.batch, rank)                 # fixed per-GPU batch
    env = rex.make(args.env, batch=B, device=local_rank, seed=0, env_offset=env_offset,
                   autoreset=not args.replay)
    nom = torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist())
    env.set_dr_training(True)
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    nact = 16
    amp = float(env.dims.act_high)   # U(-1,1) (hopper/walker/cheetah) or U(-0.4,0.4) (humanoid, humanoid.xml:6)
    actions = [((torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1) * amp).cuda(local_rank).contiguous() for _ in range(nact)]
    if args.replay:
        # the logged transition: the state reached after a few random steps of env 0, replicated; candidates: fresh xi draws
        for k in range(8):
            env.step_soa(actions[k % nact])
        q, v = env.get_state()
        q0 = q[:1].clone().expand(B, -1).contiguous(); v0 = v[:1].clone().expand(B, -1).contiguous()
        cands = env.sample_tasks(nact).contiguous()        # [nact, B, task_dim] on the device

        fused = env.kind in ("hopper", "halfcheetah")
        q0z = q0.clone(); q0z[:, 0] = 0                     # get_full_mjstate: root x zeroed
        q_soa, v_soa = q0z.t().contiguous(), v0.t().contiguous()
        cands_soa = cands.transpose(1, 2).contiguous()      # [nact, task_dim, B]

        def one_step(k):
            if fused:                                       # ONE launch: caller's SoA (state, xi, action) in, (obs', r, done) out
                env.replay_soa(q_soa, v_soa, cands_soa[k % nact], actions[0])
            else:
                env.set_task(cands[k % nact])               # device-resident: no host round trip
                env.set_state(q0, v0)
                env.step_soa(actions[0])
    else:
        def one_step(k):
            env.step_soa(actions[k % nact])

    def sync():
        sharding.barrier()
        torch.cuda.synchronize()

    # every one-time cost goes BEFORE the timed region: the HIP-event pool is created here and the warm-up launches
    # are already timing-enabled (the first timed hipEventRecord on a stream pays a one-off profiling set-up)
    for k in range(0 if args.replay else args.settle):
        one_step(k)
    torch.cuda.synchronize()
    t_setup0 = time.perf_counter()
    env.enable_timing(0 if args.no_kernel_timing else max(1, min(args.time_every, max(args.steps // 4, 1))))
    one_step(0)
    torch.cuda.synchronize()
    timing_setup_ms = 1e3 * (time.perf_counter() - t_setup0)
    for k in range(args.warmup):
        one_step(k)
    sync()
    env.read_timing()                                        # drop the warm-up samples
    counter = sharding.StepCounter(cdev, every=args.counter_every)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k)
        counter.add(B)                                       # async all-reduce every --counter-every steps (N > 1)
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = env.read_timing()
    env.enable_timing(False)

    # the only collectives of the path: SUM of the step counter (asynchronous, above), MAX of the elapsed time
    total_steps = counter.total()
    elapsed = sharding.reduce_max(elapsed, cdev)
    counters = env.counters()

    if rank == 0:
        value = total_steps / elapsed
        n_k = len(kernel_ms)
        kavg_ms = float(kernel_ms.mean()) if n_k else float("nan")
        ksum_ms = float(kernel_ms.sum()) * (args.steps / n_k) if n_k else float("nan")
        bytes_step = BYTES_PER_ENV_STEP[args.env] + (4 * env.task_dim if args.replay else 0)   # replay also writes xi
        achieved = bytes_step * B / (kavg_ms * 1e-3) / 1e9 if kavg_ms == kavg_ms else None
        wall_ms = 1e3 * elapsed / args.steps
        achieved_wall = bytes_step * B / (wall_ms * 1e-3) / 1e9
        traffic, stale = pmc_traffic(args.env) if not args.replay else (None, False)
        out = {
            "metric": METRIC, "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall_ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s%s, batch %d per GPU, uniform DR over the %d-dim xi (nominal +-10%%), "
                                   "U(-%.1f,%.1f) actions, %s" % (args.env, " [replay: 1 logged transition x B candidate xi]" if args.replay else "",
                                                                 B, env.task_dim, amp, amp,
                                                                 ("one fused rex_replay launch per call" if args.env in ("RandomHopper-v0", "RandomHalfCheetah-v0") else "set_task + set_sim_state + step per call") if args.replay else "auto-reset + xi resample"),
                       "global_batch": total_steps // max(args.steps, 1),
                       "parallelism": "index-sharded envs x%d (%s scaling), no data-path collective; step counter all-reduced "
                                      "asynchronously every %d steps" % (world, args.scaling, args.counter_every)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": KERNEL_NAME[args.env], "kernel_avg_ms": kavg_ms, "kernel_launches_timed": n_k,
                         "algorithmic_bytes_per_launch": bytes_step * B, "bytes_per_env_step": bytes_step,
                         # the same fraction on the WALL clock of the timed region (what `value` is computed from)
                         "achieved_wall": achieved_wall, "frac_wall": achieved_wall / HBM_PEAK_GBS,
                         "note": "VALU-issue/latency-bound by construction (16 forward-dynamics solves per 173 B for hopper); HBM fraction reported because the metric asks for it"},
            # wall time of the timed region minus the summed kernel time: launch + host overhead per run
            "host_gap_ms": (1e3 * elapsed - ksum_ms) if n_k else None,
            "timing_setup_ms": timing_setup_ms,
            "counter_reductions": counter.reductions,
            "solver_capped_waves": counters["solver_capped"], "nonfinite_lanes": counters["nonfinite"],
        }
        if stale:
            out["roofline"]["traffic_note"] = "profiles/hbm_traffic.json was measured on other kernel sources (digest mismatch): re-run profiles/collect.sh"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.env, B if args.env != "RandomHumanoid-v0" else 4096, args.cpu_sample_steps)
        print(json.dumps(out), flush=True)
    env.close()
    sharding.shutdown()


if __name__ == "__main__":
    main()
