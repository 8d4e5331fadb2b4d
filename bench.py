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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="envs per GPU")
    ap.add_argument("--env", default=ENV_ID)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: skip the per-launch HIP events")
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
    if rank == 0:
        graft.build()
    sharding.init(args.backend, torch.device("cuda", local_rank))   # "nccl" = RCCL over xGMI; no-op at N=1
    sharding.barrier()
    import random_envs_amd as rex

    B = args.batch
    env_offset, _ = sharding.shard(B, rank)                    # index-sharded batch, weak scaling
    env = rex.make(args.env, batch=B, device=local_rank, seed=0, env_offset=env_offset)
    nom = torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist())
    env.set_dr_training(True)
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    nact = 16
    amp = float(env.dims.act_high)   # U(-1,1) (hopper/walker/cheetah) or U(-0.4,0.4) (humanoid, humanoid.xml:6)
    actions = [((torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1) * amp).cuda(local_rank).contiguous() for _ in range(nact)]

    def sync():
        sharding.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        env.step_soa(actions[k % nact])
    sync()
    env.enable_timing(not args.no_kernel_timing)
    t0 = time.perf_counter()
    for k in range(args.steps):
        env.step_soa(actions[k % nact])
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = env.read_timing()
    env.enable_timing(False)

    # the only collectives of the path: SUM of the step counter, MAX of the elapsed time
    total_steps, elapsed = sharding.reduce_counter_and_time(args.steps * B, elapsed,
                                                            torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu"))
    counters = env.counters()

    if rank == 0:
        value = total_steps / elapsed
        kavg_ms = float(kernel_ms.mean()) if len(kernel_ms) else float("nan")
        bytes_step = BYTES_PER_ENV_STEP[args.env]
        achieved = bytes_step * B / (kavg_ms * 1e-3) / 1e9 if kavg_ms == kavg_ms else None
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC pass result (separate rocprofv3 --pmc runs)
        if os.path.exists(tp) and args.env == ENV_ID:
            try:
                traffic = json.load(open(tp)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": METRIC, "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, batch %d per GPU, uniform DR over the %d-dim xi (nominal +-10%%), "
                                   "U(-%.1f,%.1f) actions, auto-reset + xi resample" % (args.env, B, env.task_dim, amp, amp),
                       "global_batch": B * world, "parallelism": "index-sharded envs x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": KERNEL_NAME[args.env], "kernel_avg_ms": kavg_ms,
                         "algorithmic_bytes_per_launch": bytes_step * B, "bytes_per_env_step": bytes_step,
                         "note": "VALU-issue/latency-bound by construction (16 forward-dynamics solves per 173 B for hopper); HBM fraction reported because the metric asks for it"},
            "solver_capped_waves": counters["solver_capped"], "nonfinite_lanes": counters["nonfinite"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.env, B if args.env != "RandomHumanoid-v0" else 4096, args.cpu_sample_steps)
        print(json.dumps(out), flush=True)
    env.close()
    sharding.shutdown()


if __name__ == "__main__":
    main()
