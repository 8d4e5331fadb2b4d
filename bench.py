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
BYTES_PER_ENV_STEP = 173          # read (6+6+3+4)*4 = 76, write (6+6+11+1)*4+1 = 97
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
NOMINAL = [3.5342917352885173, 3.9269908169872414, 2.7143360527015816, 5.0893800988154645]


def cpu_baseline(batch, steps, seed=0, min_seconds=12.0):
    """oracle ("port") timed on the host cores: `batch` envs x `steps` env-steps from reset states."""
    import numpy as np
    from oracle_bindings import oracle_rollout
    cores = len(os.sched_getaffinity(0))
    rng = np.random.RandomState(seed)
    q = rng.uniform(-.005, .005, (batch, 6)); q[:, 1] += 1.25
    v = rng.uniform(-.005, .005, (batch, 6))
    xi = np.array(NOMINAL) * rng.uniform(0.9, 1.1, (batch, 4))
    acts = rng.uniform(-1, 1, (steps, batch, 3))
    oracle_rollout("hopper", q[:64], v[:64], acts[:2, :64], xi[:64], nthreads=cores)   # warm the library
    # bounded sample: repeat the (batch x steps) rollout from the reset states until >= min_seconds
    # of wall time has been spent (MuJoCo's own solver tolerance 1e-8)
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < min_seconds and reps < 64:
        oracle_rollout("hopper", q, v, acts, xi, nthreads=cores)
        reps += 1; dt = time.perf_counter() - t0
    steps = steps * reps
    return dict(value=batch * steps / dt, unit="env-steps/s", cores=cores, kind="port",
                sample="%d envs x %d env-steps (%d-step rollouts from reset states, %d repeats), U(-1,1) actions, "
                       "xi nominal+-10%%, fp64, %d threads, %.1f s" % (batch, steps, steps // reps, reps, cores, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="envs per GPU")
    ap.add_argument("--env", default=ENV_ID)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-steps", type=int, default=8)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    if rank == 0:
        graft.build()
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        dist.barrier()
    import random_envs_amd as rex

    B = args.batch
    env = rex.make(args.env, batch=B, device=local_rank, seed=0, env_offset=rank * B)   # index-sharded batch
    nom = torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist())
    env.set_dr_training(True)
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    nact = 16
    actions = [(torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1).cuda(local_rank).contiguous() for _ in range(nact)]
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        env.step_soa(actions[k % nact])
    sync()
    env.enable_timing(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        env.step_soa(actions[k % nact])
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = env.read_timing()
    env.enable_timing(False)

    # the only data-path-adjacent collective: reduce the step counter (and the contract's max time)
    counter += args.steps * B
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(counter, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_steps = int(counter.item()); elapsed = float(tmax.item())
    counters = env.counters()

    if rank == 0:
        value = total_steps / elapsed
        kavg_ms = float(kernel_ms.mean()) if len(kernel_ms) else float("nan")
        achieved = BYTES_PER_ENV_STEP * B / (kavg_ms * 1e-3) / 1e9 if kavg_ms == kavg_ms else None
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC pass result (separate rocprofv3 --pmc runs)
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec at batch 32768, RandomHopper-v0", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, batch %d per GPU, uniform DR over 4 link masses (nominal +-10%%), "
                                   "U(-1,1) actions, auto-reset + xi resample" % (args.env, B),
                       "global_batch": B * world, "parallelism": "index-sharded envs x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "planar_step_kernel<HopperSpec>", "kernel_avg_ms": kavg_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * B,
                         "note": "latency/VALU-bound by construction: 16 forward-dynamics solves per 173 B"},
            "solver_capped_waves": counters["solver_capped"], "nonfinite_lanes": counters["nonfinite"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, args.cpu_sample_steps)
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
