"""bench.py end to end on the GPU (small step counts): the one JSON line the driver parses, with the `roofline` and
`cpu_baseline` objects, for the headline env and for the humanoid."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, launcher=()):
    out = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *extra], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_line_contract():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = _run("--steps", "60", "--warmup", "10", "--batch", "4096", "--cpu-sample-steps", "2")
    # (a 4 096-env line names its own quantity; BASELINE.json's string is asserted on the 32 768-env line below)
    assert d["metric"] == "env-steps/sec at batch 4096, RandomHopper-v0, 1 MI355X; % HBM roofline" and d["unit"] == "env-steps/s"
    assert base["metric"].startswith("env-steps/sec at batch 32768, RandomHopper-v0")
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "RandomHopper-v0" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 * 60 / (d["ms_per_step"] * 60 / 1e3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - 173 * 4096 / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]   # algorithmic bytes / launch time
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["one_env_one_core"] > 0
    assert d["nonfinite_lanes"] == 0 and d["solver_capped_waves"] == 0


def test_line_names_the_launched_kernel_and_its_own_traffic_figure():
    """roofline.kernel follows the launch shape of the handle (rex_get_launch_shape), and roofline.traffic is reported only from a PMC record of
    the same batch, kernel instantiation and sources (profiles/hbm_traffic.json) -- never the 32 768-env figure next to another batch."""
    import random_envs_amd  # noqa: F401  (bench imports the package the same way)
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    d = _run("--steps", "24", "--warmup", "4", "--batch", "49152", "--no-cpu-baseline")
    r = d["roofline"]
    assert r["launch_shape"] == dict(lanes=64, pair=False, rolled=False, hum_pair=False) and r["kernel"] == "planar_step_kernel<rex::HopperSpec, false, false>"
    assert r["traffic"] is None
    d = _run("--steps", "24", "--warmup", "4", "--batch", "131072", "--no-cpu-baseline")
    assert d["roofline"]["launch_shape"]["rolled"] and d["roofline"]["kernel"] == "planar_step_kernel<rex::HopperSpec, false, true>"
    d = _run("--steps", "24", "--warmup", "4", "--no-cpu-baseline")
    r = d["roofline"]
    assert r["launch_shape"]["pair"] and r["kernel"] == "planar_step_kernel<rex::HopperSpec, true, false>"
    e = rec["RandomHopper-v0"]
    if e["source_digest"] == bench.source_digest():
        assert e["batch"] == 32768 and e["kernel"] == r["kernel"] and r["traffic"] == e["bytes_per_launch"]
    else:
        assert r["traffic"] is None


def test_humanoid_line():
    d = _run("--env", "RandomHumanoid-v0", "--steps", "12", "--warmup", "3", "--batch", "2048", "--no-cpu-baseline")
    assert "RandomHumanoid-v0" in d["config"]["workload"] and d["value"] > 1e5
    assert d["roofline"]["bytes_per_env_step"] == 2073 and d["roofline"]["kernel"] == "humanoid_pair_step_kernel"
    assert d["nonfinite_lanes"] == 0


def test_driver_command_reports_the_steady_state():
    """`bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command) against a 400-step run: no one-time cost inside the
    timed region (round 1 paid ~80 ms of first-use event set-up there: 4.1 ms per step instead of 0.14)."""
    short = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    long_ = _run("--gpus", "1", "--steps", "400", "--warmup", "50", "--no-cpu-baseline")
    assert short["config"]["global_batch"] == 32768
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert short["metric"] == base["metric"] and long_["metric"] == base["metric"]   # the north-star line carries BASELINE.json's metric verbatim
    assert abs(short["value"] - long_["value"]) < 0.10 * long_["value"], (short["value"], long_["value"])
    for d in (short, long_):
        r = d["roofline"]
        # the line is self-consistent: wall-clock fraction next to the kernel-time one, and the gap between them
        assert abs(r["frac_wall"] - d["value"] * r["bytes_per_env_step"] / 1e9 / r["peak"]) < 1e-9
        # (a launch bracketed by two event packets reads ~2 us longer than an un-bracketed one, and only every n-th launch
        # is bracketed: the wall-clock fraction may exceed the event-based one by that much)
        assert r["frac_wall"] <= r["frac"] * 1.06 and r["kernel_launches_timed"] >= 4
        assert d["host_gap_ms"] is not None and abs(d["host_gap_ms"]) < 0.25 * d["ms_per_step"] * d["steps"]


def test_two_rank_rehearsal_on_one_gpu():
    """The N > 1 launch path of bench.py end to end (one process per rank through torch.distributed.run, gloo, both ranks
    on GPU 0): SUM of the asynchronous step counter, MAX of the time, strong and weak sharding, and shard independence of the
    results -- RCCL itself needs the driver's multi-GPU node."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    launcher = ("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port))
    common = ("--gpus", "2", "--steps", "24", "--warmup", "4", "--backend", "gloo", "--same-device", "--no-cpu-baseline",
              "--counter-every", "8")
    weak = _run(*common, "--batch", "4096", launcher=launcher)
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["config"]["global_batch"] == 8192
    assert abs(weak["value"] - 8192 * 24 / (weak["ms_per_step"] * 24 / 1e3)) < 1e-6 * weak["value"]
    assert weak["counter_reductions"] == 4                                 # 3 in-loop + the final drain
    strong = _run(*common, "--batch", "4096", "--scaling", "strong", launcher=launcher)
    assert strong["scaling"] == "strong" and strong["config"]["global_batch"] == 4096
    assert "batch 2048 per GPU" in strong["config"]["workload"]
    assert strong["metric"] == "env-steps/sec at batch 4096 in total, RandomHopper-v0, 2 MI355X; % HBM roofline"
    assert weak["metric"] == "env-steps/sec at batch 4096 per GPU, RandomHopper-v0, 2 MI355X; % HBM roofline"


def test_self_launched_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with NO launcher: bench.py starts the two ranks itself (children created before the parent
    touches torch or the GPU), hands them RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and relays rank 0's line -- `n_gpus` is
    the size of the process group that was actually formed."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "24",
                          "--warmup", "4", "--batch", "4096", "--no-cpu-baseline", "--counter-every", "8"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env_clean)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8192
    assert abs(d["value"] - 8192 * 24 / (d["ms_per_step"] * 24 / 1e3)) < 1e-6 * d["value"]
    assert d["counter_reductions"] == 4


def test_replay_workload_line():
    """bench.py --replay: ONE logged transition replayed under a fresh candidate xi per env and step (set_task +
    set_sim_state + step, device-resident, no auto-reset) -- the second massively parallel workload (SURVEY 8 f2,
    random_hopper.py:128-152) -- with its own roofline entry (the xi row writes are part of its algorithmic bytes)."""
    d = _run("--replay", "--steps", "50", "--warmup", "10", "--batch", "8192", "--no-cpu-baseline")
    assert "replay" in d["config"]["workload"] and d["value"] > 1e6
    r = d["roofline"]
    assert r["bytes_per_env_step"] == 173 + 4 * 4 and r["traffic"] is None
    assert abs(r["achieved"] - r["bytes_per_env_step"] * 8192 / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert d["nonfinite_lanes"] == 0


def test_rccl_path_at_world_size_one():
    """backend "nccl" (= RCCL) end to end on the one GPU of this box: process group with device_id, the asynchronous
    all-reduce of the step counter on a device tensor, the MAX of the elapsed time, barrier, shutdown.  (N > 1 over xGMI
    needs the driver's multi-GPU node.)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, REX_FORCE_DIST="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "24", "--warmup", "4",
                          "--batch", "4096", "--counter-every", "8", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["counter_reductions"] == 4 and d["config"]["global_batch"] == 4096
    assert abs(d["value"] - 4096 * 24 / (d["ms_per_step"] * 24 / 1e3)) < 1e-6 * d["value"]
