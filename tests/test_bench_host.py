"""bench.py's host-side bookkeeping (no GPU): what a line calls itself, which kernel it prices, and when a committed PMC traffic
figure may be reported."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402


def test_metric_names_what_was_measured():
    base = json.load(open(os.path.join(os.path.dirname(HERE), "BASELINE.json")))
    assert bench.METRIC == base["metric"]
    m = bench.metric_of
    # BASELINE.json's string only on the line that measures it: hopper, 32 768 envs per GPU (weak) or in total (strong)
    assert m("RandomHopper-v0", 32768, 32768, 1, "weak") == bench.METRIC
    assert m("RandomHopper-v0", 32768, 8 * 32768, 8, "weak") == bench.METRIC
    assert m("RandomHopper-v0", 4096, 32768, 8, "strong") == bench.METRIC
    assert m("RandomHopper-v0", 32768, 32768, 1, "weak", replay=True) == "replayed env-steps/sec at batch 32768, RandomHopper-v0, 1 MI355X; % HBM roofline"
    assert m("RandomHopper-v0", 4096, 4096, 1, "weak") == "env-steps/sec at batch 4096, RandomHopper-v0, 1 MI355X; % HBM roofline"
    assert m("RandomHopper-v0", 4096, 8192, 2, "weak") == "env-steps/sec at batch 4096 per GPU, RandomHopper-v0, 2 MI355X; % HBM roofline"
    assert m("RandomHopper-v0", 2048, 4096, 2, "strong") == "env-steps/sec at batch 4096 in total, RandomHopper-v0, 2 MI355X; % HBM roofline"
    assert m("RandomWalker2d-v0", 32768, 32768, 1, "weak") == "env-steps/sec at batch 32768, RandomWalker2d-v0, 1 MI355X; % HBM roofline"
    assert set(bench.CONFIGS) == {"C1", "C2", "C3", "C4", "C5"}
    for k, c in bench.CONFIGS.items():     # every BASELINE configuration except the north star itself names its own workload
        assert m(c["env"], c["batch"], c["batch"], 1, "weak") != bench.METRIC, k


def test_kernel_names_follow_the_launch_shape():
    k = bench.kernel_of
    assert k("hopper", dict(pair=True, rolled=False, hum_pair=False)) == "planar_step_kernel<rex::HopperSpec, true, false>"
    assert k("hopper", dict(pair=False, rolled=True, hum_pair=False)) == "planar_step_kernel<rex::HopperSpec, false, true>"
    assert k("walker2d", dict(pair=False, rolled=False, hum_pair=False)) == "planar_step_kernel<rex::Walker2dSpec, false, false>"
    assert k("humanoid", dict(pair=False, rolled=False, hum_pair=True)) == "humanoid_pair_step_kernel"
    assert k("cartpole", dict(pair=False, rolled=False, hum_pair=False)) == "cartpole_step_kernel"


def test_pmc_traffic_only_for_the_sources_batch_and_kernel_it_was_measured_on(tmp_path, monkeypatch):
    root = tmp_path
    (root / "profiles").mkdir()
    (root / "random-envs_amd" / "csrc").mkdir(parents=True)
    (root / "random-envs_amd" / "csrc" / "a.hip").write_text("kernel v1")
    monkeypatch.setattr(bench, "ROOT", str(root))
    dig = bench.source_digest()
    rec = {"C2": {"kernel": "K", "batch": 4096, "bytes_per_launch": 123.0, "source_digest": dig},
           "RandomHopper-v0@8192": {"kernel": "K", "batch": 8192, "bytes_per_launch": 456.0, "source_digest": dig}}
    (root / "profiles" / "hbm_traffic.json").write_text(json.dumps(rec))
    assert bench.pmc_traffic("C2", 4096, "K") == (123.0, False)
    assert bench.pmc_traffic("RandomHopper-v0", 8192, "K") == (456.0, False)      # keyed by batch for the scaling legs
    assert bench.pmc_traffic("C2", 2048, "K") == (None, False)                    # another shard size: nothing measured
    assert bench.pmc_traffic("C2", 4096, "other kernel") == (None, False)         # another launch shape
    (root / "random-envs_amd" / "csrc" / "a.hip").write_text("kernel v2")
    assert bench.pmc_traffic("C2", 4096, "K") == (None, True)                     # the kernels changed: stale, never reported
    os.remove(root / "profiles" / "hbm_traffic.json")
    assert bench.pmc_traffic("C2", 4096, "K") == (None, False)
