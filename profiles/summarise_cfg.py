#!/usr/bin/env python3
"""Summarise a profiles/collect_cfg.sh run.  Usage: python profiles/summarise_cfg.py <tag> <key>
  <key> = the bench configuration ("RandomHopper-v0", "C2" ... "C5"): the entry of profiles/hbm_traffic.json bench.py reads
  -> profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
     profiles/<tag>_pmc_summary.json   per-launch means of every counter, per kernel
     profiles/hbm_traffic.json[key]    HBM bytes per launch of the step kernel + digest of the sources profiled
     profiles/<round>_configs.json[key]  the bench line next to the rocprof average of its kernel (<round> = the tag's prefix, "r04_c2" -> r04)"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
sys.path.insert(0, ROOT)
import bench  # noqa: E402

line = None
for l in open(os.path.join(P, "bench.json")):
    if l.startswith("{"):
        line = json.loads(l)
kname = line["roofline"]["kernel"]
short = kname.split("(")[0]
kstats = None
ks = sorted(glob.glob(os.path.join(P, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    for r in csv.DictReader(open(ks[-1])):
        if short in r["Name"]:
            kstats = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                      "percent": float(r["Percentage"])}
out = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in sorted(glob.glob(os.path.join(P, name, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "planar" in k or "cartpole" in k or "humanoid" in k or "walker" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                out.setdefault(k, {}).setdefault("_regs", {"VGPR": r.get("VGPR_Count"), "AGPR": r.get("Accum_VGPR_Count"), "SGPR": r.get("SGPR_Count"),
                                                           "scratch": r.get("Scratch_Size"), "LDS": r.get("LDS_Block_Size")})
        for k, v in agg.items():
            for c, vals in v.items():
                out.setdefault(k, {})[c] = {"n": len(vals), "mean_per_launch": sum(vals) / len(vals)}
# HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
# reports half the bytes of wide coalesced reads
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f = v["FETCH_SIZE"]["mean_per_launch"] * 1024; w = v["WRITE_SIZE"]["mean_per_launch"] * 1024
        v["hbm_bytes_per_launch"] = {"fetch_raw": f, "fetch_x2_gfx950": 2 * f, "write": w, "total_corrected": 2 * f + w}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
traffic = None
for k in out:
    if short.replace("void ", "") in k and "hbm_bytes_per_launch" in out[k]:
        traffic = out[k]["hbm_bytes_per_launch"]["total_corrected"]
tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
try:
    rec = json.load(open(tp))
except Exception:
    rec = {}
if traffic is not None:
    rec[key] = {"kernel": kname, "batch": int(line["config"]["global_batch"]) // max(int(line["n_gpus"]), 1), "bytes_per_launch": traffic, "source_digest": (os.environ.get("REX_SOURCE_DIGEST") or bench.source_digest()),
                "source": "profiles/%s_pmc_summary.json (FETCH_SIZE*1024*2 + WRITE_SIZE*1024)" % tag}
    json.dump(rec, open(tp, "w"), indent=1)
cp = os.path.join(ROOT, "profiles", tag.split("_")[0] + "_configs.json")
try:
    cfgs = json.load(open(cp))
except Exception:
    cfgs = {}
alg = line["roofline"]["algorithmic_bytes_per_launch"]
if traffic is not None and line["roofline"].get("traffic") != traffic:
    # the bench line of a collection is printed BEFORE its own PMC passes are summarised: on the box it read the previous collection's
    # hbm_traffic.json (other kernel sources: null, or another box's figure).  Record the line as bench.py prints it against THIS collection's file.
    line["roofline"]["traffic_as_printed"] = line["roofline"].get("traffic")
    line["roofline"]["traffic"] = traffic
    line["roofline"].pop("traffic_note", None)
    line["roofline"]["traffic_source"] = "profiles/hbm_traffic.json as written by this collection's PMC passes (summarise_cfg.py)"
cfgs[key] = {"bench_line": line, "rocprof_kernel_stats": kstats, "hbm_bytes_per_launch_pmc": traffic,
             "traffic_over_algorithmic": (traffic / alg) if traffic else None,
             "roofline_frac_from_rocprof": (alg / (kstats["avg_ns"] * 1e-9) / 1e9 / bench.HBM_PEAK_GBS) if kstats else None,
             "files": ["profiles/%s_kernel_stats.csv" % tag, "profiles/%s_pmc_summary.json" % tag], "source_digest": (os.environ.get("REX_SOURCE_DIGEST") or bench.source_digest())}
json.dump(cfgs, open(cp, "w"), indent=1)
print(key, "value %.2f M env-steps/s, kernel (events) %.4f ms, rocprof avg %s ms over %s launches, traffic %s MB = %sx algorithmic" % (
    line["value"] / 1e6, line["roofline"]["kernel_avg_ms"], kstats and "%.4f" % (kstats["avg_ns"] * 1e-6), kstats and kstats["calls"],
    traffic and "%.1f" % (traffic / 1e6), traffic and "%.2f" % (traffic / alg)))
