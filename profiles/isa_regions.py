"""Instruction counts between the REX_MARK comment markers of a -DREX_MARKS -S build (tuning aid).
Usage: python profiles/isa_regions.py file.s [kernel-symbol-prefix]"""
import re, sys, collections
src = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "_Z18planar_step_kernelIN3rex10HopperSpec"
lines = open(src).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(kern)][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
cur = "entry"; counts = []; c = collections.Counter()
for l in lines[start:end]:
    m = re.search(r"; REXMARK (\w+)", l)
    if m:
        counts.append((cur, c)); cur = m.group(1); c = collections.Counter(); continue
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    c["all"] += 1
    if op.startswith("v_"): c["valu"] += 1
    if op.startswith("v_accvgpr"): c["acc"] += 1
    if op.startswith(("v_cndmask", "v_cmp")): c["cmp/cnd"] += 1
    if op.startswith("v_mov"): c["mov"] += 1
    if op.startswith("s_cbranch") or op == "s_branch": c["br"] += 1
    if op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac", "v_subrev_f32", "v_fmaak", "v_fmamk", "v_max_f32", "v_min_f32")): c["fp"] += 1
counts.append((cur, c))
for name, c in counts:
    print("%-14s all %5d valu %5d fp %5d acc %4d cmp/cnd %4d mov %4d br %3d" % (name, c["all"], c["valu"], c["fp"], c["acc"], c["cmp/cnd"], c["mov"], c["br"]))
