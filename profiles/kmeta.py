import subprocess,sys,re
out=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf","--notes",sys.argv[1]],capture_output=True,text=True).stdout
cur={}
rows=[]
for l in out.splitlines():
    m=re.match(r"\s+-?\s*\.(\w+):\s+(.*)",l)
    if not m: continue
    k,v=m.groups()
    if k=="agpr_count" and cur.get("name"): rows.append(cur); cur={}
    cur[k]=v
rows.append(cur)
seen=set()
for r in rows:
    n=r.get("name","")
    if n in seen or not n: continue
    seen.add(n)
    d=subprocess.run(["c++filt",n],capture_output=True,text=True).stdout.strip().split("(")[0]
    print("%-70s vgpr %s agpr %s spill %s scratch %s lds %s sgpr %s" % (d[:70], r.get("vgpr_count"), r.get("agpr_count"), r.get("vgpr_spill_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size"), r.get("sgpr_count")))
