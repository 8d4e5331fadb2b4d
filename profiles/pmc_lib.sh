#!/bin/bash
# SQ counters of a tuning build: profiles/pmc_lib.sh <tag> <lib.so> [bench args...]  (env knobs: export them before calling)
# -> gpurun_out/pmc_<tag>/summary.txt : per wave and env-step VALU / SALU / branch instructions, cycles, waits
TAG=$1; LIB=$2; shift 2
export REX_LIB=$LIB TMPDIR=/tmp
P=gpurun_out/pmc_$TAG; mkdir -p $P
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d $P/a -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > $P/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d $P/b -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > $P/b.log 2>&1
python3 - $P <<'PY'
import csv,glob,sys,collections
P=sys.argv[1]; agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(P+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'step_kernel' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(P+'/summary.txt','w') as out:
    for k,v in agg.items():
        m={c:sum(x)/len(x) for c,x in v.items()}
        w=m.get('SQ_WAVES',1)
        line=k+' | waves %d'%w+' | per wave-step: '+' '.join('%s %.0f'%(c.replace('SQ_',''),m[c]/w) for c in sorted(m) if c!='SQ_WAVES')
        if 'SQ_INSTS_VALU' in m and 'SQ_WAVE_CYCLES' in m: line+=' | quad-cycles per VALU %.2f'%(m['SQ_WAVE_CYCLES']/m['SQ_INSTS_VALU'])
        print(line); out.write(line+'\n')
PY
