#!/bin/bash
# rocprofv3 evidence for ONE bench configuration (run on the GPU box through gpurun):
#   profiles/collect_cfg.sh <tag> [bench.py arguments, e.g. --config C3]
#   1. the bench line itself (roofline + cpu_baseline)                        -> gpurun_out/prof_<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command                  -> .../trace
#   3. separate --pmc passes: FETCH_SIZE, WRITE_SIZE, SQ counters            -> .../pmc_*
# summarise with profiles/summarise_cfg.py <tag> <key>
set -e
TAG=$1; shift
P=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $P
python3 bench.py "$@" > $P/bench.json 2> $P/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py "$@" --no-cpu-baseline > $P/bench_trace.log 2>&1
SHORT="--steps 40 --warmup 10 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py "$@" $SHORT > $P/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py "$@" $SHORT > $P/bench_pmc_write.log 2>&1
if [ -n "$REX_COLLECT_SQ" ]; then
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq -- python3 bench.py "$@" $SHORT > $P/bench_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d $P/pmc_sq2 -- python3 bench.py "$@" $SHORT > $P/bench_pmc_sq2.log 2>&1 || true
fi
echo collected $P
