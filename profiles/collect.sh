#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the default bench command  -> average kernel durations
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters) -> HBM traffic per launch, VALU issue
# Usage: profiles/collect.sh <tag>      (writes gpurun_out/prof_<tag>/, summarise with profiles/summarise.py)
set -e
TAG=${1:-run}
P=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --no-cpu-baseline > $P/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $P/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $P/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $P/bench_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d $P/pmc_sq2 -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $P/bench_pmc_sq2.log 2>&1 || true
echo collected $P
