#!/bin/bash
# Humanoid evidence: bench line at steady state (episodes end and restart all the time), rocprofv3 kernel stats of the same
# command, SQ instruction counters, and the in-kernel phase breakdown (diagnostic -DREX_KTIME build, if present).
# Usage: profiles/collect_humanoid.sh <tag>   (run on the GPU box through gpurun; writes gpurun_out/prof_<tag>/)
set -e
TAG=${1:-hum}
P=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $P
ARGS="--env RandomHumanoid-v0 --steps 300 --warmup 60"
python3 bench.py $ARGS > $P/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py $ARGS --no-cpu-baseline > $P/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq -- python3 bench.py --env RandomHumanoid-v0 --steps 40 --warmup 60 --no-cpu-baseline > $P/bench_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d $P/pmc_sq2 -- python3 bench.py --env RandomHumanoid-v0 --steps 40 --warmup 60 --no-cpu-baseline > $P/bench_pmc_sq2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --env RandomHumanoid-v0 --steps 40 --warmup 60 --no-cpu-baseline > $P/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --env RandomHumanoid-v0 --steps 40 --warmup 60 --no-cpu-baseline > $P/bench_pmc_write.log 2>&1
if [ -f random-envs_amd/librex_hip_ktime.so ]; then python3 profiles/ktime_probe_humanoid.py > $P/phases.log 2>&1 || true; fi
echo collected $P
