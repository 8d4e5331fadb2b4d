#!/bin/bash
# A/B of tuning builds on the GPU box: profiles/ab_bench.sh <lib.so> [ENV=VAL ...] -> one line per run (kernel avg ms, wall ms/step)
LIB=$1; shift
env REX_LIB=$LIB "$@" python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$LIB $*', 'kernel_ms %.4f wall_ms %.4f value %.1fM capped %s' % (d['roofline']['kernel_avg_ms'], d['ms_per_step'], d['value']/1e6, d['solver_capped_waves']))
"
