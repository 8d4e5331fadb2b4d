#!/bin/bash
# profiles/ab_env.sh <env-id> [ENV=VAL ...]: bench one env id with the product library (400 timed steps)
E=$1; shift
env "$@" python bench.py --env $E --steps 400 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$E $*', 'kernel_ms %.4f wall_ms %.4f value %.2fM capped %s nonfinite %s' % (d['roofline']['kernel_avg_ms'], d['ms_per_step'], d['value']/1e6, d['solver_capped_waves'], d['nonfinite_lanes']))
"
