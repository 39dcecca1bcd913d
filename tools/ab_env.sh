#!/bin/bash
# A/B of one environment switch on the bf16 step inside ONE box (box-to-box variance is 3-8 %):
# usage: tools/ab_env.sh VAR [steps]   -> prints ms_per_step for VAR=1,0,1,0
var=$1; steps=${2:-30}
for v in 1 0 1 0; do
    env $var=$v python bench.py --steps $steps --warmup 5 --precision bf16 --no-cpu-baseline --configs none 2>/dev/null | \
        python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', r['ms_per_step'], r.get('perf_build',{}).get('ms_per_step'))"
done
