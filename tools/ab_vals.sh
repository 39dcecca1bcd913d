#!/bin/bash
# bf16 step time for a list of values of one environment switch inside ONE box: tools/ab_vals.sh VAR v1 v2 ...  (twice each)
var=$1; shift
for rep in 1 2; do for v in "$@"; do
    env $var=$v python bench.py --steps 30 --warmup 5 --precision ${PRECISION:-bf16} --no-cpu-baseline --configs none 2>/dev/null | \
        python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', r['ms_per_step'])"
done; done
