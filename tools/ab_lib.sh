#!/bin/bash
# A/B of two builds of the library inside ONE box (box-to-box variance is 3-8 %): per-kernel probes and the bf16 step,
# interleaved.  usage: tools/ab_lib.sh <lib A> <lib B> [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do for L in $A $B; do
    echo "== $L"
    SISR_LIB=$PWD/$L SISR_PRECISION=${PRECISION:-bf16} python tools/probe_kernels.py 2>/dev/null | grep -E "^(fwd|dgrad|wgrad)"
    SISR_LIB=$PWD/$L python bench.py --steps 30 --warmup 5 --precision ${PRECISION:-bf16} --no-cpu-baseline --configs none 2>/dev/null | \
        python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', r['ms_per_step'])"
done; done
