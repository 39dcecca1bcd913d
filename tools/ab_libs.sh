#!/bin/bash
# A/B/C of several builds of the library inside ONE box: kernel probes + step, interleaved.  usage: PRECISION=fp32 tools/ab_libs.sh rounds lib1 lib2 ...
R=$1; shift
for r in $(seq $R); do for L in "$@"; do
    echo "== $L"
    SISR_LIB=$PWD/$L SISR_PRECISION=${PRECISION:-bf16} python tools/probe_kernels.py 2>/dev/null | grep -E "^(fwd|dgrad|wgrad)" | tr '\n' ';'; echo
    SISR_LIB=$PWD/$L python bench.py --steps 30 --warmup 5 --precision ${PRECISION:-bf16} --no-cpu-baseline --configs none 2>/dev/null | \
        python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', r['ms_per_step'])"
done; done
