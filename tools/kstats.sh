#!/bin/bash
# per-kernel average durations of the bf16 (or PRECISION=fp32) bench step under rocprofv3 for one library build
# usage: tools/kstats.sh <lib.so> <out.csv>
L=$PWD/$1; OUT=$PWD/$2; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks && SISR_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $ROOT/bench.py --precision ${PRECISION:-bf16} --steps 20 --warmup 5 --no-cpu-baseline --configs none > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT
