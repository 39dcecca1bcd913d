#!/bin/bash
# per-kernel totals of one BASELINE config's full SRGAN iteration under rocprofv3: tools/kstats_cfg.sh cfg2 out.csv
C=${1:-cfg2}; OUT=$PWD/$2; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kc && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -- python3 $ROOT/bench.py --precision bf16 --steps 1 --warmup 0 --no-cpu-baseline --configs $C --config-iters 20 > /tmp/kc.log 2>&1
cp $(find /tmp/kc -name "*kernel_stats.csv" | head -1) $OUT
