#!/bin/bash
# per-kernel stats of the three builds of the bench step + the ordered kernel list of one replayed step of each (rocprofv3 --kernel-trace --stats)
out=$PWD/gpurun_out/${1:-stats}; ROOT=$PWD
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for b in bf16 bf16x3 fp32; do
    rm -rf /tmp/k_$b
    steps=20; [ $b = fp32 ] && steps=10
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$b -- python3 $ROOT/bench.py --precision $b --steps $steps --warmup 5 --no-cpu-baseline --configs none > $out/k_$b.log 2>&1 || exit 1
    cp $(find /tmp/k_$b -name "*kernel_stats.csv" | head -1) $out/${b}_kernel_stats.csv
    python3 $ROOT/tools/trace_order.py $(find /tmp/k_$b -name "*kernel_trace.csv" | head -1) > $out/order_$b.txt
    tail -1 $out/order_$b.txt
done
