#!/bin/bash
# per-kernel stats of both builds of the bench step + ordered kernel list of one bf16 step (rocprofv3 --kernel-trace --stats)
out=$PWD/gpurun_out/${1:-stats}; ROOT=$PWD
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kb /tmp/kf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -- python3 $ROOT/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --configs none > $out/kb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -- python3 $ROOT/bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --configs none > $out/kf.log 2>&1
cd $ROOT
cp $(find /tmp/kb -name "*kernel_stats.csv" | head -1) $out/bf16_kernel_stats.csv
cp $(find /tmp/kf -name "*kernel_stats.csv" | head -1) $out/fp32_kernel_stats.csv
python tools/trace_order.py $(find /tmp/kb -name "*kernel_trace.csv" | head -1) > $out/order_bf16.txt
python tools/trace_order.py $(find /tmp/kf -name "*kernel_trace.csv" | head -1) > $out/order_fp32.txt
tail -1 $out/order_bf16.txt; tail -1 $out/order_fp32.txt
