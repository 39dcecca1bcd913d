#!/bin/bash
# Round-end evidence in one GPU call: full -m gpu suite, smoke, the default bench line, per-kernel stats of both builds
# and the ordered kernel list of one bf16 step.  Everything lands under gpurun_out/$1/.
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/test.log 2>&1; echo "pytest rc=$?" | tee $out/status.txt
tail -2 $out/test.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/status.txt
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/status.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -- python3 $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --configs none > $out/kb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -- python3 $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --configs none > $out/kf.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/kb -name "*kernel_stats.csv" | head -1) $out/bf16_kernel_stats.csv
cp $(find /tmp/kf -name "*kernel_stats.csv" | head -1) $out/fp32_kernel_stats.csv
python tools/trace_order.py $(find /tmp/kb -name "*kernel_trace.csv" | head -1) > $out/order_bf16.txt
tail -1 $out/order_bf16.txt
cut -c1-300 $out/bench.json
