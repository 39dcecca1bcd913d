mkdir -p gpurun_out/r2a
for v in new old; do for r in 1 0; do
  VARIANT=$v RELOAD=$r timeout -k 10 200 python tools/debug_cfg2_graph.py > gpurun_out/r2a/dbg_${v}_${r}.log 2>&1 || echo "variant $v $r rc=$?"
done; done
grep -h "===\|finite=False\|done\|iter\|Error" gpurun_out/r2a/dbg_*.log | head -120
