#!/bin/bash
# kernel stats + per-(kernel, grid) totals of BASELINE configs' full SRGAN iterations under rocprofv3 (one trace per config)
# usage: tools/cfg_profile.sh <out-prefix under gpurun_out/> cfg2 cfg3 ...
PFX=$1; shift
ROOT=$PWD
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in "$@"; do
  rm -rf /tmp/kc_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc_$C -- python3 $ROOT/bench.py --precision bf16 --steps 1 --warmup 0 --no-cpu-baseline --configs $C --config-iters 10 > $ROOT/gpurun_out/${PFX}_$C.log 2>&1 || { echo "profile of $C failed"; tail -5 $ROOT/gpurun_out/${PFX}_$C.log; exit 1; }
  cp $(find /tmp/kc_$C -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${PFX}_${C}_kernel_stats.csv
  python3 - "$(find /tmp/kc_$C -name '*kernel_trace.csv' | head -1)" > $ROOT/gpurun_out/${PFX}_${C}_by_grid.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    key = (name[:70], 'x'.join(str(int(r.get('Grid_Size_' + d, 1)) // max(1, int(r.get('Workgroup_Size_' + d, 1)))) for d in 'XYZ'), r.get('Workgroup_Size_X', '?'))
    a = agg[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print('total kernel time %.1f us' % tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%-72s workgroups %12s wg %4s  calls %5d  avg %7.1f us  share %5.1f %%' % (k[0], k[1], k[2], v[0], v[1] / v[0], 100 * v[1] / tot))
PY
  tail -1 $ROOT/gpurun_out/${PFX}_$C.log | cut -c1-400
done
