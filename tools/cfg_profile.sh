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
  python3 - "$(find /tmp/kc_$C -name '*kernel_trace.csv' | head -1)" $ROOT/gpurun_out/${PFX}_${C}_order.txt > $ROOT/gpurun_out/${PFX}_${C}_by_grid.txt <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# the last 5 iterations only (model construction, warm-ups and the capture are not the iteration): an iteration ends with the
# generator's Adam step, the second adam_step_kernel of the iteration
adam = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r['Kernel_Name']]
ends = adam[1::2]
K = 5
if len(ends) > K:
    rows = rows[ends[-K - 1] + 1:ends[-1] + 1]
    span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3 / K
else:
    K, span = 1, 0.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r['Kernel_Name']
    key = (name[:70], 'x'.join(str(int(r.get('Grid_Size_' + d, 1)) // max(1, int(r.get('Workgroup_Size_' + d, 1)))) for d in 'XYZ'), r.get('Workgroup_Size_X', '?'))
    a = agg[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
# launch order of the LAST iteration: start offset, duration, idle gap in front of it (us)
it = rows[ends[-2] + 1 - (ends[-K - 1] + 1):] if len(ends) > K else rows
t0, prev_end = int(it[0]['Start_Timestamp']), None
with open(sys.argv[2], 'w') as fo:
    for r in it:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        grid = 'x'.join(str(int(r.get('Grid_Size_' + d, 1)) // max(1, int(r.get('Workgroup_Size_' + d, 1)))) for d in 'XYZ')
        fo.write('%9.1f  dur %7.1f  gap %6.1f  %-60s %s\n' % ((st - t0) / 1e3, (en - st) / 1e3, 0.0 if prev_end is None else (st - prev_end) / 1e3, r['Kernel_Name'][:60], grid))
        prev_end = en
print('last %d iterations: %.1f us of kernel time per iteration, %d launches per iteration, %.1f us wall per iteration (profiled)' % (K, tot / K, len(rows) // K, span))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print('%-72s workgroups %12s wg %4s  per-iter calls %5.1f  avg %7.1f us  per-iter %7.1f us  share %5.1f %%' % (k[0], k[1], k[2], v[0] / K, v[1] / v[0], v[1] / K, 100 * v[1] / tot))
PY
  tail -1 $ROOT/gpurun_out/${PFX}_$C.log | cut -c1-400
done
