#!/bin/bash
# per (kernel, grid size) totals of one BASELINE config's full SRGAN iteration: which LAYERS the generic kernels spend their time on
# usage: tools/kgrid_cfg.sh cfg2 out.txt
C=${1:-cfg2}; OUT=$PWD/$2; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kg && rocprofv3 --kernel-trace --output-format csv -d /tmp/kg -- python3 $ROOT/bench.py --precision bf16 --steps 1 --warmup 0 --no-cpu-baseline --configs $C --config-iters 20 > /tmp/kg.log 2>&1
python3 - "$(find /tmp/kg -name '*kernel_trace.csv' | head -1)" > $OUT <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    if not ('conv_mfma' in name or 'wgrad_mfma' in name or 'fc_' in name or 'weights_' in name): continue
    key = (name[:60], 'x'.join(str(int(r.get('Grid_Size_' + d, 1)) // max(1, int(r.get('Workgroup_Size_' + d, 1)))) for d in 'XYZ'), r.get('Workgroup_Size_X', '?'))
    a = agg[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%-62s workgroups %12s wg %4s  calls %5d  avg %7.1f us  share %5.1f %%' % (k[0], k[1], k[2], v[0], v[1] / v[0], 100 * v[1] / tot))
PY
