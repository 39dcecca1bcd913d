"""Reduce the rocprofv3 --pmc passes of tools/pmc_collect.sh to per-launch averages per trunk kernel role and write
<dir>/pmc_per_launch.json (committed under profiles/ as r02_pmc_trunk_kernels_per_launch.json).  FETCH_SIZE is doubled
(gfx950: it reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM); sizes are in KB."""
import csv, glob, json, os, sys
d = sys.argv[1]
# the kernel a role launches: the persistent trunk kernels when they take the geometry, else the generic ones; the
# weight-gradient role also launches the slab reduction, whose traffic is added to the role's
KERNEL = {'fwd': ('conv_trunk_fwd_kernel', 'conv_trunk_f32_kernel', 'conv_mfma_bf16_kernel', 'conv_mfma_f32_kernel'),
          'dgrad': ('conv_trunk_bwd_kernel', 'conv_trunk_f32_kernel', 'conv_mfma_bf16_kernel', 'conv_mfma_f32_kernel'),
          'wgrad': ('wgrad_trunk_kernel', 'wgrad_trunk_f32_kernel', 'wgrad_mfma_bf16_kernel', 'wgrad_mfma_f32_kernel')}
EXTRA = {'wgrad': 'slab_reduce_kernel'}
out = {}
for role in ('fwd', 'dgrad', 'wgrad'):
    rec = {}
    for path in glob.glob(os.path.join(d, role + '_*', '**', '*counter_collection.csv'), recursive=True):
        rows = list(csv.DictReader(open(path)))
        name = next((k for k in KERNEL[role] if any(k in r['Kernel_Name'] for r in rows)), None)
        if name is None:
            continue
        rec['kernel'] = name
        per, extra = {}, {}
        for r in rows:
            if name in r['Kernel_Name']:
                per.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            elif role in EXTRA and EXTRA[role] in r['Kernel_Name']:
                extra.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        for k, v in per.items():
            v = v[5:] if len(v) > 8 else v              # drop the warm-up launches
            rec[k] = sum(v) / len(v)
        for k, v in extra.items():
            v = v[5:] if len(v) > 8 else v
            rec[EXTRA[role] + ':' + k] = sum(v) / len(v)
    if 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
        rec['read_bytes_per_launch'] = 2.0 * rec['FETCH_SIZE'] * 1024
        rec['write_bytes_per_launch'] = rec['WRITE_SIZE'] * 1024
        rec['traffic_bytes_per_launch'] = rec['read_bytes_per_launch'] + rec['write_bytes_per_launch']
        ex = EXTRA.get(role)
        if ex and ex + ':FETCH_SIZE' in rec and ex + ':WRITE_SIZE' in rec:
            rec['extra_traffic_bytes_per_launch'] = (2.0 * rec[ex + ':FETCH_SIZE'] + rec[ex + ':WRITE_SIZE']) * 1024
            rec['traffic_bytes_per_launch'] += rec['extra_traffic_bytes_per_launch']
    if 'SQ_WAVES' in rec and rec['SQ_WAVES'] > 0:
        w = rec['SQ_WAVES']
        rec['per_wave'] = {k[3:].lower(): round(rec[k] / w, 1) for k in rec if k.startswith('SQ_INSTS_')}
    out[role] = rec
json.dump(out, open(os.path.join(d, 'pmc_per_launch.json'), 'w'), indent=1)
# the file bench.py reads its `traffic` from (profiles/r02_traffic.json)
commit = os.environ.get('SISR_COMMIT', '?')
fam = {'fp32': 'f32', 'bf16x3': 'split'}.get(os.environ.get('SISR_PRECISION', 'bf16'), 'bf16')
json.dump({fam + '_' + role: {'traffic_bytes_per_launch': rec['traffic_bytes_per_launch'], 'kernel': rec.get('kernel'), 'commit': commit}
           for role, rec in out.items() if 'traffic_bytes_per_launch' in rec}, open(os.path.join(d, 'traffic.json'), 'w'), indent=1)
for role, rec in out.items():
    print(role, json.dumps(rec.get('per_wave', {})), 'traffic MB %.1f' % (rec.get('traffic_bytes_per_launch', 0) / 1e6))
