"""HBM bytes per launch of the thin-layer kernels from two rocprofv3 --pmc passes over tools/probe_thin.py
(FETCH_SIZE, WRITE_SIZE; units and the gfx950 FETCH_SIZE correction as in MI355X_MICROARCH.md / tools/pmc_reduce.py).
usage: python tools/pmc_thin.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys, collections
NAMES = ('conv_thin_kernel', 'wgrad_thin_kernel', 'conv_toimage_kernel', 'wgrad_toimage_kernel')


def mean_counter(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            k = next((n for n in NAMES if n in r['Kernel_Name']), None)
            if k:
                acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = mean_counter(sys.argv[1], 'FETCH_SIZE'), mean_counter(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write)):
    fb, wb = fetch.get(k, 0.0) * 2 * 1024, write.get(k, 0.0) * 1024
    out[k] = {'fetch_MB': round(fb / 1e6, 2), 'write_MB': round(wb / 1e6, 2), 'hbm_MB': round((fb + wb) / 1e6, 2)}
json.dump({'note': 'per launch, B16, LR 96 / HR 192, bf16 build; FETCH_SIZE x 2 x 1 KB + WRITE_SIZE x 1 KB', 'kernels': out},
          open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
