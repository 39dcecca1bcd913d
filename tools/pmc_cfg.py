"""HBM traffic of one full SRGAN iteration of a BASELINE.json config, from the PMC counters.

  run    (under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, one counter per pass):   python3 tools/pmc_cfg.py run cfg2
         -- the config's iteration exactly as bench.py builds it (bench.make_config_iteration), launched eagerly, 5 times
  reduce (after both passes):   python3 tools/pmc_cfg.py reduce <dir> cfg2 cfg3 ...   ->  <dir>/cfg_traffic.json
         -- an iteration ends with the generator's Adam step: the dispatches between two consecutive generator steps are one
            iteration (the first, which holds the one-time setup, is dropped).  FETCH_SIZE is doubled (gfx950 reports half of the
            bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); both counters are in KB.
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name):
    import torch
    import bench
    dev = torch.device('cuda:0')
    it, info, keep = bench.make_config_iteration(name, dev, 0, 1, False, print)
    for _ in range(5):
        it()
    torch.cuda.synchronize()


def per_iteration(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    ctr = rows[0]['Counter_Name']
    adam = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r['Kernel_Name']]
    ends = adam[1::2]                         # D's step, then G's step: every second one closes an iteration
    sums = []
    for a, b in zip(ends[:-1], ends[1:]):
        sums.append(sum(float(r['Counter_Value']) for r in rows[a + 1:b + 1]))
    by_kernel = {}
    a, b = ends[-2], ends[-1]
    for r in rows[a + 1:b + 1]:
        k = r['Kernel_Name'].split('(')[0][:48]
        by_kernel[k] = by_kernel.get(k, 0.0) + float(r['Counter_Value'])
    return ctr, sums, by_kernel


def reduce(d, names):
    out, detail = {}, {}
    commit = os.environ.get('SISR_COMMIT', '?')
    for name in names:
        rec = {}
        for path in glob.glob(os.path.join(d, name + '_*', '**', '*counter_collection.csv'), recursive=True):
            ctr, sums, by_kernel = per_iteration(path)
            scale = 2048.0 if ctr == 'FETCH_SIZE' else 1024.0
            rec[ctr] = {'bytes_per_iteration': scale * sum(sums) / len(sums), 'iterations': len(sums),
                        'spread': [scale * min(sums), scale * max(sums)]}
            detail.setdefault(name, {})[ctr] = {k: round(scale * v / 1e6, 2) for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:25]}
        if 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
            out[name + '_iteration'] = {
                'traffic_bytes_per_launch': rec['FETCH_SIZE']['bytes_per_iteration'] + rec['WRITE_SIZE']['bytes_per_iteration'],
                'read_bytes': rec['FETCH_SIZE']['bytes_per_iteration'], 'write_bytes': rec['WRITE_SIZE']['bytes_per_iteration'],
                'unit_of_launch': 'one full SRGAN iteration (eager launches, bf16 build, B=16)', 'iterations_averaged': rec['FETCH_SIZE']['iterations'],
                'commit': commit}
    json.dump(out, open(os.path.join(d, 'cfg_traffic.json'), 'w'), indent=1)
    json.dump(detail, open(os.path.join(d, 'cfg_traffic_by_kernel_MB.json'), 'w'), indent=1)
    for k, v in out.items():
        print(k, 'read %.1f MB  write %.1f MB' % (v['read_bytes'] / 1e6, v['write_bytes'] / 1e6))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2])
    else:
        reduce(sys.argv[2], sys.argv[3:])
