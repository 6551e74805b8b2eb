"""Build profiles/<round>_pmc/hbm_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
tools/conv_bench.py. usage: python tools/pmc_json.py <FETCH csv> <WRITE csv> <out json>"""
import collections
import csv
import json
import re
import sys


def collect(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        name = r['Kernel_Name']
        m = re.search(r'(lvae::)?(\w+)(<[^(]*>)?\(', name)
        short = (m.group(2) + (m.group(3) or '')) if m else name[:60]
        if not short.startswith(('conv', 'wgrad', 'wino')):
            continue
        key = '%s @%d workgroups' % (short, int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])))
        tot[key] += float(r['Counter_Value'])
        cnt[key] += 1
    return tot, cnt


ft, fc = collect(sys.argv[1], 'FETCH_SIZE')
wt, wc = collect(sys.argv[2], 'WRITE_SIZE')
out = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/conv_bench.py on MI355X; '
                 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts half the bytes of a wide coalesced read: '
                 'MI355X_MICROARCH.md, HBM section). rocprofv3 --pmc segfaults inside the profiler on the full training step, so the '
                 'counters were taken on the single-layer micro-benchmark (same kernels, same shapes, B=256).',
       'kernels': {}}
for k in ft:
    if k not in wt:
        continue
    f, w = ft[k] / fc[k], wt[k] / wc[k]
    out['kernels'][k] = {'FETCH_SIZE_KB': f, 'WRITE_SIZE_KB': w, 'hbm_bytes_per_launch': (2 * f + w) * 1024, 'launches': fc[k]}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:12]:
    print('%-60s %8.1f MB/launch' % (k, v['hbm_bytes_per_launch'] / 1e6))
