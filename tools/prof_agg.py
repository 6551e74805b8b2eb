"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid) — helper for reading profiles.
python tools/prof_agg.py <kernel_trace.csv> 0 <rows> [last-step]   (last-step: only the launches of the last replayed training step)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 4 and sys.argv[4] == 'last-step':
    # exactly one replayed training step: the launches behind the second-to-last Adamax launch up to and including the last one
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    opt = [i for i, r in enumerate(rows) if 'adamax_kernel' in r['Kernel_Name']]
    rows = rows[opt[-2] + 1:opt[-1] + 1]
    print('# one replayed step: %d launches, %.3f ms from the first start to the last end' %
          (len(rows), (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r['Kernel_Name']
    m = re.search(r'(lvae::)?(\w+)(<[^(]*>)?\(', name)
    short = (m.group(2) + (m.group(3) or '')) if m else name[:60]
    if 'at::native' in name:
        short = 'torch:' + (re.search(r'(\w+Functor\w*|direct_copy|fill|reduce_kernel|\w+_kernel)', name).group(1) if re.search(r'(\w+Functor\w*|direct_copy|fill|reduce_kernel|\w+_kernel)', name) else 'other')
    key = (short[:58], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
    agg[key][0] += 1
    agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print('total kernel time %.1f ms over %d launches' % (tot / 1e3, sum(v[0] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%-58s wgs=%6d calls=%5d avg=%8.1f us total=%8.1f ms %5.1f%%' % (k[0], k[1], v[0], v[1] / v[0], v[1] / 1e3, 100 * v[1] / tot))
