"""Aggregate a rocprofv3 rocpd .db kernel trace by (kernel, workgroups) — helper for reading profiles."""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(list)
for name, gx, wx, s, e in c.execute("select name, grid_x, workgroup_x, start, end from kernels"):
    m = re.search(r'(lvae::)?(\w+)(<[^(]*>)?\(', name)
    short = (m.group(2) + (m.group(3) or '')) if m else name[:50]
    if pat and pat not in short:
        continue
    agg[(short[:50], gx // max(1, wx))].append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
print('total %.2f ms' % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    v = sorted(v)
    print('%-52s wgs=%6d calls=%5d avg=%8.1f med=%8.1f us total=%8.2f ms' % (k[0], k[1], len(v), sum(v) / len(v), v[len(v) // 2], sum(v) / 1e3))
