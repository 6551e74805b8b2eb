"""Sum a rocprofv3 --pmc counter over the dispatches of the conv kernel family (per step) — profiling helper.
usage: python tools/pmc_traffic.py <counter_collection.csv> <COUNTER> <n_steps_profiled>"""
import collections
import csv
import re
import sys

path, counter, nsteps = sys.argv[1], sys.argv[2], float(sys.argv[3])
fam = re.compile(r'conv3x3_halo_kernel|conv1x1_kernel|conv_igemm_kernel')
tot = collections.Counter()
cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    if r.get('Counter_Name') != counter:
        continue
    name = r['Kernel_Name']
    m = re.search(r'(lvae::)?(\w+)(<[^(]*>)?\(', name)
    short = m.group(2) if m else name[:40]
    v = float(r['Counter_Value'])
    key = 'CONV_FWD_DGRAD' if fam.search(name) else ('WGRAD' if 'wgrad' in name else 'OTHER')
    tot[key] += v
    cnt[key] += 1
    tot[short] += v
    cnt[short] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print('%-28s %s per step = %14.1f   (dispatches per step %.0f)' % (k, counter, v / nsteps, cnt[k] / nsteps))
