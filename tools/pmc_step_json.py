"""profiles/<round>_pmc/hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over an EAGER training step
(tools/r03_pmc_step.sh). usage: python tools/pmc_step_json.py <FETCH csv> <WRITE csv> <n steps in the run> <out json> [bytes per image of the
algorithmic convolution traffic: 115.93e6 fp32 (default), 57.96e6 for bf16 storage]"""
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
        if short.startswith('__amd_rocclr'):
            continue  # model construction (host -> device parameter copies), not the step
        key = '%s @%d workgroups' % (short, int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])))
        tot[key] += float(r['Counter_Value'])
        cnt[key] += 1
    return tot, cnt


ft, fc = collect(sys.argv[1], 'FETCH_SIZE')
wt, wc = collect(sys.argv[2], 'WRITE_SIZE')
nsteps = float(sys.argv[3])
b_img = float(sys.argv[5]) if len(sys.argv) > 5 else 115.93e6
out = {'scope': 'whole training step, eager launches (bench.py --no-graph), %d steps in the run' % nsteps,
       'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/r03_pmc_step.sh) on MI355X; hbm_bytes = '
                 '(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts half the bytes of a wide coalesced read: '
                 'MI355X_MICROARCH.md, HBM section; Infinity-Cache hits are included in the counters)',
       'kernels': {}}
step_f = step_w = 0.0
for k in ft:
    if k not in wt:
        continue
    f, w = ft[k] / fc[k], wt[k] / wc[k]
    out['kernels'][k] = {'FETCH_SIZE_KB': f, 'WRITE_SIZE_KB': w, 'hbm_bytes_per_launch': (2 * f + w) * 1024, 'launches_per_step': fc[k] / nsteps}
    step_f += ft[k] / nsteps
    step_w += wt[k] / nsteps
out['step'] = {'FETCH_SIZE_KB': step_f, 'WRITE_SIZE_KB': step_w, 'hbm_bytes_per_step': (2 * step_f + step_w) * 1024,
               'algorithmic_conv_bytes_per_step': b_img * 256}
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print('step: %.2f GB (2*FETCH + WRITE); algorithmic conv traffic %.2f GB' % (out['step']['hbm_bytes_per_step'] / 1e9, b_img * 256 / 1e9))
for k, v in sorted(out['kernels'].items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches_per_step'])[:14]:
    print('%-60s %8.1f MB/launch x %6.1f /step' % (k, v['hbm_bytes_per_launch'] / 1e6, v['launches_per_step']))
