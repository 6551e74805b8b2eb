"""Static check of the emitted gfx950 code: which kernels wait on the vector-memory counter (s_waitcnt vmcnt) AFTER their first
write-through (sc1) store? The counter retires in order, so such a wait also waits for the store's trip to memory (1-2 us): loads that
an epilogue needs must be requested AND consumed before its first store. Compiles csrc/*.hip to assembly (no GPU needed).
python tools/wt_store_waits.py [-v]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'ladder-vae-pytorch_amd', 'csrc')
out = tempfile.mkdtemp(prefix='lvae_isa_')
procs = []
for f in sorted(glob.glob(os.path.join(src, '*.hip'))):
    s = os.path.join(out, os.path.basename(f)[:-4] + '.s')
    procs.append((s, subprocess.Popen(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(root, 'include'),
                                       '-S', '--cuda-device-only', '-o', s, f], stderr=subprocess.DEVNULL)))
rows = []
for s, p in procs:
    if p.wait() != 0:
        print('compile failed:', s)
        continue
    name, first, waits, stores, where = None, None, 0, 0, []
    for i, l in enumerate(open(s).read().split('\n')):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, first, waits, stores, where = m.group(1), None, 0, 0, []
            continue
        if name is None:
            continue
        if 'global_store' in l and 'sc1' in l:
            stores += 1
            first = i if first is None else first
        elif first is not None and re.search(r's_waitcnt.*vmcnt\(', l):
            waits += 1
            where.append(i + 1)
        if 's_endpgm' in l:
            if stores:
                rows.append((os.path.basename(s), name, stores, waits, where))
            name = None
try:
    dem = subprocess.run(['c++filt'] + [r[1] for r in rows], capture_output=True, text=True).stdout.split('\n')
except OSError:
    dem = [r[1] for r in rows]
for (f, n, st, w, where), d in zip(rows, dem):
    if w or '-v' in sys.argv:
        print('%-28s %-100s sc1 stores %3d  vmcnt waits behind the first %3d %s' % (f, d[:100], st, w, where[:6] if '-v' in sys.argv else ''))
