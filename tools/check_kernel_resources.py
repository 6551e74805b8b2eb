"""Static resource check of every gfx950 kernel in csrc/ (ADVICE r4): compiles each .hip to assembly (device only, no GPU needed) and
lists, per kernel, VGPRs, spilled VGPRs and scratch bytes. Exits non-zero when a kernel that must not spill does:
  * every rb_conv_kernel instantiation (resblock_img.hip: one wave per SIMD, each spill is on the critical path of a dependent chain),
  * the 256-pixel Winograd kernels and the persistent gate kernels.
    python tools/check_kernel_resources.py [file.hip ...]          (default: the files named below; ~1 minute per file)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ladder-vae-pytorch_amd', 'csrc')
DEFAULT = ['resblock_img.hip', 'conv3x3_wino.hip', 'conv1x1_gate_bwd_fused.hip', 'conv1x1_gate_fwd.hip', 'conv_wgrad_img.hip', 'conv3x3_bf16.hip']
MUST_NOT_SPILL = ('rb_conv_kernel', 'conv3x3_wino2_kernel', 'conv1x1_gate_bwd_fused_bf16_kernel', 'conv1x1_gate_fwd_kernel', 'wgrad_img_kernel')


def kernels_of(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '--cuda-device-only', '-S', '-o', out, path],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    recs, cur = [], {}
    for line in text.split('\n'):
        m = re.match(r'\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size|sgpr_spill_count):\s+(\S+)', line)
        if not m:
            continue
        k, v = m.groups()
        if k == 'name' and cur.get('name') and 'vgpr_count' in cur:
            recs.append(cur)
            cur = {}
        cur[k] = v
    if cur.get('name') and 'vgpr_count' in cur:
        recs.append(cur)
    return recs


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    files = sys.argv[1:] or DEFAULT
    bad = 0
    for f in files:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        for r in kernels_of(path):
            name = demangle(r['name']).replace('lvae::', '').split('(')[0]
            spill, scratch = int(r.get('vgpr_spill_count', 0)), int(r.get('private_segment_fixed_size', 0))
            flag = ''
            if (spill or scratch) and any(k in name for k in MUST_NOT_SPILL):
                flag = '   <-- SPILLS'
                bad += 1
            print('%-28s %-70s vgpr %3s spill %3d scratch %4d%s' % (os.path.basename(f), name[:70], r['vgpr_count'], spill, scratch, flag))
    if bad:
        print('%d kernel(s) that must not spill do' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
