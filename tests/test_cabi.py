"""The C-ABI library loads and exports every symbol include/lvae_hip.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'lvae_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(lvae_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol():
    import lvae_amd  # noqa: F401
    from lvae_amd import _C
    assert os.path.exists(_C.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_C.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for name in syms:
        assert hasattr(lib, name), name
    assert sorted(_C.SIGNATURES) == syms  # the Python binding types exactly the declared surface
    assert _C.load().lvae_abi_version() == _C.ABI_VERSION
    assert _C.load().lvae_last_error() is not None


def test_abi_version_matches_header():
    import lvae_amd  # noqa: F401
    from lvae_amd import _C
    hdr = open(os.path.join(ROOT, 'include', 'lvae_hip.h')).read()
    assert int(re.search(r'#define LVAE_ABI_VERSION (\d+)', hdr).group(1)) == _C.ABI_VERSION


def test_struct_layout_matches_header():
    import lvae_amd  # noqa: F401
    from lvae_amd._C import ConvDesc
    # field order of struct lvae_conv_desc in the header
    hdr = open(os.path.join(ROOT, 'include', 'lvae_hip.h')).read()
    body = hdr[hdr.index('typedef struct lvae_conv_desc {'):hdr.index('} lvae_conv_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    body = body[body.index('{') + 1:]
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl or decl.startswith('typedef'):
            continue
        decl = decl.replace('const struct lvae_bn_fold*', '').replace('const float*', '').replace('float*', '').replace('void*', '').replace('int32_t', '').replace('int64_t', '').replace('uint8_t', '')
        names += [n.strip() for n in decl.split(',') if n.strip()]
    assert names == [f[0] for f in ConvDesc._fields_]


def _header_fields(struct):
    hdr = open(os.path.join(ROOT, 'include', 'lvae_hip.h')).read()
    body = hdr[hdr.index('typedef struct %s {' % struct):hdr.index('} %s;' % struct)]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    body = body[body.index('{') + 1:]
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        for ty in ('const float*', 'const void*', 'float*', 'void*', 'int32_t', 'int64_t', 'uint8_t', 'float '):
            decl = decl.replace(ty, '')
        names += [re.sub(r'\[\d+\]', '', n).strip() for n in decl.split(',') if n.strip()]
    return names


@pytest.mark.parametrize('struct,cls', [('lvae_rb_ext', 'RbExt'), ('lvae_bn_apply', 'BnApply'), ('lvae_bn_fold', 'BnFold')])
def test_extension_struct_layouts_match_header(struct, cls):
    """the structs that carry the fused launches' extras (round 5 added the deferred BatchNorm-backward apply to both)"""
    import lvae_amd  # noqa: F401
    from lvae_amd import _C
    assert _header_fields(struct) == [f[0] for f in getattr(_C, cls)._fields_]


def test_product_path_refuses_cpu_tensors():
    import torch
    import lvae_amd  # noqa: F401
    from lvae_amd import _C
    from lvae_amd.models.lvae import LadderVAE
    with pytest.raises(_C.LvaeHipError):
        _C.ptr(torch.zeros(4))
    m = LadderVAE(1, [4, 4], downsample=[1, 1], merge_type='residual', n_filters=8, dropout=0.1, img_shape=(16, 16),
                  likelihood_form='bernoulli', res_block_type='bacdbacd', gated=True)
    with pytest.raises(_C.LvaeHipError):
        m(torch.zeros(2, 1, 16, 16))  # no CPU fallback
