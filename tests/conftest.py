import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never run by accident on a machine without a GPU
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _default_kernel_arithmetic():
    """kernels.precision / kernels.form are module-level requests that a model's forward pass (compute_dtype) or a test may have left
    behind: every test starts from the product defaults (fp32, LVAE_FORM_AUTO)."""
    mod = sys.modules.get('lvae_amd.kernels')
    if mod is not None:
        mod.precision, mod.form = mod.PREC_F32, mod._C.FORM_AUTO
    yield


class Golden:
    """One tests/golden/<name>.npz: dotted keys regrouped into lists / dicts."""

    def __init__(self, name):
        self.name = name
        self.raw = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
        self.cfg = json.loads(bytes(self.raw['cfg']).decode()) if 'cfg' in self.raw else None
        if self.cfg is not None:
            self.cfg['img_shape'] = tuple(self.cfg['img_shape'])

    def t(self, key):
        return torch.from_numpy(np.asarray(self.raw[key]))

    def group(self, prefix):
        """{suffix: tensor} of every key starting with `prefix.`"""
        n = len(prefix) + 1
        return {k[n:]: torch.from_numpy(v) for k, v in self.raw.items() if k.startswith(prefix + '.')}

    def seq(self, prefix):
        g = self.group(prefix)
        return [g[str(i)] for i in range(len(g))]

    def state_dict(self):
        return {k: v.clone() for k, v in self.group('sd').items()}


def load_golden(name):
    return Golden(name)
