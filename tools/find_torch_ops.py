"""Which torch-native ops (copies, adds, reductions) run inside one training step, with the Python call sites that issue them.
usage (GPU box): python tools/find_torch_ops.py > gpurun_out/torch_ops.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import lvae_amd  # noqa: F401
from lvae_amd.configs import CIFAR15, synthetic_images
from lvae_amd.engine import TrainStep
from lvae_amd.models.lvae import LadderVAE
from lvae_amd.noise import PhiloxNoise
from lvae_amd.optim import Adamax

torch.manual_seed(42)
model = LadderVAE(**CIFAR15).cuda().train()
model.noise = PhiloxNoise(seed=1)
opt = Adamax(model)
step = TrainStep(model, opt, use_graph=False)
x = synthetic_images(CIFAR15, 256, torch.Generator().manual_seed(1)).cuda()
for _ in range(2):
    step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(x)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by='count', row_limit=60, max_name_column_width=40, max_src_column_width=110))
