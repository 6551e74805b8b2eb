"""Data-dependent initialisation (`--data-dep-init`, experiment_manager.py:61-72 of the reference).

PARITY UNPINNED: the reference calls `boilr.nn.init.data_dependent_init(model, {'x': batch})`, and boilr is not part of the
reference tree, so there is no fixture to pin this against. What is implemented is the algorithm the
reference's README and call site describe — one forward pass over the first training batch in which every convolution, in
execution order, is rescaled so that its output on that batch has zero mean and unit standard deviation per channel,
    w[co] <- w[co] / (std[co] + 1e-5),   b[co] <- (b[co] - mean[co]) / (std[co] + 1e-5),
and the rest of the pass sees the corrected output — on the HIP path: convolution, statistics and correction are the engine's
own kernels (lvae_conv2d_f32, lvae_bn_stats_f32, lvae_affine_act_f32); only the (Cout,)-sized parameter updates are torch
in-place ops.
"""
import torch

from . import kernels as K


def data_dependent_init(model, x):
    """x: (N, C, H, W) batch on the model's device. Modifies the convolution weights / biases in place; returns the number of
    convolutions that were rescaled."""
    was_training = model.training
    model.train()
    state = {'done': set(), 'count': 0}
    K._ddi = state
    try:
        with torch.no_grad():
            model(x)
    finally:
        K._ddi = None
        model.train(was_training)
    K.prepared.weights_written()
    return state['count']
