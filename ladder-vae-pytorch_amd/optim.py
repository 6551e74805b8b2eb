"""Adamax over the flat parameter arena — one kernel per step instead of torch.optim.Adamax's per-tensor loop
(experiment/experiment_manager.py:76-81: lr 3e-4, betas (0.9, 0.999), eps 1e-8, L2 weight decay added to the grad).
"""
import torch

from . import kernels as K


class Adamax:
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.exp_avg = self.exp_inf = self.step_count = None
        self.gscale = None  # device float[1]: 1/world_size after a SUM all-reduce
        self._arena = None

    def _state(self):
        arena = self.model.pack()
        if self._arena is not arena:
            dev = arena.params.device
            self.exp_avg = torch.zeros(arena.n_train, dtype=torch.float32, device=dev)
            self.exp_inf = torch.zeros(arena.n_train, dtype=torch.float32, device=dev)
            self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
            self._arena = arena
        return arena

    def zero_grad(self, set_to_none=False):
        self._state().zero_grad()

    def step(self):
        arena = self._state()
        K.adamax_step(arena.params[:arena.n_train], arena.grads, self.exp_avg, self.exp_inf, None, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, self.gscale, self.step_count)
        K.counter_advance(self.step_count, 1)

    def state_dict(self):
        self._state()
        return {'exp_avg': self.exp_avg, 'exp_inf': self.exp_inf, 'step': self.step_count, 'lr': self.lr,
                'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay}

    def load_state_dict(self, sd):
        self._state()
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_inf.copy_(sd['exp_inf'])
        self.step_count.copy_(sd['step'])
