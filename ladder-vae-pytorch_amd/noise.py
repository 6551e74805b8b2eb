"""Noise sources for the stochastic ops of the hot path (Dropout2d masks, reparameterisation eps, likelihood
sample uniforms).

* `PhiloxNoise` — product path: on-device Philox4x32-10 (lvae_rng_fill_f32). A device-resident step counter is
  advanced once per forward by a plain kernel launch, so a captured hipGraph draws fresh numbers on each replay.
* `TapeNoise` — parity path: replays a recorded tape of the reference's draws, converting each
  entry to the NHWC / (N,C) layout the kernels read. On-device Philox cannot reproduce the CPU generator.
"""
import torch

from . import kernels as K


class PhiloxNoise:
    def __init__(self, seed=0, rank=0):
        self.seed = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & (2 ** 63 - 1)
        self.step = None
        self.site = 0
        self._masks = None
        self._mask_next = 0
        self._mask_shape = None

    def begin(self, device, mask_plan=None):
        """mask_plan = (count, N, C, p): all Dropout2d keep-masks of this forward are drawn by ONE launch into a
        (count, N, C) buffer and handed out in call order (306 tiny launches per CIFAR-15 step otherwise)."""
        if self.step is None or self.step.device != device:
            self.step = torch.zeros(1, dtype=torch.int64, device=device)
        self.site = 0
        self._masks = None
        self._mask_next = 0
        if mask_plan is not None and mask_plan[0] > 0:
            count, N, C, p = mask_plan
            self._mask_shape = (N, C, p)
            self._masks = self._fill((count, N, C), 'bernoulli', 1.0 - p, 1.0 / (1.0 - p), device)

    def end(self):
        K.counter_advance(self.step, 1)

    def _fill(self, shape, kind, lo, hi, device):
        self.site += 1
        return K.rng_fill(torch.empty(shape, dtype=torch.float32, device=device), kind, lo, hi, self.seed, self.step,
                          self.site)

    def dropout_mask(self, N, C, p, device):
        if self._masks is not None and self._mask_next < self._masks.shape[0] and self._mask_shape == (N, C, p):
            m = self._masks[self._mask_next]
            self._mask_next += 1
            return m
        return self._fill((N, C), 'bernoulli', 1.0 - p, 1.0 / (1.0 - p), device)

    def normal(self, shape_nhwc, device):
        return self._fill(shape_nhwc, 'normal', 0.0, 0.0, device)

    def uniform(self, shape, lo, hi, device, channel_last=True):
        return self._fill(shape, 'uniform', lo, hi, device)


class FrozenNoise:
    """The on-device draws of the FIRST forward, kept and handed out again in call order by every later forward — of this model or of
    another one that is given the same object. For tests that compare eager, hipGraph and reduced-precision steps on one noise tape at
    sizes where no CPU oracle tape exists (tests/test_fullsize_gpu.py)."""

    def __init__(self, seed=0):
        self.src = PhiloxNoise(seed)
        self.tape, self.pos, self.recorded = [], 0, False

    def begin(self, device, mask_plan=None):
        self.pos = 0
        if not self.recorded:
            self.src.begin(device, None)   # no mask plan: one draw per call, so that the tape is in call order

    def end(self):
        if not self.recorded:
            self.src.end()
            self.recorded = True

    def _get(self, make):
        if self.recorded:
            t = self.tape[self.pos]
        else:
            t = make()
            self.tape.append(t)
        self.pos += 1
        return t

    def dropout_mask(self, N, C, p, device):
        return self._get(lambda: self.src.dropout_mask(N, C, p, device))

    def normal(self, shape_nhwc, device):
        return self._get(lambda: self.src.normal(shape_nhwc, device))

    def uniform(self, shape, lo, hi, device, channel_last=True):
        return self._get(lambda: self.src.uniform(shape, lo, hi, device, channel_last))


class TapeNoise:
    """entries: the reference's draws in call order, in the reference's own shapes (NCHW / (B,C,1,1) / channel-last).

    Each entry is converted to the layout the kernels read and uploaded ONCE (cached per position), so a second pass over the
    same tape does no host work and can be captured into a hipGraph. loop=True rewinds at every `begin()` (every forward
    replays the same draws: the full-size parity tests run eager warm-up steps, the capture and the replay on one tape)."""

    def __init__(self, entries, loop=False):
        self.entries = [torch.as_tensor(e).float() for e in entries]
        self.pos = 0
        self.loop = loop
        self._dev = {}

    def begin(self, device, mask_plan=None):
        if self.loop:
            self.pos = 0

    def end(self):
        pass

    def _next(self, shape, device, convert):
        if self.pos >= len(self.entries):
            raise RuntimeError("noise tape exhausted at draw #%d" % self.pos)
        i = self.pos
        t = self.entries[i]
        self.pos += 1
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError("noise tape entry %d has shape %s, expected %s" % (i, tuple(t.shape), tuple(shape)))
        d = self._dev.get(i)
        if d is None or d.device != torch.device(device):
            if torch.device(device).type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("noise tape entry %d is not on the device yet: run one eager pass before capturing" % i)
            d = convert(t).contiguous().to(device)
            self._dev[i] = d
        return d

    def exhausted(self):
        return self.pos == len(self.entries)

    def dropout_mask(self, N, C, p, device):
        return self._next((N, C, 1, 1), device, lambda t: t.view(N, C) / (1.0 - p))

    def normal(self, shape_nhwc, device):
        N, H, W, Cn = shape_nhwc
        return self._next((N, Cn, H, W), device, lambda t: t.permute(0, 2, 3, 1))

    def uniform(self, shape, lo, hi, device, channel_last=True):
        if channel_last:
            return self._next(shape, device, lambda t: t)
        N, H, W, Cn = shape
        return self._next((N, Cn, H, W), device, lambda t: t.permute(0, 2, 3, 1))
