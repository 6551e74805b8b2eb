"""Generate tests/golden/*.npz by running the REAL reference modules (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY; runs in the build container only (the reference never travels to the GPU box).
The reference has no tests or golden vectors of its own, so these files are the pins:
inputs, the full state_dict (or the seed it was initialised from), the NOISE TAPE (every RNG draw of the
forward, captured with a TorchDispatchMode), all 12 forward outputs, the forward_pass scalars, parameter
gradients, and the parameters after one torch.optim.Adamax step.

The reference is imported unmodified; the five absent `boilr` names are provided in memory by
oracle/boilr_standins.py (our restatement -> parity unpinned at that boundary).

usage:  python oracle/gen_golden.py [case ...]
"""
import json
import os
import sys

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import boilr_standins  # noqa: E402

boilr_standins.install()
from models.lvae import LadderVAE  # noqa: E402  (the reference)
from lib import likelihoods as ref_lik  # noqa: E402
from lib import stochastic as ref_stoch  # noqa: E402

_RANDOM_OPS = ('bernoulli_', 'normal_', 'uniform_', 'rand_like', 'randn_like', 'rand', 'randn', 'normal')


class RecordRNG(TorchDispatchMode):
    """Records the result of every random ATen op, in call order."""

    def __init__(self):
        super().__init__()
        self.tape = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split('.')[0]
        if name in _RANDOM_OPS:
            self.tape.append(out.detach().clone())
        return out


BASE = dict(color_ch=1, z_dims=[8, 8], blocks_per_layer=1, downsample=[1, 1], nonlin='elu', merge_type='residual',
            batchnorm=True, stochastic_skip=True, n_filters=16, dropout=0.2, free_bits=0.5, learn_top_prior=False,
            img_shape=(28, 28), likelihood_form='bernoulli', res_block_type='bacdbacd', gated=True,
            no_initial_downscaling=False, analytical_kl=False)


def cfg_of(**over):
    c = dict(BASE)
    c.update(over)
    return c


# name -> (cfg, batch, seed, extras)
CASES = {
    'tiny_mnist': (cfg_of(), 4, 11, {}),
    'tiny_cifar': (cfg_of(color_ch=3, img_shape=(32, 32), likelihood_form='discr_log_mix', z_dims=[8, 8, 8],
                          downsample=[0, 1, 1], blocks_per_layer=2, learn_top_prior=True, free_bits=1.0, n_filters=8), 4, 12, {}),
    'tiny_eval': (cfg_of(), 4, 13, {'eval': True}),
    'tiny_cabdcabd': (cfg_of(res_block_type='cabdcabd', gated=False, stochastic_skip=False, merge_type='linear',
                             nonlin='relu', free_bits=0.0), 3, 14, {}),
    'tiny_bacdbac': (cfg_of(res_block_type='bacdbac', nonlin='leakyrelu', analytical_kl=True,
                            img_shape=(16, 16), z_dims=[4, 8], n_filters=8), 3, 15, {}),
    'tiny_nobn_selu': (cfg_of(batchnorm=False, nonlin='selu', no_initial_downscaling=True, img_shape=(16, 16),
                              downsample=[1, 0], n_filters=8, learn_top_prior=True), 3, 16, {}),
    'tiny_gauss': (cfg_of(color_ch=3, img_shape=(16, 16), likelihood_form='gaussian', n_filters=8), 3, 17, {}),
    'tiny_discrlog': (cfg_of(color_ch=3, img_shape=(16, 16), likelihood_form='discr_log', n_filters=8), 3, 18, {}),
    'tiny_prior': (cfg_of(color_ch=3, img_shape=(16, 16), likelihood_form='discr_log_mix', z_dims=[4, 4, 4],
                          downsample=[1, 0, 1], n_filters=8), 3, 19, {'prior': True}),
    # BASELINE config 1: static-MNIST-shaped 3-layer, B=64. State dict is rebuilt from the seed by the test
    # (the build's constructor must reproduce the reference's default init bit for bit).
    'cfg1_mnist3': (cfg_of(z_dims=[32, 32, 32], downsample=[1, 1, 1], blocks_per_layer=2, n_filters=64), 64, 42,
                    {'seed_only': True}),
}


def synth_x(cfg, batch, seed):
    g = torch.Generator().manual_seed(1234 + seed)
    shape = (batch, cfg['color_ch']) + tuple(cfg['img_shape'])
    u = torch.rand(shape, generator=g)
    if cfg['likelihood_form'] == 'bernoulli':
        return (u > 0.5).float()
    x = torch.floor(256 * u) / 255
    # make sure the DMoL edge branches (x == 0 and x == 1) are exercised
    x.view(-1)[0::97] = 0.0
    x.view(-1)[1::89] = 1.0
    return x


def flat(prefix, v, store):
    if v is None:
        return
    if isinstance(v, (list, tuple)):
        for i, t in enumerate(v):
            flat('%s.%d' % (prefix, i), t, store)
    elif isinstance(v, dict):
        for k, t in v.items():
            flat('%s.%s' % (prefix, k), t, store)
    else:
        store[prefix] = torch.as_tensor(v).detach().cpu().numpy()


def run_case(name):
    cfg, batch, seed, extra = CASES[name]
    torch.manual_seed(seed)
    model = LadderVAE(**cfg)
    if not extra.get('seed_only'):
        # perturb BN affine/buffers and the top prior so that nothing is tested only at its default value
        g = torch.Generator().manual_seed(seed + 1000)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                if k.endswith('running_mean'):
                    v.copy_(0.1 * torch.randn(v.shape, generator=g))
                elif k.endswith('running_var'):
                    v.copy_(1.0 + 0.2 * torch.rand(v.shape, generator=g))
                elif k.endswith('top_prior_params'):
                    v.copy_(0.1 * torch.randn(v.shape, generator=g))
            for mod in model.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.copy_(1.0 + 0.1 * torch.randn(mod.weight.shape, generator=g))
                    mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
    store = {}
    if extra.get('seed_only'):
        store['init_seed'] = np.int64(seed)
    else:
        flat('sd', {k: v.clone() for k, v in model.state_dict().items()}, store)
    x = synth_x(cfg, batch, seed)
    store['x'] = x.numpy()
    if extra.get('prior'):  # before the training step: `sd` above is the state the samples come from
        model.eval()
        for tag, ml, cl in (('a', None, None), ('b', [0, 1], [2]), ('c', [0], [1, 2])):
            torch.manual_seed(seed + 7)
            rec = RecordRNG()
            with rec, torch.no_grad():
                s = model.sample_prior(3, ml, cl)
            flat('prior_%s.tape' % tag, rec.tape, store)
            store['prior_%s.sample' % tag] = s.numpy()
    model.train(not extra.get('eval', False))
    torch.manual_seed(seed + 1)
    rec = RecordRNG()
    with rec:
        out = model(x)
    flat('tape', rec.tape, store)
    flat('out', out, store)
    # experiment/experiment_manager.py:329-350
    recons_sep = -out['ll']
    elbo_sep = -(recons_sep + out['kl_sep'])
    loss = recons_sep.mean() + out['kl_loss']
    l2 = sum(torch.sum(p ** 2) for p in model.parameters()).sqrt()
    flat('fp', {'loss': loss, 'elbo': elbo_sep.mean(), 'elbo_sep': elbo_sep, 'recons': recons_sep.mean(), 'l2': l2},
         store)
    if model.training:
        opt = torch.optim.Adamax(model.parameters(), lr=3e-4, weight_decay=0.0)
        opt.zero_grad()
        loss.backward()
        named = dict(model.named_parameters())
        gsq = 0.0
        keys = list(named)
        if extra.get('seed_only'):
            # every 12th tensor plus the stem and head: enough to pin the backward without a 8 MB file
            keys = sorted(set(keys[::12] + keys[:2] + keys[-2:]))
        for k, p in named.items():
            if p.grad is not None:
                gsq += float(p.grad.double().pow(2).sum())
        for k in keys:
            if named[k].grad is not None:
                store['grad.' + k] = named[k].grad.numpy().copy()
        store['gradnorm'] = np.float64(gsq ** 0.5)
        opt.step()
        for k in keys:
            store['post.' + k] = named[k].detach().numpy().copy()
        for k, v in model.state_dict().items():
            if k.endswith('running_mean') or k.endswith('running_var'):
                if not extra.get('seed_only') or k.startswith('first_bottom_up') or k.startswith('final_top_down.1'):
                    store['bnpost.' + k] = v.numpy().copy()
    store['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **store)
    print('%-16s %7.1f KB  tape=%d  ll[0]=%.4f  loss=%.5f' % (name, os.path.getsize(path) / 1024, len(rec.tape),
                                                              float(out['ll'][0]), float(loss)))


def run_ops():
    """Per-function vectors for the likelihood edge cases (SURVEY.md §8c list)."""
    store = {}
    g = torch.Generator().manual_seed(5)
    # log_bernoulli incl. saturated probabilities (BCE log clamp at -100)
    mean = torch.rand(3, 1, 6, 6, generator=g)
    mean.view(-1)[0:4] = torch.tensor([0.0, 1.0, 1.0, 0.0])
    xb = (torch.rand(3, 1, 6, 6, generator=g) > 0.5).float()
    xb.view(-1)[0:4] = torch.tensor([1.0, 0.0, 1.0, 0.0])
    store['bern.mean'], store['bern.x'] = mean.numpy(), xb.numpy()
    store['bern.ll'] = ref_lik.log_bernoulli(xb, mean, reduce='none').numpy()
    # DMoL: edge pixels, tiny cdf_delta (large |x - mean| with small scale), clamped log-scales
    l = torch.randn(2, 100, 5, 5, generator=g)
    l[:, 40:50] = -9.0 + torch.rand(2, 10, 5, 5, generator=g)  # colour-1 log scales below the -7 clamp
    l[0, 10:20, 0, 0] = 30.0  # means far away -> cdf_delta < 1e-5 branch
    l[0, 20:30, 0, 0] = -6.0
    xd = torch.floor(256 * torch.rand(2, 3, 5, 5, generator=g)) / 255
    xd[0, :, 0, 1] = 0.0
    xd[0, :, 0, 2] = 1.0
    l.requires_grad_(True)
    ll = -ref_lik.discretized_mix_logistic_loss(xd * 2 - 1, l)
    ll.sum().backward()
    store['dmol.l'], store['dmol.x'], store['dmol.ll'] = l.detach().numpy(), xd.numpy(), ll.detach().numpy()
    store['dmol.dl'] = l.grad.numpy()
    torch.manual_seed(9)
    rec = RecordRNG()
    with rec, torch.no_grad():
        s = ref_stoch.sample_from_discretized_mix_logistic(l.detach())
    store['dmol.sample'] = s.numpy()
    flat('dmol.tape', rec.tape, store)
    # discretized logistic
    mean = torch.rand(2, 3, 4, 4, generator=g)
    ls = torch.randn(2, 3, 4, 4, generator=g) - 2
    xq = torch.floor(256 * torch.rand(2, 3, 4, 4, generator=g)) / 255
    xq.view(-1)[0], xq.view(-1)[1] = 0.0, 1.0
    store['dlog.mean'], store['dlog.ls'], store['dlog.x'] = mean.numpy(), ls.numpy(), xq.numpy()
    store['dlog.ll'] = ref_lik.log_discretized_logistic(xq * (255 / 256) + 1 / 512, mean, ls, reduce='none').numpy()
    path = os.path.join(OUT, 'ops.npz')
    np.savez_compressed(path, **store)
    print('ops              %7.1f KB' % (os.path.getsize(path) / 1024))


def run_forced():
    """topdown_pass(bu_values, forced_latent=[...]) of the reference (models/lvae.py:229-315, lib/stochastic.py:66-67): layers 0 and 2
    forced, layer 1 sampled; eval mode, so the tape holds the one normal draw and the likelihood draws are not involved."""
    cfg, batch, seed, _ = CASES['tiny_cifar']
    torch.manual_seed(seed)
    model = LadderVAE(**cfg)
    g = torch.Generator().manual_seed(seed + 2000)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('running_mean'):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif k.endswith('running_var'):
                v.copy_(1.0 + 0.2 * torch.rand(v.shape, generator=g))
            elif k.endswith('top_prior_params'):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
    store = {}
    flat('sd', {k: v.clone() for k, v in model.state_dict().items()}, store)
    x = synth_x(cfg, batch, seed)
    store['x'] = x.numpy()
    model.eval()
    with torch.no_grad():
        z_shapes = [tuple(z.shape) for z in model(x)['z']]
        forced = [0.7 * torch.randn(z_shapes[0], generator=g), None, 0.7 * torch.randn(z_shapes[2], generator=g)]
        bu = model.bottomup_pass(model.pad_input(x))
        torch.manual_seed(seed + 3)
        rec = RecordRNG()
        with rec:
            out, data = model.topdown_pass(bu, forced_latent=forced)
    flat('bu', bu, store)
    flat('forced', {str(i): f for i, f in enumerate(forced) if f is not None}, store)
    flat('tape', rec.tape, store)
    store['out'] = out.numpy()
    flat('data', {'z': data['z'], 'kl': data['kl'], 'kl_spatial': data['kl_spatial'], 'logprob_p': data['logprob_p']}, store)
    store['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(OUT, 'tiny_forced.npz')
    np.savez_compressed(path, **store)
    print('tiny_forced      %7.1f KB  tape=%d' % (os.path.getsize(path) / 1024, len(rec.tape)))


def run_stoch():
    """NormalStochasticBlock2d (lib/stochastic.py:7-112) called directly — every key of its data dict incl. `kl_elementwise`, Monte-Carlo
    and analytical, sampled / forced / mode, with input gradients through kl_elementwise — and kl_normal_mc (:209-226) with a
    batch-broadcast prior."""
    store = {}
    g = torch.Generator().manual_seed(21)
    torch.manual_seed(22)
    blk = ref_stoch.NormalStochasticBlock2d(c_in=8, c_vars=4, c_out=8)
    flat('sd', {k: v.clone() for k, v in blk.state_dict().items()}, store)
    p_in = torch.randn(3, 8, 4, 4, generator=g)
    q_in = torch.randn(3, 8, 4, 4, generator=g)
    forced = 0.5 * torch.randn(3, 4, 4, 4, generator=g)
    w_el = torch.randn(3, 4, 4, 4, generator=g)
    store['p_in'], store['q_in'], store['forced'], store['w_el'] = p_in.numpy(), q_in.numpy(), forced.numpy(), w_el.numpy()
    for tag, kw in (('mc', {}), ('an', {'analytical_kl': True}), ('forced', {'forced_latent': forced}), ('mode', {'use_mode': True})):
        pi, qi = p_in.clone().requires_grad_(True), q_in.clone().requires_grad_(True)
        torch.manual_seed(23)
        rec = RecordRNG()
        with rec:
            out, data = blk(pi, qi, **kw)
        blk.zero_grad()
        ((data['kl_elementwise'] * w_el).sum() + 0.1 * out.sum()).backward()
        flat(tag + '.tape', rec.tape, store)
        store[tag + '.out'] = out.detach().numpy()
        flat(tag + '.data', data, store)
        store[tag + '.dp_in'], store[tag + '.dq_in'] = pi.grad.numpy(), qi.grad.numpy()
        flat(tag + '.grad', {k: v.grad.clone() for k, v in blk.named_parameters()}, store)
    z = torch.randn(3, 4, 4, 4, generator=g)
    p1 = torch.randn(1, 8, 4, 4, generator=g)
    q3 = torch.randn(3, 8, 4, 4, generator=g)
    store['klmc.z'], store['klmc.p'], store['klmc.q'] = z.numpy(), p1.numpy(), q3.numpy()
    store['klmc.out'] = ref_stoch.kl_normal_mc(z, p1, q3).numpy()
    path = os.path.join(OUT, 'stoch.npz')
    np.savez_compressed(path, **store)
    print('stoch            %7.1f KB' % (os.path.getsize(path) / 1024))


def run_celeba():
    """CelebA input transform of experiment/data.py:76-80 — transforms.CenterCrop(148), transforms.Resize((64, 64)), ToTensor. torchvision
    is absent here; both transforms delegate to Pillow (`Image.crop` with torchvision's rounded offsets, `Image.resize(..., BILINEAR)`),
    which IS installed, so the expected outputs below come from Pillow itself on random and smooth 218x178 images."""
    from PIL import Image
    rng = np.random.default_rng(7)
    imgs = rng.integers(0, 256, (4, 218, 178, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:218, 0:178]
    imgs[3] = np.stack([yy * 255 / 217, xx * 255 / 177, (yy + xx) * 255 / 394], -1).astype(np.uint8)
    out = []
    for im in imgs:
        top, left = int(round((218 - 148) / 2.0)), int(round((178 - 148) / 2.0))   # torchvision.transforms.functional.center_crop
        pil = Image.fromarray(im).crop((left, top, left + 148, top + 148)).resize((64, 64), Image.BILINEAR)
        out.append(np.asarray(pil, dtype=np.uint8))
    path = os.path.join(OUT, 'celeba_resize.npz')
    np.savez_compressed(path, imgs=imgs, out=np.stack(out))
    print('celeba_resize    %7.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    torch.set_num_threads(1)  # one thread: the reference output is then bit-reproducible
    names = sys.argv[1:] or (list(CASES) + ['ops', 'tiny_forced', 'stoch', 'celeba_resize'])
    special = {'ops': run_ops, 'tiny_forced': run_forced, 'stoch': run_stoch, 'celeba_resize': run_celeba}
    for n in names:
        special[n]() if n in special else run_case(n)
