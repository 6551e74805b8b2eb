"""Experiment glue on the HIP engine — the hooks of the reference's experiment/experiment_manager.py:17-415
(`LVAEExperiment`) that the training hot path needs, re-hosted without `boilr` (absent):

  * the command-line flag surface of `_add_args` (:107-259) and the overridden defaults (:88-103), plus the boilr-owned
    flags seen at call sites (--seed, --batch-size, --lr, ...), plus engine flags (--synthetic, --steps, --no-graph);
  * `_check_args` (:262-290), `_make_run_description` (:293-320);
  * `_make_model` (:38-74), `_make_optimizer` (:76-81, Adamax) and `forward_pass` (:322-367);
  * `train_log_str` / `test_log_str` / `get_metrics_dict` (:369-415).

Out of scope: boilr's trainer loop, checkpoint rotation, tensorboard, dataset downloads. main.py drives
a minimal loop over synthetic batches (or an .npz file) with the same log lines.
"""
import argparse

import torch

from .. import engine
from ..models.lvae import LadderVAE
from ..noise import PhiloxNoise
from ..optim import Adamax

DATASETS = {
    # name: (color_ch, (H, W), default likelihood)  — experiment/data.py:32-97, experiment_manager.py:279-288
    'static_mnist': (1, (28, 28), 'bernoulli'),
    'cifar10': (3, (32, 32), 'discr_log_mix'),
    'svhn': (3, (32, 32), 'discr_log_mix'),
    'celeba': (3, (64, 64), 'discr_log_mix'),
    'multi_dsprites_binary_rgb': (3, (64, 64), 'bernoulli'),
    'multi_mnist_binary': (1, (64, 64), 'bernoulli'),
}


def build_parser():
    p = argparse.ArgumentParser(description='Ladder VAE on MI355X (HIP engine)', allow_abbrev=False,
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    # boilr-owned flags visible at the reference's call sites, with the defaults of experiment_manager.py:88-103
    p.add_argument('--batch-size', type=int, default=64, dest='batch_size')
    p.add_argument('--test-batch-size', type=int, default=1000, dest='test_batch_size')
    p.add_argument('--lr', type=float, default=3e-4)
    p.add_argument('--seed', type=int, default=54321)
    p.add_argument('--tr-log-every', type=int, default=10000, dest='train_log_every')
    p.add_argument('--ts-log-every', type=int, default=10000, dest='test_log_every')
    p.add_argument('--ts-img-every', type=int, default=-1, dest='test_imgs_every')
    p.add_argument('--checkpoint-every', type=int, default=100000, dest='checkpoint_every')
    p.add_argument('--keep-checkpoint-max', type=int, default=2, dest='keep_checkpoint_max')
    p.add_argument('--max-steps', type=int, default=10 ** 10, dest='max_steps')
    p.add_argument('--max-epochs', type=int, default=10 ** 7, dest='max_epochs')
    p.add_argument('--nocuda', action='store_true', dest='no_cuda')
    p.add_argument('--descr', type=str, default='', dest='additional_descr')
    p.add_argument('--dry-run', action='store_true', dest='dry_run')
    p.add_argument('--resume', type=str, default='')
    p.add_argument('--ll-every', type=int, default=50000, dest='loglikelihood_every')
    p.add_argument('--ll-samples', type=int, default=100, dest='loglikelihood_samples')
    # experiment_manager.py:107-259
    p.add_argument('-d', '--dataset', type=str, choices=list(DATASETS), default='static_mnist', dest='dataset_name')
    p.add_argument('--likelihood', type=str, choices=['bernoulli', 'gaussian', 'discr_log', 'discr_log_mix'], default=None)
    p.add_argument('--zdims', type=int, nargs='+', default=[32, 32, 32], dest='z_dims')
    p.add_argument('--blocks-per-layer', type=int, default=2, dest='blocks_per_layer')
    p.add_argument('--nfilters', type=int, default=64, dest='n_filters')
    p.add_argument('--no-bn', action='store_true', dest='no_batch_norm')
    p.add_argument('--skip', action='store_true', dest='skip_connections')
    p.add_argument('--gated', action='store_true', dest='gated')
    p.add_argument('--downsample', type=int, nargs='+', default=[1, 1, 1])
    p.add_argument('--learn-top-prior', action='store_true', dest='learn_top_prior')
    p.add_argument('--residual-type', type=str, default='bacdbacd', dest='residual_type')
    p.add_argument('--merge-layers', type=str, choices=['linear', 'residual'], default='residual', dest='merge_layers')
    p.add_argument('--beta-anneal', type=int, default=0, dest='beta_anneal')
    p.add_argument('--data-dep-init', action='store_true', dest='simple_data_dependent_init')
    p.add_argument('--wd', type=float, default=0.0, dest='weight_decay')
    p.add_argument('--nonlin', type=str, choices=['relu', 'leakyrelu', 'elu', 'selu'], default='elu')
    p.add_argument('--dropout', type=float, default=0.2)
    p.add_argument('--freebits', type=float, default=0.0, dest='free_bits')
    p.add_argument('--analytical-kl', action='store_true', dest='analytical_kl')
    p.add_argument('--no-initial-downscaling', action='store_true', dest='no_initial_downscaling')
    # engine flags (new)
    p.add_argument('--synthetic', action='store_true', help='train on synthetic batches (no dataset files needed)')
    p.add_argument('--data-npz', type=str, default='', help="train on an .npz with key 'data' (N,C,H,W) float32 in [0,1]")
    p.add_argument('--dtype', type=str, choices=['f32', 'bf16'], default='f32', dest='compute_dtype',
                   help='bf16: bf16 matrix-core operands, fp32 accumulate / statistics / KL / likelihood')
    p.add_argument('--steps', type=int, default=0, help='stop after this many steps (0: --max-steps)')
    p.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a captured hipGraph')
    p.add_argument('--log-every', type=int, default=100)
    p.add_argument('--save-checkpoint', type=str, default='', help='write a reference-layout checkpoint here at the end')
    return p


class LVAEExperiment:
    """Holds args, model, optimizer; `forward_pass` is the hot path (experiment_manager.py:322-367)."""

    def __init__(self, args=None, argv=None):
        if args is None:
            args = build_parser().parse_args(argv)
        self.args = self._check_args(args)
        self.run_description = self._make_run_description(self.args)
        if self.args.no_cuda or not torch.cuda.is_available():
            raise RuntimeError("the HIP engine needs an MI355X: no CPU path exists in the product (--nocuda is rejected)")
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.color_ch, self.img_size, _ = DATASETS[self.args.dataset_name]
        self.model = self._make_model()
        self.optimizer = self._make_optimizer()

    @classmethod
    def _check_args(cls, args):
        if len(args.z_dims) != len(args.downsample):
            raise RuntimeError("length of list of latent dimensions ({}) does not match length of list of downsampling "
                               "factors ({})".format(len(args.z_dims), len(args.downsample)))
        assert args.weight_decay >= 0.0
        assert 0.0 <= args.dropout <= 1.0
        if args.dropout < 1e-5:
            args.dropout = None
        assert args.free_bits >= 0.0
        args.batch_norm = not args.no_batch_norm
        if args.likelihood is None:
            args.likelihood = DATASETS[args.dataset_name][2]
        return args

    @staticmethod
    def _make_run_description(args):
        s = args.dataset_name
        s += ',{}ly'.format(len(args.z_dims))
        s += ',{}bpl'.format(args.blocks_per_layer)
        s += ',{}ch'.format(args.n_filters)
        if args.skip_connections:
            s += ',skip'
        if args.gated:
            s += ',gate'
        s += ',block=' + args.residual_type
        if args.beta_anneal != 0:
            s += ',b{}'.format(args.beta_anneal)
        s += ',{}'.format(args.nonlin)
        if args.free_bits > 0:
            s += ',freeb={}'.format(args.free_bits)
        if args.dropout is not None:
            s += ',drop={}'.format(args.dropout)
        if args.learn_top_prior:
            s += ',learnp'
        if args.weight_decay > 0.0:
            s += ',wd={}'.format(args.weight_decay)
        s += ',seed{}'.format(args.seed)
        if len(args.additional_descr) > 0:
            s += ',' + args.additional_descr
        return s

    def _make_model(self):
        a = self.args
        torch.manual_seed(a.seed)
        model = LadderVAE(self.color_ch, z_dims=a.z_dims, blocks_per_layer=a.blocks_per_layer, downsample=a.downsample,
                          merge_type=a.merge_layers, batchnorm=a.batch_norm, nonlin=a.nonlin,
                          stochastic_skip=a.skip_connections, n_filters=a.n_filters, dropout=a.dropout,
                          res_block_type=a.residual_type, free_bits=a.free_bits, learn_top_prior=a.learn_top_prior,
                          img_shape=self.img_size, likelihood_form=a.likelihood, gated=a.gated,
                          no_initial_downscaling=a.no_initial_downscaling, analytical_kl=a.analytical_kl).to(self.device)
        model.noise = PhiloxNoise(seed=a.seed)
        model.compute_dtype = a.compute_dtype
        return model

    def _make_optimizer(self):
        return Adamax(self.model, lr=self.args.lr, weight_decay=self.args.weight_decay)

    def beta(self):
        if self.args.beta_anneal != 0:
            return engine.linear_anneal(self.model.global_step, 0.0, 1.0, self.args.beta_anneal)
        return 1.0

    def forward_pass(self, x, y=None):
        x = x.to(self.device, non_blocking=True)
        return engine.forward_pass(self.model, x, self.beta())

    @classmethod
    def train_log_str(cls, summaries, step, epoch=None):
        s = "       [step {}]   loss: {:.5g}   ELBO: {:.5g}   recons: {:.3g}   KL: {:.3g}"
        return s.format(step, summaries['loss/loss'], summaries['elbo/elbo'], summaries['elbo/recons'], summaries['elbo/kl'])

    @classmethod
    def test_log_str(cls, summaries, step, epoch=None):
        s = "       "
        if epoch is not None:
            s += "[step {}, epoch {}]   ".format(step, epoch)
        s += "ELBO {:.5g}   recons: {:.3g}   KL: {:.3g}".format(summaries['elbo/elbo'], summaries['elbo/recons'],
                                                               summaries['elbo/kl'])
        for k in summaries.keys():
            if k.find('elbo_IW') > -1:
                s += "   marginal log-likelihood ({}) {:.5g}".format(k.split('_')[-1], summaries[k])
                break
        return s

    @classmethod
    def get_metrics_dict(cls, results):
        d = {'loss/loss': results['loss'].item(), 'elbo/elbo': results['elbo'].item(),
             'elbo/recons': results['recons'].item(), 'elbo/kl': results['kl'].item(), 'l2/l2': results['l2'].item()}
        if 'kl_avg_layerwise' in results:
            for i in range(len(results['kl_avg_layerwise'])):
                d['kl_layers/kl_layer_{}'.format(i)] = results['kl_avg_layerwise'][i].item()
        return d
