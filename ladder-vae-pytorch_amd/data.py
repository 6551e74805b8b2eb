"""Data pipeline of the reference (experiment/data.py:17-115, lib/datasets.py:9-72) without torchvision and without
downloads (there is no network here): the same on-disk formats, the same tensors.

* static_mnist: `binarized_mnist_{train,valid,test}.npz` (key `data`, float32 (N,1,28,28)) under ./data/static_bin_mnist/;
  a Larochelle `.amat` text file next to a missing `.npz` is converted exactly as the reference does after its download
  (lib/datasets.py:55-64). train = train + valid concatenated (:14-18), labels are NaN (:21), both splits are shuffled once at
  load (`shuffle_init=True`, experiment/data.py:43-50).
* cifar10: the `cifar-10-batches-py` pickle batches under ./data/cifar10/ — uint8 HWC -> float CHW / 255 (`ToTensor`, :47-52).
* svhn: `train_32x32.mat` / `test_32x32.mat` under ./data/svhn/ (scipy.io) — same `ToTensor` scaling (:64).
* celeba: CenterCrop(148) + Resize(64) of the aligned JPEGs (:76-80) is NOT built (needs an image decoder on the input path).
Loaders: train shuffled with drop_last, test in order with `test_batch_size` (:99-106).
"""
import os
import pickle

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset


def amat_to_npz(path_amat, path_npz=None):
    """One line of 784 space-separated 0/1 per image -> float32 (N,1,28,28) saved compressed under key `data`."""
    with open(path_amat) as f:
        rows = [np.array(line.split(), dtype=np.float32) for line in f if line.strip()]
    x = np.stack(rows).reshape(-1, 1, 28, 28)
    if path_npz is None:
        path_npz = path_amat[:-len('.amat')] + '.npz'
    np.savez_compressed(path_npz, data=x)
    return x


class StaticBinaryMnist(TensorDataset):
    """lib/datasets.py:9-72."""

    def __init__(self, folder, train, shuffle_init=False):
        splits = ['train', 'valid'] if train else ['test']
        x = np.concatenate([self._load(folder, sp, shuffle_init) for sp in splits], axis=0)
        labels = torch.full((len(x),), float('nan'))
        super().__init__(torch.from_numpy(x), labels)

    @staticmethod
    def _load(folder, split, shuffle_init):
        npz = os.path.join(folder, 'binarized_mnist_%s.npz' % split)
        amat = os.path.join(folder, 'binarized_mnist_%s.amat' % split)
        if os.path.exists(npz):
            x = np.load(npz)['data']
        elif os.path.exists(amat):
            x = amat_to_npz(amat, npz)
        else:
            raise RuntimeError("Dataset file '%s' not found and nothing can be downloaded here: place binarized_mnist_%s.npz "
                               "(or the .amat it is made from) in %s" % (npz, split, folder))
        x = np.ascontiguousarray(x, dtype=np.float32)
        if shuffle_init:
            np.random.shuffle(x)
        return x


def _cifar10(folder, train):
    root = os.path.join(folder, 'cifar-10-batches-py')
    names = ['data_batch_%d' % i for i in range(1, 6)] if train else ['test_batch']
    xs, ys = [], []
    for n in names:
        path = os.path.join(root, n)
        if not os.path.exists(path):
            raise RuntimeError("CIFAR10 batch '%s' not found and nothing can be downloaded here" % path)
        with open(path, 'rb') as f:
            d = pickle.load(f, encoding='latin1')
        xs.append(np.asarray(d['data'], dtype=np.uint8).reshape(-1, 3, 32, 32))
        ys.append(np.asarray(d['labels'] if 'labels' in d else d['fine_labels'], dtype=np.int64))
    x = torch.from_numpy(np.concatenate(xs)).float().div_(255.0)   # ToTensor: 0, 1/255, ..., 1
    return TensorDataset(x, torch.from_numpy(np.concatenate(ys)))


def _svhn(folder, train):
    from scipy.io import loadmat
    path = os.path.join(folder, '%s_32x32.mat' % ('train' if train else 'test'))
    if not os.path.exists(path):
        raise RuntimeError("SVHN file '%s' not found and nothing can be downloaded here" % path)
    m = loadmat(path)
    x = torch.from_numpy(np.transpose(m['X'], (3, 2, 0, 1)).copy()).float().div_(255.0)
    y = torch.from_numpy(m['y'].astype(np.int64).squeeze() % 10)
    return TensorDataset(x, y)


FOLDERS = {'static_mnist': './data/static_bin_mnist/', 'cifar10': './data/cifar10/', 'svhn': './data/svhn/'}


class DatasetLoader:
    """experiment/data.py:17-115: `.train`, `.test` (DataLoaders), `.data_shape`, `.img_size`, `.color_ch`."""

    def __init__(self, args, folder=None):
        name = args.dataset_name
        folder = folder or FOLDERS.get(name)
        if name == 'static_mnist':
            train_set = StaticBinaryMnist(folder, train=True, shuffle_init=True)
            test_set = StaticBinaryMnist(folder, train=False, shuffle_init=True)
        elif name == 'cifar10':
            train_set, test_set = _cifar10(folder, True), _cifar10(folder, False)
        elif name == 'svhn':
            train_set, test_set = _svhn(folder, True), _svhn(folder, False)
        else:
            raise RuntimeError("data set '%s' has no loader in this build (static_mnist, cifar10, svhn; or --data-npz / --synthetic)" % name)
        self.train = DataLoader(train_set, batch_size=args.batch_size, shuffle=True, drop_last=True)
        self.test = DataLoader(test_set, batch_size=args.test_batch_size, shuffle=False)
        self.data_shape = self.train.dataset[0][0].size()
        self.img_size = self.data_shape[1:]
        self.color_ch = self.data_shape[0]
