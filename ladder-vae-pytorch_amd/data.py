"""Data pipeline of the reference (experiment/data.py:17-115, lib/datasets.py:9-72) without torchvision and without
downloads (there is no network here): the same on-disk formats, the same tensors.

* static_mnist: `binarized_mnist_{train,valid,test}.npz` (key `data`, float32 (N,1,28,28)) under ./data/static_bin_mnist/;
  a Larochelle `.amat` text file next to a missing `.npz` is converted exactly as the reference does after its download
  (lib/datasets.py:55-64). train = train + valid concatenated (:14-18), labels are NaN (:21), both splits are shuffled once at
  load (`shuffle_init=True`, experiment/data.py:43-50).
* cifar10: the `cifar-10-batches-py` pickle batches under ./data/cifar10/ — uint8 HWC -> float CHW / 255 (`ToTensor`, :47-52).
* svhn: `train_32x32.mat` / `test_32x32.mat` under ./data/svhn/ (scipy.io) — same `ToTensor` scaling (:64).
* celeba: pre-decoded aligned images `celeba_aligned_uint8.npy` ((N,218,178,3) uint8, memory-mapped) + the stock
  `list_eval_partition.txt` under ./data/celeba/ — CenterCrop(148) + Resize((64,64)) + ToTensor (:76-80) restated on the arrays:
  torchvision's crop offsets (round((218-148)/2) = 35, round((178-148)/2) = 15) and Pillow's two-pass fixed-point antialiased
  bilinear resize, bit for bit (tests/golden/celeba_resize.npz was made with Pillow itself). JPEG decoding is not on this path.
Loaders: train shuffled with drop_last, test in order with `test_batch_size` (:99-106).
"""
import os
import pickle

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, TensorDataset


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


# ---------------------------------------------------------------------------------------------------------------------
# CelebA (experiment/data.py:76-89): transforms.CenterCrop(148) -> transforms.Resize((64, 64)) -> transforms.ToTensor()
# ---------------------------------------------------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2  # Pillow's 8-bit resampler: coefficients in 22-bit fixed point


def _pil_bilinear_coeffs(in_size, out_size):
    """Pillow `precompute_coeffs` + `normalize_coeffs_8bpc` for the BILINEAR filter over the whole axis: per output position the first
    source index, the tap count and the integer taps (antialiased: the triangle is widened by the downscale factor)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        w = np.array([max(0.0, 1.0 - abs((x + lo - center + 0.5) / fscale)) for x in range(hi - lo)], dtype=np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        xmin[xx] = lo
        kk[xx, :hi - lo] = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64)
    return xmin, kk


def _pil_resample_axis(a, axis, out_size):
    """One pass of Pillow's 8-bit resampler along `axis` of a uint8 array: sum of integer taps, rounded, clipped to [0, 255]."""
    xmin, kk = _pil_bilinear_coeffs(a.shape[axis], out_size)
    idx = np.minimum(xmin[:, None] + np.arange(kk.shape[1])[None, :], a.shape[axis] - 1)   # taps beyond the edge have weight 0
    g = np.take(a, idx.reshape(-1), axis=axis).astype(np.int64)
    shp = list(a.shape)
    shp[axis:axis + 1] = [out_size, kk.shape[1]]
    g = g.reshape(shp)
    kshape = [1] * g.ndim
    kshape[axis], kshape[axis + 1] = out_size, kk.shape[1]
    acc = (g * kk.reshape(kshape)).sum(axis=axis + 1) + (1 << (_PRECISION_BITS - 1))
    return np.clip(acc >> _PRECISION_BITS, 0, 255).astype(np.uint8)


def celeba_transform_uint8(imgs, crop=148, size=64):
    """(N,H,W,3) uint8 aligned CelebA images -> (N,size,size,3) uint8: torchvision CenterCrop(crop) then Pillow's
    Image.resize((size, size), BILINEAR) — horizontal pass first, then vertical, each rounded to 8 bits as Pillow does."""
    imgs = np.asarray(imgs)
    H, W = imgs.shape[1:3]
    top, left = int(round((H - crop) / 2.0)), int(round((W - crop) / 2.0))
    c = imgs[:, top:top + crop, left:left + crop, :]
    return _pil_resample_axis(_pil_resample_axis(c, 2, size), 1, size)


class CelebAArrays(Dataset):
    """torchvision.datasets.CelebA(split=...) + the transform above, from pre-decoded arrays. Items: (float32 (3,64,64) in [0,1], 0).
    The resized set is kept as uint8 (2.5 GB for the 202,599 images); ToTensor's /255 happens per item, as in the reference."""

    SPLITS = {'train': 0, 'valid': 1, 'test': 2}

    def __init__(self, folder, split, chunk=2048):
        arr = os.path.join(folder, 'celeba_aligned_uint8.npy')
        part = os.path.join(folder, 'list_eval_partition.txt')
        if not (os.path.exists(arr) and os.path.exists(part)):
            raise RuntimeError("CelebA needs '%s' ((N,218,178,3) uint8, the aligned images decoded once) and '%s'; nothing can be "
                               "downloaded or JPEG-decoded here" % (arr, part))
        raw = np.load(arr, mmap_mode='r')
        with open(part) as f:
            which = np.array([int(line.split()[1]) for line in f if line.strip()], dtype=np.int64)
        if len(which) != raw.shape[0]:
            raise RuntimeError("CelebA: %d partition lines for %d images" % (len(which), raw.shape[0]))
        idx = np.nonzero(which == self.SPLITS[split])[0]
        out = np.empty((len(idx), 64, 64, 3), dtype=np.uint8)
        for i in range(0, len(idx), chunk):
            out[i:i + chunk] = celeba_transform_uint8(raw[idx[i:i + chunk]])
        self.data = torch.from_numpy(out)

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        return self.data[i].permute(2, 0, 1).float().div_(255.0), 0


FOLDERS = {'static_mnist': './data/static_bin_mnist/', 'cifar10': './data/cifar10/', 'svhn': './data/svhn/',
           'celeba': './data/celeba/'}


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
        elif name == 'celeba':
            train_set, test_set = CelebAArrays(folder, 'train'), CelebAArrays(folder, 'valid')
        else:
            raise RuntimeError("data set '%s' has no loader in this build (static_mnist, cifar10, svhn, celeba; or --data-npz / "
                               "--synthetic)" % name)
        self.train = DataLoader(train_set, batch_size=args.batch_size, shuffle=True, drop_last=True)
        self.test = DataLoader(test_set, batch_size=args.test_batch_size, shuffle=False)
        self.data_shape = self.train.dataset[0][0].size()
        self.img_size = self.data_shape[1:]
        self.color_ch = self.data_shape[0]
