"""Input batches for the hot path (host side; counterpart of ldm/dataset.py:248-322,379-410).

The reference streams TFDS -> tf.data; neither is installed on the target image and the metric is
quoted on synthetic batches, so this module provides
  * 'synthetic'  : uniform random uint8 images, seeded per rank (bench / smoke / plumbing);
  * 'cifar10'    : the python-pickle CIFAR-10 archive (cifar-10-batches-py) read from
                   $MULAN_DATA_DIR, test split unshuffled for create_one_time_eval_dataset;
  * 'imagenet32' : the downsampled-ImageNet 32x32 python pickles of Chrabaszcz et al. (train_data_batch_1..10,
                   val_data: dict with 'data' [N, 3072] uint8 in channel-major order) under
                   $MULAN_DATA_DIR/Imagenet32_train and Imagenet32_val (or both in $MULAN_DATA_DIR/imagenet32).
                   NOTE: the reference reads TFDS `downsampled_imagenet/32x32` = the van den Oord et al. PNG archives,
                   a DIFFERENT downsampling of ImageNet; likelihoods on the two variants are not comparable.  For
                   numbers comparable with the reference / the paper, dump the Oord train_32x32 / valid_32x32 images
                   to an .npz and use 'npz:<file>'.  Labels are zero like the reference's (label_key=None);
  * 'npz:<path>' : any .npz with uint8 `images` [N,32,32,3] (e.g. a downsampled-ImageNet-32 dump).
Batch dict keys follow _preprocess_cifar10 (ldm/dataset.py:310-322): images u8, labels, conditioning.
"""
import os
import pickle

import numpy as np
import torch


def _cifar_split(root, train):
    d = os.path.join(root, "cifar-10-batches-py")
    names = [f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]
    xs, ys = [], []
    for n in names:
        with open(os.path.join(d, n), "rb") as f:
            e = pickle.load(f, encoding="bytes")
        xs.append(np.asarray(e[b"data"], dtype=np.uint8).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
        ys.append(np.asarray(e[b"labels"], dtype=np.int32))
    return np.concatenate(xs), np.concatenate(ys)


def _imagenet32_split(root, train):
    """downsampled ImageNet 32x32 as distributed by image-net.org (Chrabaszcz et al. box-resize pickles: dicts with the
    images flattened [N, 3 * 32 * 32] channel-major).  This is NOT the variant TFDS downsampled_imagenet/32x32 serves
    to the reference (van den Oord et al., ldm/dataset.py:187-199): see the module docstring.  Labels are returned as
    zeros, as the reference does for this dataset (label_key=None)."""
    cands = [os.path.join(root, "Imagenet32_train" if train else "Imagenet32_val"), os.path.join(root, "imagenet32")]
    d = next((c for c in cands if os.path.isdir(c)), None)
    if d is None:
        raise FileNotFoundError(
            "ImageNet-32 not found: set MULAN_DATA_DIR to the directory holding Imagenet32_train/ and Imagenet32_val/ "
            "(train_data_batch_1..10 / val_data pickles), or use --config.data.dataset=npz:<file> / synthetic")
    names = [f"train_data_batch_{i}" for i in range(1, 11)] if train else ["val_data"]
    xs, ys = [], []
    for n in names:
        path = os.path.join(d, n)
        if not os.path.isfile(path):
            if train and xs:
                break            # a partial download still trains
            raise FileNotFoundError(path)
        with open(path, "rb") as f:
            e = pickle.load(f, encoding="latin1")
        xs.append(np.asarray(e["data"], dtype=np.uint8).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
        ys.append(np.zeros(len(xs[-1]), dtype=np.int32))          # the reference drops the labels (label_key=None)
    return np.concatenate(xs), np.concatenate(ys)


def load_arrays(name, train):
    if name == "synthetic":
        return None, None
    if name.startswith("npz:"):
        z = np.load(name[4:])
        key = "images" if train or "test_images" not in z else "test_images"
        x = np.asarray(z[key], dtype=np.uint8)
        return x, np.zeros(len(x), dtype=np.int32)
    root = os.environ.get("MULAN_DATA_DIR", "")
    if name in ("cifar10", "cifar10_aug"):
        if not os.path.isdir(os.path.join(root, "cifar-10-batches-py")):
            raise FileNotFoundError(
                "CIFAR-10 not found: set MULAN_DATA_DIR to the directory holding cifar-10-batches-py/ "
                "(or use --config.data.dataset=synthetic)")
        return _cifar_split(root, train)
    if name == "imagenet32":
        return _imagenet32_split(root, train)
    raise NotImplementedError(f"dataset {name!r}: supported are synthetic, cifar10, imagenet32, npz:<file>")


class BatchStream:
    """Infinite (train/eval) or one-pass iterator of per-rank batches on `device`.

    Infinite streams: every rank walks the SAME permutation of the data set (drawn from a rank-independent generator
    seeded with (seed, train, epoch)) and takes the strided shard perm[pos * world + rank], so that one epoch over all
    ranks is one pass over the data set (the reference gives each host a disjoint split, ldm/dataset.py:66,264); the
    per-rank generator only draws synthetic images and augmentations.
    One pass (create_one_time_eval_dataset): `batch_size` is the GLOBAL batch of the reference; whole global batches
    are dealt to the ranks round-robin (batch k goes to rank k % world), the remainder is dropped like the reference's
    drop_remainder batching, so the evaluated image set and every per-batch statistic are the same for any world size.
    """

    def __init__(self, name, batch_size, *, train, device, seed=0, rank=0, world=1, substeps=None, one_pass=False):
        if not one_pass and batch_size % world != 0:
            raise ValueError("Batch size must be divisible by the number of devices")   # ldm/dataset.py:256-259
        self.local = batch_size if one_pass else batch_size // world
        self.device = device
        self.substeps = substeps
        self.one_pass = one_pass
        self.rank, self.world = rank, world
        self.seed, self.train = int(seed), bool(train)
        self.x, self.y = load_arrays(name, train)
        # cifar10_aug (ldm/dataset.py:125-131, 358-376): the train stream gets a random left/right flip and a random
        # rotation by 90 / 180 / 270 degrees, each with probability 1/2; `conditioning` flags augmented images
        self.augment = train and name == "cifar10_aug"
        self.gen = np.random.default_rng([int(seed), int(rank), int(train)])
        self.pos = 0
        self.epoch = 0
        self.shuffle = self.x is not None and not one_pass and train
        self.perm = None if self.x is None else self._permutation()

    def _permutation(self):
        if not self.shuffle:
            return np.arange(len(self.x))
        return np.random.default_rng([self.seed, int(self.train), self.epoch]).permutation(len(self.x))

    def __len__(self):
        if self.x is None or not self.one_pass:
            raise TypeError("infinite stream")
        nb = len(self.x) // self.local                   # global batches
        return (nb - self.rank + self.world - 1) // self.world

    def _take(self, n):
        if self.x is None:
            img = self.gen.integers(0, 256, size=(n, 32, 32, 3), dtype=np.uint8)
            lab = np.zeros(n, dtype=np.int32)
            return img, lab
        if self.one_pass:                                # whole global batch number pos * world + rank
            k = self.pos * self.world + self.rank
            if (k + 1) * n > len(self.x):
                return self.x[:0], self.y[:0]
            self.pos += 1
            return self.x[k * n:(k + 1) * n], self.y[k * n:(k + 1) * n]
        idx = []
        while len(idx) < n:
            # rank r reads a strided shard of the (shuffled) index stream shared by all ranks; the epoch ends for every
            # rank at the same position (the len % world tail is dropped, like the reference's per-host split), so the
            # ranks stay on the same permutation for any dataset size
            if self.pos >= len(self.perm) // self.world:
                self.pos = 0
                self.epoch += 1
                self.perm = self._permutation()
            idx.append(self.perm[self.pos * self.world + self.rank])
            self.pos += 1
        idx = np.asarray(idx, dtype=np.int64)
        return self.x[idx], self.y[idx]

    def seek(self, samples_drawn):
        """positions the infinite train stream as if `samples_drawn` samples per rank had been taken (resume from a
        checkpoint: state.step * batch per rank): permutation, position and -- because they are derived from the position --
        the cifar10_aug flips / rotations continue exactly where an uninterrupted run would be.  A no-op for synthetic
        data (drawn from the per-rank generator: a resumed run sees other random images) and one-pass streams; the eval
        stream is not repositioned (the reference's is not either: ldm/experiment.py:236-247 restarts it)."""
        if self.x is None or self.one_pass:
            return
        per_epoch = max(1, len(self.x) // self.world)
        self.epoch, self.pos = divmod(int(samples_drawn), per_epoch)
        self.perm = self._permutation()

    def _batch(self, n):
        img, lab = self._take(n)
        if len(img) < n:
            raise StopIteration
        return img, lab

    def __iter__(self):
        return self

    def __next__(self):
        s = self.substeps
        n = self.local * (s or 1)
        img, lab = self._batch(n)
        shape = (s, self.local) if s else (self.local,)
        cond = np.zeros(n, dtype=np.uint8)
        if self.augment:
            img = img.copy()
            # the draws are a function of the stream position of this batch (epoch, pos after the take), not of how many
            # batches this process has drawn: a run resumed with seek() sees the augmentations an uninterrupted run sees
            aug = np.random.default_rng([self.seed, int(self.rank), 7, int(self.epoch), int(self.pos)])
            flip = aug.random(n) > 0.5
            k = np.ceil(3.0 * aug.random(n)).astype(np.int64)
            rot = aug.random(n) > 0.5
            for i in range(n):
                if flip[i]:
                    img[i] = img[i][:, ::-1]
                if rot[i]:
                    img[i] = np.rot90(img[i], k=int(k[i]), axes=(0, 1))
            cond = (flip | rot).astype(np.uint8)
        images = torch.from_numpy(np.ascontiguousarray(img).reshape(*shape, 32, 32, 3)).to(self.device, non_blocking=True)
        labels = torch.from_numpy(lab.reshape(*shape)).to(self.device)
        return {"images": images, "labels": labels,
                "conditioning": torch.from_numpy(cond.reshape(*shape)).to(self.device)}

    next = __next__


def create_dataset(config, device, seed, rank=0, world=1):
    """(train_iter, eval_iter) like ldm/dataset.py:65-246 for the supported datasets."""
    name = config.data.dataset
    tr = config.training
    train = BatchStream(name, tr.batch_size_train, train=True, device=device, seed=seed, rank=rank, world=world,
                        substeps=tr.substeps)
    evl = BatchStream(name, tr.batch_size_eval, train=False, device=device, seed=seed + 1, rank=rank, world=world)
    return train, evl


def create_one_time_eval_dataset(config, batch_size=None, device="cpu", rank=0, world=1):
    """Unshuffled single pass over the test split in batches of `batch_size` (the reference's global batch,
    ldm/dataset.py:379-410; default config.training.batch_size_eval); whole batches are dealt round-robin to ranks."""
    if batch_size is None:
        batch_size = config.training.batch_size_eval
    return BatchStream(config.data.dataset, batch_size, train=False, device=device, rank=rank, world=world,
                       one_pass=True)
