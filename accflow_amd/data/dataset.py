"""Validation data with the reference's batch-dict contract (data/dataset.py:72-108, 146-161):

    fetch_valid_dataloader(keys, split, batch) -> (loader, dataset); each batch is a dict
      "imgs"   (N, 21, H, W)  7 RGB frames, float32 in [0, 255]
      "fflows" (N, 10, H, W)  forward flows  F(0 -> i), i = 2..6
      "bflows" (N, 10, H, W)  backward flows F(i -> 0), i = 2..6

Two sources:
  * the CVO LMDB (data/README.md of the reference; `cvo_test.lmdb`, 536 sequences of 7 x 512 x 512): `CVO_sampler_lmdb`
    / `CVO` below mirror data/dataset.py:23-108 - key scheme `{index:05d}_{key}`, values in pyarrow's legacy
    serialisation, flows uint16-coded as (v - 2^15) / 128 (:60-67), HWC -> CHW float (`totensor`, :19-20) - on the
    pure-Python LMDB reader and legacy-pyarrow decoder of this package (neither `lmdb` nor `pa.deserialize` exists in
    the image).  Looked up at $ACCFLOW_CVO_LMDB (the .lmdb directory, its data.mdb, or a directory holding
    cvo_test.lmdb) and at the reference's location data/datasets/CVO_full/cvo_test.lmdb under the repo root;
  * when neither exists: the analytic moving-texture generator of accflow_amd.data.synthetic with exact ground-truth
    flow (same contract).  This is NOT CVO - EPEs printed on it are not CVO numbers - so the fallback announces itself
    on stderr unless ACCFLOW_SYNTHETIC=1 asks for it explicitly.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
from torch.utils import data

from .synthetic import gt_flow, make_sequence

ALL_KEYS = ["fflows", "bflows", "delta_fflows", "delta_bflows"]


class SyntheticCVO(data.Dataset):
    all_keys = ALL_KEYS

    def __init__(self, keys=None, split="clean", n_samples=20, size=(512, 512), n_frames=7):
        keys = list(self.all_keys) if keys is None else [k.lower() for k in keys]
        for k in keys:
            assert k in self.all_keys, f"Invalid key value: {k}"
        if any(k.startswith("delta_") for k in keys):
            raise NotImplementedError("delta flows are training-only keys (train_acc.py), not on the inference path")
        self.keys, self.split = keys, split
        self.n_samples, self.size, self.n_frames = n_samples, size, n_frames
        H, W = size
        self._f = torch.cat([gt_flow(0, i, H, W) for i in range(2, n_frames)], 0)
        self._b = torch.cat([gt_flow(i, 0, H, W) for i in range(2, n_frames)], 0)

    def __len__(self):
        return self.n_samples

    def __getitem__(self, index):
        H, W = self.size
        seed = 5000 + index + (100000 if self.split != "clean" else 0)
        frames = make_sequence(seed, self.n_frames, H, W, batch=1)
        out = {}
        if "fflows" in self.keys:
            out["fflows"] = self._f.clone()
        if "bflows" in self.keys:
            out["bflows"] = self._b.clone()
        out["imgs"] = torch.cat([f[0] for f in frames], 0)
        return out


def totensor(x):
    """HWC numpy -> CHW float tensor (data/dataset.py:19-20)"""
    return torch.from_numpy(x).permute(2, 0, 1).float()


def find_cvo_lmdb(is_training=False):
    """Path of cvo_{test,train}.lmdb, or None."""
    name = "cvo_train.lmdb" if is_training else "cvo_test.lmdb"
    cands = []
    env = os.environ.get("ACCFLOW_CVO_LMDB")
    if env:
        cands += [env, os.path.join(env, name)]
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cands.append(os.path.join(root, "data", "datasets", "CVO_full", name))   # the reference's location (dataset.py:29-34)
    for c in cands:
        if os.path.isfile(c) or os.path.isfile(os.path.join(c, "data.mdb")):
            return c
    if env:
        raise FileNotFoundError("ACCFLOW_CVO_LMDB=%s holds no %s (data.mdb)" % (env, name))
    return None


class CVO_sampler_lmdb:
    """Data sampling (data/dataset.py:23-69)."""

    all_keys = ["imgs", "imgs_blur", "fflows", "bflows", "delta_fflows", "delta_bflows"]

    def __init__(self, is_training=True, keys=None, db_path=None):
        from .lmdb_reader import ReadOnlyLMDB
        from .pa_legacy import deserialize
        self.db_path = db_path or find_cvo_lmdb(is_training)
        if self.db_path is None:
            raise FileNotFoundError("CVO LMDB not found (set ACCFLOW_CVO_LMDB)")
        self._deserialize = deserialize
        self.env = ReadOnlyLMDB(self.db_path)
        raw = self.env.get(b"__samples__")
        if raw is None:
            raise RuntimeError("%s: no __samples__ record - not a CVO LMDB" % self.db_path)
        self.samples = deserialize(raw)
        self.length = len(self.samples)
        self.keys = self.all_keys if keys is None else [x.lower() for x in keys]
        self._check_keys(self.keys)

    def _check_keys(self, keys):
        for k in keys:
            assert k in self.all_keys, f"Invalid key value: {k}"

    def __len__(self):
        return self.length

    def sample(self, index):
        sample = OrderedDict()
        for k in self.keys:
            key = "{:05d}_{:s}".format(index, k)
            raw = self.env.get(key.encode())
            if raw is None:
                raise KeyError(key)
            value = self._deserialize(raw)
            # what a CVO record must be (data/README.md of the reference): (H, W, 3*7) uint8 frames, (H, W, 2*n) uint16
            # flow codes - a mis-decoded legacy-pyarrow stream or a foreign LMDB fails here, not as garbage EPEs
            want = np.uint16 if "flow" in k else np.uint8
            if not (isinstance(value, np.ndarray) and value.ndim == 3 and value.dtype == want
                    and value.shape[2] % (2 if "flow" in k else 3) == 0 and (("flow" in k) or value.shape[2] == 21)):
                raise RuntimeError("%s: record %s is %s, expected an (H, W, %s) %s array" % (
                    self.db_path, key, (getattr(value, "shape", None), getattr(value, "dtype", type(value))),
                    "2n" if "flow" in k else "21", np.dtype(want).name))
            if "flow" in key:  # uint16 code -> float (dataset.py:65-67)
                value = value.astype(np.float32)
                value = (value - 2 ** 15) / 128.0
            sample[k] = value
        return sample


class FlowAugmentor:
    """The training augmentation (data/augmentor.py:4-26): ONE random crop window applied to every array of the sample
    (frames and flows are (H, W, C) arrays of the same H x W); numpy's global generator, like the reference."""

    def __init__(self, size):
        self.crop_size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, sample_dict):
        ht, wd = list(sample_dict.values())[0].shape[:2]
        ch, cw = self.crop_size
        y0 = np.random.randint(0, ht - ch)      # (exclusive upper bound: a full-size crop raises, as in the reference)
        x0 = np.random.randint(0, wd - cw)
        for k, v in sample_dict.items():
            sample_dict[k] = v[y0:y0 + ch, x0:x0 + cw, :]
        return sample_dict


class CVO(data.Dataset):
    """data/dataset.py:72-108."""

    all_keys = ["fflows", "bflows", "delta_fflows", "delta_bflows"]

    def __init__(self, keys=None, split="clean", is_training=False, crop_size=256, db_path=None):
        self.augmentor = FlowAugmentor(crop_size) if is_training else None
        keys = list(self.all_keys) if keys is None else [x.lower() for x in keys]
        self._check_keys(keys)
        keys.append("imgs" if split == "clean" else "imgs_blur")
        self.sampler = CVO_sampler_lmdb(is_training, keys, db_path=db_path)

    def __getitem__(self, index):
        sample_dict = self.sampler.sample(index)
        if self.augmentor is not None:
            sample_dict = self.augmentor(sample_dict)
        out_dict = {}
        for k, v in sample_dict.items():
            v_ = totensor(np.ascontiguousarray(v).copy())
            out_dict["imgs" if "imgs" in k else k] = v_
        return out_dict

    def _check_keys(self, keys):
        for k in keys:
            assert k in self.all_keys, f"Invalid key value: {k}"

    def __len__(self):
        return len(self.sampler)


class _Shard(data.Sampler):
    """Per-epoch seeded permutation of the dataset, rank r of `world` takes elements r, r + world, ... of it - the same
    number on every rank (the tail that does not fill all ranks is dropped), so that the ranks of a data-parallel run
    stay in step.  world = 1: a plain shuffle.  set_epoch(e, skip=k): the NEXT pass over the sampler starts behind the
    first k indices of epoch e's permutation (a resume from the middle of an epoch: nothing is decoded for the skipped
    samples - ADVICE r05; iterating and discarding them read and augmented every skipped LMDB record)."""

    def __init__(self, n, rank=0, world=1, seed=0):
        self.n, self.rank, self.world, self.seed, self.epoch, self.skip = n, rank, world, seed, 0, 0

    def set_epoch(self, epoch, skip=0):
        self.epoch, self.skip = epoch, int(skip)

    def __len__(self):
        return self.n // self.world

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        perm = torch.randperm(self.n, generator=g).tolist()
        mine = perm[self.rank:(self.n // self.world) * self.world:self.world]
        skip, self.skip = self.skip, 0        # (one pass only: the following epochs are whole)
        return iter(mine[skip:])


def fetch_train_dataloader(keys, batch=16, crop_size=256, split="clean", workers=0, rank=0, world=1, seed=0):
    """data/dataset.py:111-143: shuffled, drop_last, random 256 x 256 crops of cvo_train.lmdb ('clean+final' = both
    passes).  One process per GPU instead of nn.DataParallel: `batch` is the PER-RANK batch and the shuffle is sharded
    over `world` ranks (loader.sampler.set_epoch(e) reshuffles).  Without the LMDB: synthetic sequences of the crop size
    (announced on stderr, as in fetch_valid_dataloader)."""
    db = find_cvo_lmdb(True)
    if db is not None:
        make = lambda sp: CVO(keys=list(keys), is_training=True, split=sp, crop_size=crop_size, db_path=db)  # noqa: E731
    else:
        if os.environ.get("ACCFLOW_SYNTHETIC", "0") != "1":
            print("accflow_amd.data: NO CVO training LMDB found (ACCFLOW_CVO_LMDB unset, data/datasets/CVO_full/cvo_train.lmdb "
                  "absent) - training on SYNTHETIC moving-texture sequences (set ACCFLOW_SYNTHETIC=1 to silence this).",
                  file=sys.stderr, flush=True)
        n = int(os.environ.get("ACCFLOW_SYNTH_SAMPLES", "20"))
        size = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
        make = lambda sp: SyntheticCVO(keys, sp, n, size=size)  # noqa: E731
    dataset = make("clean") + make("final") if "+" in split else make(split)
    sampler = _Shard(len(dataset), rank, world, seed)
    loader = data.DataLoader(dataset, batch_size=batch, pin_memory=False, sampler=sampler, num_workers=workers, drop_last=True)
    return loader, dataset


def fetch_valid_dataloader(keys, split="clean", batch=1):
    """data/dataset.py:146-161."""
    db = find_cvo_lmdb(False)
    if db is not None:
        make = lambda sp: CVO(keys=list(keys), is_training=False, split=sp, db_path=db)  # noqa: E731
    else:
        if os.environ.get("ACCFLOW_SYNTHETIC", "0") != "1":
            print("accflow_amd.data: NO CVO LMDB found (ACCFLOW_CVO_LMDB unset, data/datasets/CVO_full/cvo_test.lmdb absent) - "
                  "serving the SYNTHETIC moving-texture sequences instead; EPEs printed on them are NOT CVO results "
                  "(set ACCFLOW_SYNTHETIC=1 to silence this).", file=sys.stderr, flush=True)
        n = int(os.environ.get("ACCFLOW_SYNTH_SAMPLES", "20"))
        make = lambda sp: SyntheticCVO(keys, sp, n)  # noqa: E731
    dataset = make("clean") + make("final") if "+" in split else make(split)
    loader = data.DataLoader(dataset, batch_size=batch, pin_memory=False, shuffle=False, num_workers=0, drop_last=False)
    return loader, dataset
