"""Validation data with the reference's batch-dict contract (data/dataset.py:72-108, 146-161):

    fetch_valid_dataloader(keys, split, batch) -> (loader, dataset); each batch is a dict
      "imgs"   (N, 21, H, W)  7 RGB frames, float32 in [0, 255]
      "fflows" (N, 10, H, W)  forward flows  F(0 -> i), i = 2..6
      "bflows" (N, 10, H, W)  backward flows F(i -> 0), i = 2..6

The CVO LMDB (Baidu / OneDrive download, lmdb + legacy pyarrow serialisation) is not available offline,
so the shipped dataset is the analytic moving-texture generator of accflow_amd.data.synthetic with exact
ground-truth flow.  `ACCFLOW_CVO_LMDB=<path>` is reserved for the real reader (not built yet).
"""
import os

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


def fetch_valid_dataloader(keys, split="clean", batch=1):
    if os.environ.get("ACCFLOW_CVO_LMDB"):
        raise NotImplementedError("the CVO LMDB reader is not part of this round (SURVEY 8(f) #1)")
    n = int(os.environ.get("ACCFLOW_SYNTH_SAMPLES", "20"))
    if "+" in split:
        dataset = SyntheticCVO(keys, "clean", n) + SyntheticCVO(keys, "final", n)
    else:
        dataset = SyntheticCVO(keys, split, n)
    loader = data.DataLoader(dataset, batch_size=batch, pin_memory=False, shuffle=False, num_workers=0, drop_last=False)
    return loader, dataset
