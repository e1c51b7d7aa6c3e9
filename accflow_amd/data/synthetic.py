"""Deterministic synthetic weights and frame sequences (numpy only, identical on every machine).

There is no network access for the CVO dataset or the released checkpoints, so parity tests, the golden
fixtures and bench.py all use: (1) name-keyed pseudo-random weights of the reference architecture
(He-scaled convs, non-trivial BatchNorm statistics, NON-zero ZeroConv2d so the deformable branch is
exercised) and (2) analytic moving-texture sequences with known ground-truth flow.
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()) & 0xFFFFFFFF, seed & 0xFFFFFFFF]))


def make_state_dict(model, seed=1234):
    """A full state_dict for `model` (any of RAFT / RAFTGMA / AccFlow, ours or the reference's):
    every tensor is a function of (key name, shape, seed) only."""
    out = {}
    for name, ref in model.state_dict().items():
        shape = tuple(ref.shape)
        g = _rng(name, seed)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            val = np.zeros(shape, dtype=np.int64)
        elif leaf == "rel_ind":
            out[name] = ref.clone()
            continue
        elif leaf == "running_var":
            val = g.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            val = g.normal(0.0, 0.1, shape)
        elif leaf == "gamma":
            val = np.full(shape, 0.5)
        elif leaf == "scale":                       # ZeroConv2d.scale
            val = g.normal(0.0, 0.1, shape)
        elif len(shape) == 4:                       # conv / deformable-conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if ".conv2.4.conv." in name:            # ZeroConv2d inner conv: keep offsets ~1 px
                gain = 0.25
            elif name.endswith("flow_head.conv2.weight") or name.endswith("flow_decoder.flow.2.weight"):
                gain = 0.1
            val = g.normal(0.0, gain * np.sqrt(2.0 / fan_in), shape)
        elif len(shape) == 2:                       # embeddings (never evaluated)
            val = g.normal(0.0, 1.0, shape)
        elif len(shape) == 1 and ".norm" in name and leaf == "weight" or \
                len(shape) == 1 and ".downsample.1" in name and leaf == "weight":
            val = g.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and (".norm" in name or ".downsample.1" in name) and leaf == "bias":
            val = g.normal(0.0, 0.1, shape)
        elif len(shape) == 1:                       # conv bias
            val = g.normal(0.0, 0.05, shape)
        else:
            val = g.normal(0.0, 0.05, shape)
        out[name] = torch.from_numpy(np.asarray(val)).to(ref.dtype)
    # shared modules (norm3 <-> downsample.1) must carry identical tensors under both names
    for name in list(out):
        if ".downsample.1." in name:
            twin = name.replace(".downsample.1.", ".norm3.")
            if twin in out:
                out[name] = out[twin].clone()
    return out


# ------------------------------------------------------------------------------------------------
# analytic moving texture


def _texture_params(seed, n_waves=28):
    g = _rng("texture", seed)
    lam = np.exp(g.uniform(np.log(6.0), np.log(220.0), n_waves))       # wavelengths in px
    ang = g.uniform(0, 2 * np.pi, n_waves)
    kx, ky = 2 * np.pi * np.cos(ang) / lam, 2 * np.pi * np.sin(ang) / lam
    amp = g.uniform(0.3, 1.0, (3, n_waves)) * (lam / lam.max()) ** 0.35
    ph = g.uniform(0, 2 * np.pi, (3, n_waves))
    return kx, ky, amp, ph


def _texture(qx, qy, params):
    kx, ky, amp, ph = params
    arg = qx[None, None] * kx[None, :, None, None] + qy[None, None] * ky[None, :, None, None]  # (1,K,H,W)
    val = (amp[:, :, None, None] * np.sin(arg + ph[:, :, None, None])).sum(axis=1)              # (3,H,W)
    norm = np.abs(amp).sum(axis=1)[:, None, None]
    return 127.5 + 127.5 * val / norm * 2.2


def _motion(t, H, W, shift=(3.0, -1.5), rot_deg=0.2):
    th = np.deg2rad(rot_deg) * t
    c, s = np.cos(th), np.sin(th)
    A = np.array([[c, -s], [s, c]])
    ctr = np.array([(W - 1) / 2.0, (H - 1) / 2.0])
    b = ctr - A @ ctr + t * np.array(shift)
    return A, b


def make_sequence(seed, n_frames, H, W, batch=1):
    """-> list of n_frames float32 tensors (batch,3,H,W) in [0,255]; sample s uses seed+s."""
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    frames = [[] for _ in range(n_frames)]
    for s in range(batch):
        params = _texture_params(seed + s)
        for t in range(n_frames):
            A, b = _motion(t, H, W)
            qx = A[0, 0] * xs + A[0, 1] * ys + b[0]
            qy = A[1, 0] * xs + A[1, 1] * ys + b[1]
            frames[t].append(np.clip(_texture(qx, qy, params), 0.0, 255.0))
    return [torch.from_numpy(np.stack(f).astype(np.float32)) for f in frames]


def gt_flow(i, j, H, W):
    """analytic flow frame i -> frame j, (2,H,W): pixel p of frame i shows scene point A_i p + b_i, which
    frame j shows at A_j^-1 (A_i p + b_i - b_j)."""
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    Ai, bi = _motion(i, H, W)
    Aj, bj = _motion(j, H, W)
    M = np.linalg.inv(Aj) @ Ai
    v = np.linalg.inv(Aj) @ (bi - bj)
    px = M[0, 0] * xs + M[0, 1] * ys + v[0]
    py = M[1, 0] * xs + M[1, 1] * ys + v[1]
    return torch.from_numpy(np.stack([px - xs, py - ys]).astype(np.float32))


def normalize(frame_0_255):
    """2*(x/255) - 1 exactly as test_cvo.py:41"""
    return 2 * (frame_0_255 / 255.0) - 1
