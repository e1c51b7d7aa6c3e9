"""Tensor-level wrappers over the C-ABI: take torch CUDA (ROCm) tensors, pass raw device pointers,
batch strides and the current HIP stream to libaccflow_hip.  PyTorch is used only for device memory
and streams; no torch operator computes anything here.

Every function raises RuntimeError when a tensor is not a float32 CUDA tensor or when the kernel
library reports an error - there is no CPU / eager fallback.
"""
import contextlib
import ctypes
import functools
import threading

import torch

from . import _lib, profiler
from ._lib import ConvDesc

import os

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
CONV_F32, CONV_BF16X3, CONV_BF16X6, CONV_F16X3 = 0, 2, 3, 4
_MODES = {"f32": CONV_F32, "bf16x3": CONV_BF16X3, "bf16x6": CONV_BF16X6, "f16x3": CONV_F16X3}
# arithmetic of the MFMA conv kernel: exact fp32 MFMA, or fp32 operands split into 2 / 3 bf16 terms on
# the bf16 matrix cores with fp32 accumulation (see csrc/conv2d.hip).
# Default f16x3: the direct conv kernel splits every fp32 operand into fp16 hi + lo of x * 2^s (3 MFMAs per product;
# 22 significant bits per operand thanks to power-of-two row / activation scales, see include/accflow_hip.h; measured
# 2.4e-5 px EPE vs the reference on C3) with a range guard - see with_range_guard(); every other kernel then runs bf16x6.
# "bf16x6" = 3 bf16 terms, 6 MFMAs, unconditionally fp32-equivalent (2.0e-5 px), 1.35x slower; "bf16x3" = ~2^-16 per
# product (1.7e-4 px); "f32" = bit-exact fp32 fmaf chains on the fp32 MFMA.
# CONV_MODE is the process-wide DEFAULT (configuration: set it before running, not concurrently with forwards); the mode
# a call really uses is current_mode(): a per-thread override (conv_mode() context, the guard's bf16x6 retry) wins, so
# one thread's retry never changes what another thread (nn.DataParallel: one per GPU) computes.
CONV_MODE = _MODES[os.environ.get("ACCFLOW_CONV_MODE", "f16x3").lower()]
_tls = threading.local()


USE_PATCH = os.environ.get("ACCFLOW_CONV_PATCH", "1") == "1"


def set_conv_mode(name):
    global CONV_MODE
    CONV_MODE = _MODES[name.lower()]


def current_mode():
    md = getattr(_tls, "mode", None)
    return CONV_MODE if md is None else md


def conv_mode_name():
    return {v: k for k, v in _MODES.items()}[current_mode()]


class conv_mode:
    """with ops.conv_mode("bf16x6"): ...   - per-thread override of the conv arithmetic."""

    def __init__(self, name):
        self.md = _MODES[name.lower()] if isinstance(name, str) else int(name)

    def __enter__(self):
        self.saved = getattr(_tls, "mode", None)
        _tls.mode = self.md

    def __exit__(self, *exc):
        _tls.mode = self.saved

EPI_STORE, EPI_RES_RELU, EPI_GRU_ZR, EPI_GRU_Q, EPI_ACCUM, EPI_TAPGEMM = 0, 1, 2, 3, 4, 5


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("libaccflow_hip: %s failed with hipError %d" % (name, rc))


def _plane4(t, name):
    """(B, C, H, W) fp32 CUDA tensor whose (C, H, W) block is dense; returns the batch stride."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise RuntimeError("%s: expected a 4-D float32 CUDA tensor, got %s" % (
            name, (tuple(t.shape), t.dtype, t.device) if isinstance(t, torch.Tensor) else type(t)))
    B, C, H, W = t.shape
    st = t.stride()
    if not (st[3] == 1 and (H == 1 or st[2] == W) and (C == 1 or st[1] == H * W)):
        raise RuntimeError("%s: channel block must be dense NCHW (strides %s)" % (name, st))
    return st[0] if B > 1 else C * H * W


def _dense(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError("%s: expected a contiguous float32 CUDA tensor" % name)
    return t


USE_DEFORM_COLUMNS = os.environ.get("ACCFLOW_DEFORM_COLUMNS", "1") == "1"
USE_TAPSUM = os.environ.get("ACCFLOW_CONV_TAPSUM", "1") == "1"
# The <= 4-channel regressions (flow heads, blending mask) take the fp16 split too.  Round 1 kept them on bf16x6 because
# its UNSCALED fp16 split raised the C3 EPE from 2.4e-5 to 6.3e-5 px; with the row / activation scales the mean EPE is
# unchanged (1.8e-5 px, max 5.0e-4 vs 4.6e-4) and the step is 0.25 ms shorter.  ACCFLOW_TAPSUM_F16=0: bf16x6 again.
TAPSUM_F16 = os.environ.get("ACCFLOW_TAPSUM_F16", "1") == "1"
# every pack of a convolution in one launch (accflow_conv_pack_all_f32); 0: one launch per pack (round 1-4 behaviour, A/B, tests)
PACK_FUSED = os.environ.get("ACCFLOW_PACK_FUSED", "1") == "1"
TAPGEMM_MAXROWS = 18          # ACCFLOW_EPI_TAPGEMM: rows of the second product (3x3 taps x 2 flow channels)
# FlowHead as ONE convolution launch + the tap sum (conv2d_tapgemm); 0: conv1 -> S16 tensor -> 18-row 1x1 conv -> tap sum
FUSE_TAPGEMM = os.environ.get("ACCFLOW_FUSE_FLOWHEAD", "1") == "1"
TAPSUM_MIN_PIXELS = 4096      # below this the dedicated small-Cout kernels are as fast
USE_KSPLIT = os.environ.get("ACCFLOW_CONV_KSPLIT", "1") == "1"
KSPLIT_MAX_PIXELS = 4 * 7680  # B*OH*OW up to which a split-K workspace is offered (the C side decides whether to split)


def _ksplit_on():
    return USE_KSPLIT and not getattr(_tls, "no_ksplit", False)


@contextlib.contextmanager
def ksplit_scope(enabled):
    """Split-K workspaces on / off for the convolutions this thread launches inside the scope.  Split-K buys a lone small launch
    (the batch-1 fusion chain: 60-120 workgroups) a full chip at the price of partial sums written and reduced; when another
    stream fills the chip anyway (parallel.SequencePipeline) the plain launch is the cheaper one.  Results differ in the order
    of fp32 sums only."""
    saved = getattr(_tls, "no_ksplit", False)
    _tls.no_ksplit = not enabled
    try:
        yield
    finally:
        _tls.no_ksplit = saved


def _ksplit_ws(n, device):
    """One scratch buffer per (host thread, device, stream): kernels of one stream issued by one thread are ordered,
    so the buffer can be reused launch after launch; two threads never share one (their launches interleave)."""
    store = _tls.__dict__.setdefault("ksplit_ws", {})
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = store.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(n, dtype=torch.float32, device=device)
        store[key] = t
    return t


def _guard(device):
    """Device int32 the f16x3 kernels OR with 1 when a value does not fit the scaled fp16 range: one flag per
    (host thread, device), passed to the library with every call (the library keeps no state)."""
    scoped = getattr(_tls, "guard_flag", None)
    if scoped is not None:     # guard_scope: the caller owns the flag (and reads it itself)
        return scoped
    store = _tls.__dict__.setdefault("guards", {})
    t = store.get(device.index)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        store[device.index] = t
    return t


def guard_tripped(device=None, reset=True):
    """True if an f16x3 kernel launched by THIS thread saw an out-of-range value since the last reset (synchronises).
    Raw ops called outside a guarded region report here; the modules' entry points check it themselves."""
    tripped = False
    for idx, t in getattr(_tls, "guards", {}).items():
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        if int(t.item()):
            tripped = True
            if reset:
                t.zero_()
    return tripped


def guard_report(reset=True):
    """Names of the guarded STAGES (estimator encoders, refinement, context encoder, fusion chain, ...) this thread re-ran
    in bf16x6 since the last reset because a value left the fp16 split's range - how much of a forward fell back."""
    trips = list(getattr(_tls, "guard_trips", []))
    if reset:
        _tls.guard_trips = []
    return trips


def with_range_guard(fn, device=None, name=None):
    """Run fn(); in f16x3 mode, if a kernel reported a value outside the fp16 split's range, run it again in bf16x6.
    name: what guard_report() lists when that happens.  The modules guard their STAGES one by one (round 4: a trip in the
    context encoder no longer recomputes the estimator), the outermost active guard decides.

    The state (nesting depth, retry mode, flag) is per host thread, the flag per device, so concurrent forwards from
    several threads / for several GPUs do not interact.  Cost: one asynchronous memset before and ONE device-to-host
    read of the flag after fn() (which is also the only host synchronisation of a forward)."""
    if current_mode() != CONV_F16X3 or getattr(_tls, "depth", 0):
        return fn()            # not the fp16 mode, or already inside a guarded region (the outermost one decides)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    flag = _guard(dev)
    flag.zero_()               # stream-ordered: clears stale reports of unguarded raw-op calls
    _tls.depth = 1
    try:
        out = fn()
        if not int(flag.item()):
            return out
        flag.zero_()
        _tls.__dict__.setdefault("guard_trips", []).append(name or getattr(fn, "__qualname__", "?"))
        with conv_mode(CONV_BF16X6):
            return fn()
    finally:
        _tls.depth = 0


def optimistic(fn, device):
    """Run fn() ONCE with every stage guard inside it transparent and all of its kernels reporting to one flag, read with a
    single host synchronisation at the end - the common case (nothing leaves the fp16 split's range) then costs what one
    guard costs, not one device-to-host read per stage.  Returns (result, tripped); on a trip the caller runs fn() again
    with the stage guards active, so that only the stages that saw the value fall back to bf16x6."""
    if current_mode() != CONV_F16X3 or inside_guard():
        return fn(), False
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    with guard_scope(flag):
        out = fn()
    return out, bool(int(flag.item()))


def inside_guard():
    """True inside a guarded region / guard scope of this thread (nested entry points then neither read a flag nor retry)."""
    return bool(getattr(_tls, "depth", 0))


def note_guard_trip(name):
    _tls.__dict__.setdefault("guard_trips", []).append(name)


@contextlib.contextmanager
def guard_scope(flag):
    """Everything launched by this thread inside the scope reports range violations of the f16x3 mode to `flag` (a
    device int32 the caller zeroed), and the modules' own guarded entry points neither read a flag nor retry: the
    caller reads `flag` when it wants to (parallel.SequencePipeline: asynchronously, one sequence later)."""
    saved = (getattr(_tls, "depth", 0), getattr(_tls, "guard_flag", None))
    _tls.depth, _tls.guard_flag = 1, flag
    try:
        yield flag
    finally:
        _tls.depth, _tls.guard_flag = saved


def _first_tensor(args, kwargs):
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor):
            return a
        if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
            return a[0]
    return None


def range_guarded(method):
    """Decorator for module entry points (forward, iter, estimate_pairs, ...): the call runs inside with_range_guard
    on the device of its first tensor argument.  Nested guarded calls are transparent."""
    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        if current_mode() != CONV_F16X3 or getattr(_tls, "depth", 0):
            return method(self, *args, **kwargs)
        t = _first_tensor(args, kwargs)
        dev = t.device if (t is not None and t.is_cuda) else None
        return with_range_guard(lambda: method(self, *args, **kwargs), dev,
                                name="%s.%s" % (getattr(self, "guard_name", type(self).__name__), method.__name__))
    return wrapper


USE_S16 = os.environ.get("ACCFLOW_S16", "1") == "1"
S16_VIA_MULTI = os.environ.get("ACCFLOW_S16_VIA_MULTI", "0") == "1"   # in0 / in1 S16 convs on the multi-source kernel (A/B)


def s16_active():
    """True when the update block / encoders keep their conv-to-conv activations in the pre-split S16 format: the fp16
    split mode only (a guard retry in bf16x6 runs the fp32-activation path), ACCFLOW_S16=0 switches it off (A/B)."""
    return USE_S16 and current_mode() == CONV_F16X3


class S16:
    """(B, C, H, W) activations PRE-SPLIT for the matrix-core kernels (accflow_conv_desc.in_fmt / out16): storage
    (B, O = ceil(C/8), 2 terms, H, W) of 16-byte chunks = the 8 fp16 halfs {hi | lo}(x[b, 8o + j] * 2^4).  `data` is an
    int32 tensor (B, O, 2, H, W, 4) - possibly an octet slice of a larger buffer (channel slices at multiples of 8 are
    free, like channel slices of the fp32 buffers)."""
    __slots__ = ("data", "C")

    def __init__(self, data, C):
        B, O, T, H, W, Q = data.shape
        if not (data.is_cuda and data.dtype == torch.int32 and T == 2 and Q == 4 and O == (C + 7) // 8):
            raise RuntimeError("S16: expected an int32 CUDA tensor (B, ceil(C/8), 2, H, W, 4)")
        st = data.stride()
        if not (st[5] == 1 and st[4] == 4 and st[3] == 4 * W and st[2] == 4 * W * H and (O == 1 or st[1] == 8 * W * H)):
            raise RuntimeError("S16: the octet block must be dense (strides %s)" % (st,))
        self.data, self.C = data, C

    @classmethod
    def empty(cls, B, C, H, W, device, zero=False):
        O = (C + 7) // 8
        mk = torch.zeros if zero else torch.empty
        return cls(mk((B, O, 2, H, W, 4), dtype=torch.int32, device=device), C)

    @property
    def shape(self):
        return (self.data.shape[0], self.C, self.data.shape[3], self.data.shape[4])

    @property
    def device(self):
        return self.data.device

    @property
    def bs(self):   # batch stride in 4-byte words
        return self.data.stride(0) if self.data.shape[0] > 1 else self.data.shape[1] * self.data.stride(1)

    def ptr(self):
        return self.data.data_ptr()

    def channels(self, c0, c1):
        """Channel slice [c0, c1) as a view; c0 must be a multiple of 8 (c1 may end inside the last octet)."""
        if c0 % 8 or not (0 <= c0 < c1 <= ((self.C + 7) // 8) * 8):
            raise RuntimeError("S16.channels: the slice must start at a multiple of 8 inside the tensor")
        return S16(self.data[:, c0 // 8:(c1 + 7) // 8], c1 - c0)

    def batch(self, b0, b1):
        return S16(self.data[b0:b1], self.C)

    def to_float(self):
        """fp32 (B, C, H, W) = (hi + lo) / 2^4 (tests / debugging; exact when the producer's value had <= 22 bits)"""
        B, O, _, H, W, _ = self.data.shape
        halfs = self.data.contiguous().view(torch.float16).view(B, O, 2, H, W, 8).float()
        v = (halfs[:, :, 0] + halfs[:, :, 1]) / 16.0                      # (B, O, H, W, 8)
        return v.permute(0, 1, 4, 2, 3).reshape(B, O * 8, H, W)[:, :self.C].contiguous()

    @classmethod
    def from_float(cls, x):
        """The split the kernels perform, with torch ops (tests / one-off conversions, not the hot path)."""
        B, C, H, W = x.shape
        O = (C + 7) // 8
        xp = torch.zeros((B, O * 8, H, W), dtype=torch.float32, device=x.device)
        xp[:, :C] = x.float() * 16.0
        hi = xp.half()
        lo = (xp - hi.float()).half()
        t = torch.stack([hi, lo], dim=0).view(2, B, O, 8, H, W).permute(1, 2, 0, 4, 5, 3).contiguous()   # (B, O, 2, H, W, 8)
        return cls(t.view(torch.int32).view(B, O, 2, H, W, 4), C)


def to_p32(x):
    """fp32 (B, C, H, W), C % 8 == 0 -> the same-shaped tensor holding the PIXEL-MAJOR layout (B, C/8, H*W, 8)
    (accflow_conv_desc.p32; tests / module boundaries - the hot path's producers write it themselves)."""
    B, C, H, W = x.shape
    return x.reshape(B, C // 8, 8, H * W).permute(0, 1, 3, 2).contiguous().view(B, C, H, W)


def from_p32(t):
    """Inverse of to_p32."""
    B, C, H, W = t.shape
    return t.reshape(B, C // 8, H * W, 8).permute(0, 1, 3, 2).contiguous().view(B, C, H, W)


def tapgemm_channel_order(cin, device=None):
    """Input-channel order of the second product of ACCFLOW_EPI_TAPGEMM: position p = 32 blk + 16 s + 8 h + e of the
    packed reduction holds channel 32 blk + 8 (2 s + e // 4) + 4 h + e % 4 - lane half h of a 32x32 MFMA accumulator tile
    holds rows 8 i + 4 h + j (i, j = 0..3), and K-step s of the next product takes i = 2 s, 2 s + 1 from it as that
    half's 8 consecutive k (e = 4 (i - 2 s) + j)."""
    pos = torch.arange(cin, device=device)      # (built on the weights' device: packs may be made inside a graph capture)
    blk, s_, h_, e_ = pos // 32, (pos % 32) // 16, (pos % 16) // 8, pos % 8
    return 32 * blk + 8 * (2 * s_ + e_ // 4) + 4 * h_ + e_ % 4


class PackedConv:
    """Device-side packed weights of one nn.Conv2d ([Kpad][CoutPad] + k-table), with optional folded
    per-output-channel scale (BatchNorm eval / ZeroConv2d / constant factor)."""

    __slots__ = ("wpack", "ktab", "bias", "Cout", "Cin", "KH", "KW", "stride", "padH", "padW", "C0",
                 "Kpad", "CoutPad", "tap_major", "wsplit", "wpatch", "wpatch16", "wscale16", "wsplit16", "ztaps", "zcols", "ztaps_acc")

    def __init__(self, weight, bias, stride=1, padding=(0, 0), scale=None, C0=None, tap_major=False, transpose_flip=False):
        """transpose_flip: `weight` is the (Cin, Cout, KH, KW) weight of a FORWARD convolution and the packs are those of its
        input-gradient convolution W[o][c][ky][kx] = weight[c][o][KH-1-ky][KW-1-kx] (backward.conv_dgrad) - no transposed
        copy is made (accflow_conv_pack_all_f32); needs more than 4 logical output channels and PACK_FUSED."""
        lib = _lib.load()
        w = _dense(weight.detach().float().contiguous(), "weight")
        if transpose_flip:
            if not PACK_FUSED or tap_major or w.shape[1] <= 4:
                raise RuntimeError("PackedConv: transpose_flip needs the fused pack entry point, (c, tap) order and > 4 output channels")
            self.Cin, self.Cout, self.KH, self.KW = w.shape
        else:
            self.Cout, self.Cin, self.KH, self.KW = w.shape
        self.stride = int(stride[0] if isinstance(stride, (tuple, list)) else stride)
        if isinstance(padding, (tuple, list)):
            self.padH, self.padW = int(padding[0]), int(padding[1])
        else:
            self.padH = self.padW = int(padding)
        self.C0 = self.Cin if C0 is None else int(C0)
        self.tap_major = bool(tap_major)
        self.Kpad = lib.accflow_conv_kpad(self.Cin, self.KH, self.KW)
        self.CoutPad = lib.accflow_conv_coutpad(self.Cout)
        self.wpack = torch.empty(self.Kpad * self.CoutPad, dtype=torch.float32, device=w.device)
        self.ktab = torch.empty(self.Kpad * 4, dtype=torch.int32, device=w.device)
        sc = _dense(scale.detach().float().contiguous(), "scale") if scale is not None else None
        # which packs this convolution can use: the split-bf16 im2col pack, the direct kernel's patch pack (stride 1, >= 16 input
        # channels), their fp16 hi/lo forms of the row-scaled weights + the per-row inverse scales (f16x3 mode; finite weights
        # always fit; the im2col fp16 pack serves strided convs, 7x7 stems, < 16 input channels)
        has_split = not self.tap_major and self.Cout > 4
        has_patch = has_split and self.stride == 1 and self.Cin >= 16
        has_split16 = has_split and self.Cout > 32
        dev_ = w.device
        self.wsplit = torch.empty(3 * self.Kpad * self.CoutPad, dtype=torch.int16, device=dev_) if has_split else None
        self.wpatch = (torch.empty(lib.accflow_conv_patch_elems(self.Cout, self.Cin, self.KH, self.KW), dtype=torch.int16, device=dev_)
                       if has_patch else None)
        self.wpatch16 = torch.empty_like(self.wpatch) if has_patch else None
        self.wsplit16 = torch.empty_like(self.wsplit) if has_split16 else None
        self.wscale16 = torch.empty(self.CoutPad, dtype=torch.float32, device=dev_) if (has_patch or has_split16) else None
        if PACK_FUSED:     # every pack in ONE launch (bit-identical to the single-purpose entry points below)
            _check(lib.accflow_conv_pack_all_f32(_p(w), _p(sc), self.Cout, self.Cin, self.KH, self.KW, self.C0, int(self.tap_major),
                                                 int(bool(transpose_flip)), _p(self.wpack), _p(self.ktab), _p(self.wsplit),
                                                 _p(self.wpatch), _p(self.wpatch16), _p(self.wscale16), _p(self.wsplit16), _stream()),
                   "accflow_conv_pack_all_f32")
        else:
            _check(lib.accflow_conv_pack_f32(_p(w), _p(sc), self.Cout, self.Cin, self.KH, self.KW, self.C0,
                                             int(self.tap_major), _p(self.wpack), _p(self.ktab), _stream()),
                   "accflow_conv_pack_f32")
            if has_split:
                _check(lib.accflow_conv_pack_bf16s(_p(w), _p(sc), self.Cout, self.Cin, self.KH, self.KW,
                                                   _p(self.wsplit), _stream()), "accflow_conv_pack_bf16s")
            if has_patch:
                _check(lib.accflow_conv_pack_patch(_p(w), _p(sc), self.Cout, self.Cin, self.KH, self.KW,
                                                   _p(self.wpatch), _stream()), "accflow_conv_pack_patch")
                _check(lib.accflow_conv_pack_patch16(_p(w), _p(sc), self.Cout, self.Cin, self.KH, self.KW, _p(self.wpatch16),
                                                     _p(self.wscale16), _stream()), "accflow_conv_pack_patch16")
            if has_split16:
                _check(lib.accflow_conv_pack_split16(_p(w), _p(sc), self.Cout, self.Cin, self.KH, self.KW, _p(self.wsplit16),
                                                     _p(self.wscale16), _stream()), "accflow_conv_pack_split16")
        # deformable (tap-major) pack, stride 1: the same weights as a 1x1 conv over accflow_deform_columns_f32's output
        self.zcols = None
        if self.tap_major and self.stride == 1 and USE_DEFORM_COLUMNS:
            wz = w * sc.view(-1, 1, 1, 1) if sc is not None else w
            wz = wz.permute(0, 2, 3, 1).reshape(self.Cout, self.KH * self.KW * self.Cin, 1, 1).contiguous()
            self.zcols = PackedConv(wz, bias)
        # <= 4 output channels, stride 1, "same": all taps as ONE 1x1 conv on the matrix cores + accflow_tap_sum_f32
        self.ztaps = self.ztaps_acc = None
        if (USE_TAPSUM and not self.tap_major and self.Cout <= 4 and self.stride == 1 and self.KH * self.KW >= 2
                and self.Cin >= 16 and 2 * self.padH == self.KH - 1 and 2 * self.padW == self.KW - 1):
            wz = w * sc.view(-1, 1, 1, 1) if sc is not None else w
            wz = wz.permute(2, 3, 0, 1).reshape(self.KH * self.KW * self.Cout, self.Cin, 1, 1).contiguous()
            self.ztaps = PackedConv(wz, None, C0=self.C0)
            # the same tap matrix with the input channels of every 32-block in MFMA-ACCUMULATOR order (position 16 s + 8 h
            # + e holds channel 8 (2 s + e // 4) + 4 h + e % 4): the second product of ACCFLOW_EPI_TAPGEMM, whose B operand
            # is the producing convolution's accumulator tile (conv2d_tapgemm below)
            if self.Cin % 128 == 0 and self.KH * self.KW * self.Cout <= TAPGEMM_MAXROWS and self.C0 == self.Cin:
                self.ztaps_acc = PackedConv(wz[:, tapgemm_channel_order(self.Cin, w.device)].contiguous(), None)
        self.bias = _dense(bias.detach().float().contiguous(), "bias") if bias is not None else None
        # Every pack kernel above was enqueued on the CURRENT stream: a pack must be built on a stream every later
        # consumer is ordered after (the modules pre-pack on the main stream before forking side streams, see
        # RAFT._refine, and networks/_packs.PackCache orders uses on other streams behind an event recorded here);
        # w / sc may be temporaries, same-stream reuse is ordered by the caching allocator.

    def out_size(self, H, W):
        OH = (H + 2 * self.padH - self.KH) // self.stride + 1
        OW = (W + 2 * self.padW - self.KW) // self.stride + 1
        return OH, OW


class ConvStats:
    """InstanceNorm partial statistics a convolution gathered in its epilogue: (B, Cout, slots, 3) records
    {sum, M2, n}; consumed by instance_norm(..., stats=...)."""
    __slots__ = ("partial", "slots")

    def __init__(self, partial, slots):
        self.partial, self.slots = partial, slots


USE_NORM_STATS = os.environ.get("ACCFLOW_NORM_STATS", "1") == "1"
USE_NORM_ON_LOAD = os.environ.get("ACCFLOW_NORM_ON_LOAD", "1") == "1"


def conv2d(pk, in0, in1=None, out=None, act=ACT_NONE, epi=EPI_STORE, e0=None, e1=None, out2=None,
           offset=None, dmask=None, mode=None, want_stats=False, pre=None, algo_cin=None, in_norm=None, out16=None,
           fp32_out=True, cache=None, p32_out=False):
    """out = epilogue(act(conv(cat[in0, in1]) + bias)); `out` may be a channel slice of a larger
    buffer.  Returns `out`; with want_stats (plain store, no activation) returns (out, ConvStats or None): the
    InstanceNorm statistics of the output gathered by the kernel's epilogue when the chosen kernel supports it.
    S16 tensors (f16x3 mode, direct-kernel shapes): in0 / in1 may be ops.S16; out16 = an ops.S16 that receives the
    pre-split copy of the result (GRU_ZR: of r*h); fp32_out=False with out16 skips the fp32 destination (GRU_ZR: out2),
    the call then returns out16.
    p32_out (S16 sources, plain store onto a multiple of 128 channels): the fp32 result in the PIXEL-MAJOR layout
    (accflow_conv_desc.p32; ops.from_p32 gives NCHW back) - what the GRU epilogues read their context addend in.  With an
    ops.S16 `e0` (GRU_ZR / GRU_Q, 1x5 / 5x1): the packed-operand GRU epilogue - `pre`, `e1` (z) and GRU_ZR's `out` (z) are
    pixel-major tensors, the state is read from / written to S16 tensors only.
    cache = (dict, key): a call site that repeats with the SAME tensors (the 12 refinement iterations run the same
    convolutions on the same workspace buffers) keeps its filled descriptor there and re-launches it without rebuilding
    it - most of the host time of a launch; only S16 calls use it, and not while the per-launch profiler is active."""
    if cache is not None and profiler.ACTIVE is None:
        hit = cache[0].get(cache[1])
        if hit is not None:
            _check(_lib.load().accflow_conv2d_f32(ctypes.byref(hit[0]), _stream()), "accflow_conv2d_f32 (cached)")
            return hit[1]
    if pk is None:
        # a replay call site (pk / tensors omitted) without a stored descriptor: the per-launch profiler was switched on
        # between two iterations of one refinement, or the cache was cleared - the caller must pass the real arguments
        raise RuntimeError("conv2d: no packed weights given and no cached descriptor for %r to replay"
                           % (cache[1] if cache is not None else None,))
    if isinstance(in0, S16) and pk.Cout <= 4 and out16 is None and (
            pk.ztaps is None or ((current_mode() if mode is None else mode) == CONV_F16X3 and not TAPSUM_F16)):
        # a <= 4-channel regression fed a pre-split tensor while the tap-sum path is switched off (ACCFLOW_CONV_TAPSUM=0) or
        # held on bf16x6 (ACCFLOW_TAPSUM_F16=0; both A/B switches): the kernels that then run read fp32, and (hi + lo) / 2^4
        # is the tensor's value to 22 bits
        in0 = in0.to_float()
        in1 = in1.to_float() if isinstance(in1, S16) else in1
    if (out16 is not None or isinstance(in0, S16)) and not (pk.ztaps is not None and out16 is None):
        if want_stats or offset is not None or in_norm is not None:
            raise RuntimeError("conv2d: S16 tensors are for plain direct-kernel convolutions")
        return _conv2d_s16(pk, in0, in1, out, act, epi, e0, e1, out2, mode, pre, algo_cin, out16, fp32_out, cache, p32_out)
    if p32_out or isinstance(e0, S16):
        raise RuntimeError("conv2d: pixel-major / pre-split operands belong to the S16 direct-kernel form")
    if want_stats:
        if act != ACT_NONE or epi != EPI_STORE or offset is not None:
            raise RuntimeError("conv2d: statistics are gathered for plain convolutions only (store, no activation)")
        holder = []
        o = _conv2d(pk, in0, in1, out, act, epi, e0, e1, out2, offset, dmask, mode, holder, None, None, in_norm)
        if o is None:
            return None   # (in_norm given but the kernel chosen for this call cannot normalise on load)
        return o, (holder[0] if holder else None)
    return _conv2d(pk, in0, in1, out, act, epi, e0, e1, out2, offset, dmask, mode, None, pre, algo_cin, in_norm)


def _conv2d(pk, in0, in1, out, act, epi, e0, e1, out2, offset, dmask, mode, stats_holder, pre=None, algo_cin=None,
            in_norm=None):
    """algo_cin: input channels of the convolution this launch stands for in the reference's formulation (profiler
    accounting only): the GRU gate convs run over 2/3 of their input channels per iteration, the context third being
    convolved once per pair (algo_cin = 0 there) - the algorithmic work is the reference's full conv per iteration."""
    lib = _lib.load()
    md = current_mode() if mode is None else mode
    if (pk.ztaps is not None and md != CONV_F32 and offset is None and epi in (EPI_STORE, EPI_ACCUM, EPI_RES_RELU)
            and (isinstance(in0, S16) or in0.shape[0] * in0.shape[2] * in0.shape[3] >= TAPSUM_MIN_PIXELS)):
        z = conv2d(pk.ztaps, in0, in1, mode=CONV_BF16X6 if (md == CONV_F16X3 and not TAPSUM_F16) else md)
        B, _, H, W = in0.shape
        if out is None:
            out = torch.empty((B, pk.Cout, H, W), dtype=torch.float32, device=in0.device)
        out_bs = _plane4(out, "out")
        e0_bs = _plane4(e0, "e0") if e0 is not None else 0
        tm = profiler.ACTIVE
        t0 = tm.begin() if tm is not None and tm.wants("conv2d") else None
        _check(lib.accflow_tap_sum_f32(_p(z), _p(pk.bias) if pk.bias is not None else None, _p(e0) if e0 is not None else None,
                                       e0_bs, _p(out), out_bs, B, pk.Cout, H, W, pk.KH, pk.KW, pk.padH, pk.padW, int(act),
                                       int(epi), _stream()), "accflow_tap_sum_f32")
        if t0 is not None:
            tm.end("conv2d", t0, 0.0, "tap_sum Cout%d k%dx%d B%d %dx%d" % (pk.Cout, pk.KH, pk.KW, B, H, W))
        return out
    if offset is not None and pk.zcols is not None and md != CONV_F32 and in1 is None:
        # deformable conv in two passes: deformed im2col columns (memory-bound), then a 1x1 conv on the matrix cores
        B, C, H, W = in0.shape
        cols = torch.empty((B, pk.KH * pk.KW * C, H, W), dtype=torch.float32, device=in0.device)
        tm = profiler.ACTIVE
        t0 = tm.begin() if tm is not None and tm.wants("conv2d") else None
        _check(lib.accflow_deform_columns_f32(_p(in0), _plane4(in0, "in0"), _p(offset), _plane4(offset, "offset"), _p(dmask),
                                              _plane4(dmask, "dmask"), _p(cols), B, C, H, W, pk.KH, pk.KW, pk.padH, pk.padW,
                                              _stream()), "accflow_deform_columns_f32")
        if t0 is not None:
            tm.end("conv2d", t0, 0.0, "deform_columns C%d k%dx%d B%d %dx%d" % (C, pk.KH, pk.KW, B, H, W))
        return conv2d(pk.zcols, cols, out=out, act=act, epi=epi, e0=e0, e1=e1, out2=out2, mode=md)
    d = ConvDesc()
    d.in0_bs = _plane4(in0, "in0")
    B, C0, H, W = in0.shape
    C1 = 0
    if in1 is not None:
        d.in1_bs = _plane4(in1, "in1")
        if in1.shape[0] != B or in1.shape[2:] != in0.shape[2:]:
            raise RuntimeError("conv2d: in0/in1 shape mismatch")
        C1 = in1.shape[1]
    if C0 != pk.C0 or C0 + C1 != pk.Cin:
        raise RuntimeError("conv2d: channel split (%d,%d) does not match packed weights (C0=%d, Cin=%d)"
                           % (C0, C1, pk.C0, pk.Cin))
    OH, OW = pk.out_size(H, W)
    if out is None:
        out = torch.empty((B, pk.Cout, OH, OW), dtype=torch.float32, device=in0.device)
    d.out_bs = _plane4(out, "out")
    n_out = pk.Cout // 2 if epi == EPI_GRU_ZR else pk.Cout
    if tuple(out.shape) != (B, n_out, OH, OW):
        raise RuntimeError("conv2d: out shape %s != %s" % (tuple(out.shape), (B, n_out, OH, OW)))
    d.in0, d.in1 = in0.data_ptr(), (in1.data_ptr() if in1 is not None else None)
    d.C0, d.C1, d.B, d.H, d.W, d.OH, d.OW = C0, C1, B, H, W, OH, OW
    d.KH, d.KW, d.stride, d.padH, d.padW, d.Cout = pk.KH, pk.KW, pk.stride, pk.padH, pk.padW, pk.Cout
    d.wpack, d.ktab, d.Kpad, d.CoutPad = pk.wpack.data_ptr(), pk.ktab.data_ptr(), pk.Kpad, pk.CoutPad
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.out, d.act, d.epi = out.data_ptr(), act, epi
    d.mode = md
    d.wsplit = pk.wsplit.data_ptr() if pk.wsplit is not None else None
    d.wpatch = pk.wpatch.data_ptr() if (pk.wpatch is not None and USE_PATCH) else None
    if d.mode == CONV_F16X3 and pk.wscale16 is not None:
        if d.wpatch and pk.wpatch16 is not None:
            d.wpatch16 = pk.wpatch16.data_ptr()
        if pk.wsplit16 is not None:
            d.wsplit16 = pk.wsplit16.data_ptr()
        d.wscale16 = pk.wscale16.data_ptr()
        d.guard = _guard(in0.device).data_ptr()
    if d.wpatch and _ksplit_on() and B * OH * OW <= KSPLIT_MAX_PIXELS and pk.Cout > 4:
        # small grids (the batch-1 fusion chain): scratch for 4 K-parts, summed by a second kernel
        ws = _ksplit_ws(4 * B * pk.Cout * OH * OW, in0.device)
        d.kws, d.kws_elems = ws.data_ptr(), ws.numel()
    if e0 is not None:
        d.e0_bs = _plane4(e0, "e0")
        d.e0 = e0.data_ptr()
    if e1 is not None:
        d.e1_bs = _plane4(e1, "e1")
        d.e1 = e1.data_ptr()
    if out2 is not None:
        d.out2_bs = _plane4(out2, "out2")
        d.out2 = out2.data_ptr()
    if pre is not None:  # pre-activation addend of the GRU epilogues, (B, Cout, OH, OW)
        if epi not in (EPI_GRU_ZR, EPI_GRU_Q) or tuple(pre.shape) != (B, pk.Cout, OH, OW):
            raise RuntimeError("conv2d: `pre` is a (B, Cout, OH, OW) addend of the GRU epilogues")
        d.pre_bs = _plane4(pre, "pre")
        d.pre = pre.data_ptr()
    if offset is not None:
        d.offset_bs = _plane4(offset, "offset")
        d.offset = offset.data_ptr()
        d.dmask_bs = _plane4(dmask, "dmask")
        d.dmask = dmask.data_ptr()
        if not pk.tap_major:
            raise RuntimeError("deformable conv needs tap-major packed weights")
    if in_norm is not None:
        # in0 = the raw output of a convolution; in_norm = its (B, C0, 2) {mean, rstd} (instance_stats_finalize): the
        # kernel reads relu((x - mean) * rstd).  None is returned if the kernel chosen for this call cannot do that.
        if in1 is not None or tuple(_dense(in_norm, "in_norm").shape) != (B, C0, 2):
            raise RuntimeError("conv2d: in_norm is (B, C0, 2) for a single-source input")
        if not lib.accflow_conv_in_norm_supported(ctypes.byref(d)):
            return None
        d.in_norm = in_norm.data_ptr()
    if stats_holder is not None and USE_NORM_STATS:
        slots = lib.accflow_conv_stat_slots(ctypes.byref(d))
        if slots > 0:
            st = ConvStats(torch.empty((B, pk.Cout, slots, 3), dtype=torch.float32, device=in0.device), slots)
            d.stats, d.stat_slots = st.partial.data_ptr(), slots
            stats_holder.append(st)
    tm = profiler.ACTIVE
    if tm is not None and tm.wants("conv2d"):
        t0 = tm.begin()
        _check(lib.accflow_conv2d_f32(ctypes.byref(d), _stream()), "accflow_conv2d_f32")
        acin = pk.Cin if algo_cin is None else algo_cin
        tm.end("conv2d", t0, 2.0 * acin * pk.KH * pk.KW * pk.Cout * B * OH * OW,  # algorithmic flop
               "Cin%d Cout%d k%dx%d s%d B%d %dx%d%s%s" % (pk.Cin, pk.Cout, pk.KH, pk.KW, pk.stride, B, OH, OW,
                                                         " deform" if offset is not None else "",
                                                         "" if algo_cin is None else " (stands for Cin%d)" % algo_cin),
               work_exec=2.0 * pk.Cin * pk.KH * pk.KW * pk.Cout * B * OH * OW)
        return out
    _check(lib.accflow_conv2d_f32(ctypes.byref(d), _stream()), "accflow_conv2d_f32")
    return out


def _conv2d_s16(pk, in0, in1, out, act, epi, e0, e1, out2, mode, pre, algo_cin, out16, fp32_out, cache=None, p32_out=False):
    lib = _lib.load()
    md = current_mode() if mode is None else mode
    if md != CONV_F16X3 or (pk.wpatch16 is None and pk.wsplit16 is None):
        raise RuntimeError("conv2d: S16 tensors need the f16x3 mode and a direct-kernel weight pack")
    d = ConvDesc()
    fmt = 0
    srcs = []
    for k, t in enumerate((in0, in1)):
        if t is None:
            srcs.append((None, 0, 0))
        elif isinstance(t, S16):
            fmt |= 1 << k
            srcs.append((t.ptr(), t.bs, t.C))
        else:
            srcs.append((t.data_ptr(), _plane4(t, "in%d" % k), t.shape[1]))
    if fmt and fmt != (3 if in1 is not None else 1):
        raise RuntimeError("conv2d: both sources must have the same format")
    B, _, H, W = in0.shape
    if in1 is not None and (in1.shape[0] != B or tuple(in1.shape[2:]) != (H, W)):
        raise RuntimeError("conv2d: in0/in1 shape mismatch")
    C0, C1 = srcs[0][2], srcs[1][2]
    if C0 != pk.C0 or C0 + C1 != pk.Cin:
        raise RuntimeError("conv2d: channel split (%d,%d) does not match packed weights (C0=%d, Cin=%d)" % (C0, C1, pk.C0, pk.Cin))
    OH, OW = pk.out_size(H, W)
    n_out = pk.Cout // 2 if epi == EPI_GRU_ZR else pk.Cout
    dev = in0.data.device if isinstance(in0, S16) else in0.device
    if out is None and (fp32_out or epi == EPI_GRU_ZR):
        out = torch.empty((B, n_out, OH, OW), dtype=torch.float32, device=dev)
    if out is not None:
        d.out_bs = _plane4(out, "out")
        if tuple(out.shape) != (B, n_out, OH, OW):
            raise RuntimeError("conv2d: out shape %s != %s" % (tuple(out.shape), (B, n_out, OH, OW)))
        d.out = out.data_ptr()
    if out16 is not None:
        if tuple(out16.shape) != (B, n_out, OH, OW):
            raise RuntimeError("conv2d: out16 shape %s != %s" % (out16.shape, (B, n_out, OH, OW)))
        d.out16, d.out16_bs = out16.ptr(), out16.bs
    d.in0, d.in0_bs = srcs[0][0], srcs[0][1]
    d.in1, d.in1_bs = srcs[1][0], srcs[1][1]
    d.in_fmt = fmt
    if S16_VIA_MULTI and fmt:
        # the same convolution through the multi-source kernel (same pack, same results; per-call switch for A/B runs)
        d.nsrc = 2 if in1 is not None else 1
        for k in range(d.nsrc):
            S = d.src[k]
            S.ptr, S.bs, S.C, S.Hs, S.Ws = srcs[k][0], srcs[k][1], srcs[k][2], H, W
            S.step, S.oy, S.ox, S.KH, S.KW, S.padH, S.padW = 1, 0, 0, pk.KH, pk.KW, pk.padH, pk.padW
    d.C0, d.C1, d.B, d.H, d.W, d.OH, d.OW = C0, C1, B, H, W, OH, OW
    d.KH, d.KW, d.stride, d.padH, d.padW, d.Cout = pk.KH, pk.KW, pk.stride, pk.padH, pk.padW, pk.Cout
    d.wpack, d.ktab, d.Kpad, d.CoutPad = pk.wpack.data_ptr(), pk.ktab.data_ptr(), pk.Kpad, pk.CoutPad
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.act, d.epi, d.mode = act, epi, md
    d.wsplit = pk.wsplit.data_ptr() if pk.wsplit is not None else None
    if pk.wpatch16 is not None:
        d.wpatch, d.wpatch16 = pk.wpatch.data_ptr(), pk.wpatch16.data_ptr()
    if pk.wsplit16 is not None:     # (the stem kernel - 7x7 stride 2 of the image, S16 output - reads the im2col fp16 pack)
        d.wsplit16 = pk.wsplit16.data_ptr()
    d.wscale16 = pk.wscale16.data_ptr()
    d.guard = _guard(dev).data_ptr()
    ws_keep = None
    if _ksplit_on() and B * OH * OW <= KSPLIT_MAX_PIXELS and pk.Cout > 4 and not isinstance(e0, S16) and not p32_out:
        ws_keep = _ksplit_ws(4 * B * pk.Cout * OH * OW, dev)    # small grids: see _conv2d
        d.kws, d.kws_elems = ws_keep.data_ptr(), ws_keep.numel()
    if isinstance(e0, S16):
        # the GRU state kept pre-split only (accflow_conv_desc.e0_fmt, GRU_ZR / GRU_Q of the 5-tap convolutions): the epilogue
        # reads h = (hi + lo) / 2^4 with 8-byte block loads
        if epi not in (EPI_GRU_ZR, EPI_GRU_Q) or tuple(e0.shape) != (B, n_out, OH, OW):
            raise RuntimeError("conv2d: an S16 e0 is the GRU epilogues' state operand, shape (B, hidden, OH, OW)")
        d.e0_bs, d.e0, d.e0_fmt = e0.bs, e0.ptr(), 1
        d.p32 = 3 if epi == EPI_GRU_ZR else 6      # z out + pre / pre + e1 (z) in the pixel-major layout
    elif e0 is not None:
        d.e0_bs, d.e0 = _plane4(e0, "e0"), e0.data_ptr()
    if e1 is not None:
        d.e1_bs, d.e1 = _plane4(e1, "e1"), e1.data_ptr()
    if out2 is not None:
        d.out2_bs, d.out2 = _plane4(out2, "out2"), out2.data_ptr()
    if pre is not None:
        if epi not in (EPI_GRU_ZR, EPI_GRU_Q) or tuple(pre.shape) != (B, pk.Cout, OH, OW):
            raise RuntimeError("conv2d: `pre` is a (B, Cout, OH, OW) addend of the GRU epilogues")
        d.pre_bs, d.pre = _plane4(pre, "pre"), pre.data_ptr()
    if p32_out:
        if epi != EPI_STORE or out is None or pk.Cout % 128 or isinstance(e0, S16):
            raise RuntimeError("conv2d: p32_out is a plain fp32 store onto a multiple of 128 channels")
        d.p32 = 1
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("conv2d") else None
    _check(lib.accflow_conv2d_f32(ctypes.byref(d), _stream()), "accflow_conv2d_f32 (S16)")
    if t0 is not None:
        acin = pk.Cin if algo_cin is None else algo_cin
        tm.end("conv2d", t0, 2.0 * acin * pk.KH * pk.KW * pk.Cout * B * OH * OW,
               "Cin%d Cout%d k%dx%d s%d B%d %dx%d S16%s%s" % (pk.Cin, pk.Cout, pk.KH, pk.KW, pk.stride, B, OH, OW,
                                                            "in" if fmt else "", "" if algo_cin is None else " (stands for Cin%d)" % algo_cin),
               work_exec=2.0 * pk.Cin * pk.KH * pk.KW * pk.Cout * B * OH * OW)
    ret = out if (out is not None and (fp32_out or out16 is None)) else out16
    if cache is not None and profiler.ACTIVE is None:   # (the predicate conv2d() replays under)
        # the entry keeps every tensor the descriptor points at alive: the pack, the S16 / fp32 operands, and the
        # split-K scratch d.kws points into (_ksplit_ws drops its buffer when a later conv asks for a larger one)
        cache[0][cache[1]] = (d, ret, (pk, in0, in1, out, out16, e0, e1, out2, pre, ws_keep))
    return ret



def tapgemm_eligible(pk1, pk2, in0, in1=None):
    """conv2d_tapgemm's shapes: f16x3 mode, S16 sources, conv1 = a direct-kernel convolution onto a multiple of 128 channels,
    conv2 = a small-Cout "same" convolution with <= 18 tap rows, and a grid of >= 320 workgroups."""
    if not (FUSE_TAPGEMM and current_mode() == CONV_F16X3 and isinstance(in0, S16) and (in1 is None or isinstance(in1, S16))):
        return False
    B, _, H, W = in0.shape
    # grids of < 320 workgroups (one or two items of 60x128) go to the split-K form of the direct kernel (conv2d_direct.hip:
    # launch_conv_direct), whose partial sums a reduce kernel finishes - the three-launch flow head stays there
    nb = B * ((W + 31) // 32) * ((H + 3) // 4) * (pk1.Cout // 128)
    return (pk1.wpatch16 is not None and pk1.Cout % 128 == 0 and pk1.stride == 1 and pk2.ztaps_acc is not None
            and pk2.Cin == pk1.Cout and pk1.out_size(H, W) == (H, W) and not (_ksplit_on() and nb < 320))


def conv2d_tapgemm(pk1, pk2, in0, in1=None, out=None, epi=EPI_STORE, e0=None, cache=None):
    """out = epi(conv2(relu(conv1(cat[in0, in1])))) - FlowHead (update.py:12-13) - in TWO launches: conv1 on the direct kernel
    with the ACCFLOW_EPI_TAPGEMM epilogue (its output never reaches HBM: the workgroup multiplies it by conv2's tap matrix and
    writes 18 rows per pixel and 128-channel block), then accflow_tap_sum_parts_f32 (adds the blocks, shifts and sums the
    taps, bias, epilogue).  Same arithmetic as the three-launch form up to the order of the fp32 sums over conv1's channels.
    cache = (dict, key): replay protocol of conv2d (pass pk1 = None to replay)."""
    lib = _lib.load()
    if cache is not None and profiler.ACTIVE is None:
        hit = cache[0].get(cache[1])
        if hit is not None:
            _check(lib.accflow_conv2d_f32(ctypes.byref(hit[0]), _stream()), "accflow_conv2d_f32 (tapgemm, cached)")
            _check(lib.accflow_tap_sum_parts_f32(*hit[2], _stream()), "accflow_tap_sum_parts_f32 (cached)")
            return hit[1]
    if pk1 is None:
        raise RuntimeError("conv2d_tapgemm: no packed weights given and no cached descriptor for %r to replay"
                           % (cache[1] if cache is not None else None,))
    if not tapgemm_eligible(pk1, pk2, in0, in1):
        raise RuntimeError("conv2d_tapgemm: shapes / mode outside the fused form (ops.tapgemm_eligible)")
    if epi not in (EPI_STORE, EPI_ACCUM, EPI_RES_RELU) or (epi != EPI_STORE and e0 is None):
        raise RuntimeError("conv2d_tapgemm: epi is STORE, ACCUM (+ e0) or RES_RELU (+ e0)")
    B, _, H, W = in0.shape
    C0, C1 = in0.C, (in1.C if in1 is not None else 0)
    if C0 != pk1.C0 or C0 + C1 != pk1.Cin:
        raise RuntimeError("conv2d_tapgemm: channel split (%d,%d) does not match packed weights" % (C0, C1))
    dev = in0.data.device
    pz = pk2.ztaps_acc
    rows, nparts = pz.Cout, pk1.Cout // 128
    z = torch.empty((nparts, B, rows, H, W), dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty((B, pk2.Cout, H, W), dtype=torch.float32, device=dev)
    d = ConvDesc()
    d.in0, d.in0_bs = in0.ptr(), in0.bs
    if in1 is not None:
        d.in1, d.in1_bs = in1.ptr(), in1.bs
    d.in_fmt = 3 if in1 is not None else 1
    d.C0, d.C1, d.B, d.H, d.W, d.OH, d.OW = C0, C1, B, H, W, H, W
    d.KH, d.KW, d.stride, d.padH, d.padW, d.Cout = pk1.KH, pk1.KW, 1, pk1.padH, pk1.padW, pk1.Cout
    d.wpack, d.ktab, d.Kpad, d.CoutPad = pk1.wpack.data_ptr(), pk1.ktab.data_ptr(), pk1.Kpad, pk1.CoutPad
    d.bias = pk1.bias.data_ptr() if pk1.bias is not None else None
    d.act, d.epi, d.mode = ACT_RELU, EPI_TAPGEMM, CONV_F16X3
    d.wpatch, d.wpatch16, d.wscale16 = pk1.wpatch.data_ptr(), pk1.wpatch16.data_ptr(), pk1.wscale16.data_ptr()
    d.guard = _guard(dev).data_ptr()
    d.tg_w16, d.tg_scale, d.tg_out = pz.wpatch16.data_ptr(), pz.wscale16.data_ptr(), z.data_ptr()
    d.tg_out_bs, d.tg_out_ps, d.tg_rows, d.tg_coutpad = rows * H * W, B * rows * H * W, rows, pz.CoutPad
    out_bs = _plane4(out, "out")
    e0_bs = _plane4(e0, "e0") if e0 is not None else 0
    targs = (z.data_ptr(), nparts, B * rows * H * W, rows * H * W, _p(pk2.bias) if pk2.bias is not None else None,
             _p(e0) if e0 is not None else None, e0_bs, _p(out), out_bs, B, pk2.Cout, H, W, pk2.KH, pk2.KW, pk2.padH, pk2.padW,
             int(ACT_NONE), int(epi))
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("conv2d") else None
    _check(lib.accflow_conv2d_f32(ctypes.byref(d), _stream()), "accflow_conv2d_f32 (tapgemm)")
    if t0 is not None:
        px = B * H * W
        tm.end("conv2d", t0, 2.0 * (pk1.Cin * pk1.KH * pk1.KW * pk1.Cout + pk1.Cout * rows) * px,
               "Cin%d Cout%d k%dx%d s1 B%d %dx%d S16in +tapgemm%d" % (pk1.Cin, pk1.Cout, pk1.KH, pk1.KW, B, H, W, rows))
        t0 = tm.begin()
    _check(lib.accflow_tap_sum_parts_f32(*targs, _stream()), "accflow_tap_sum_parts_f32")
    if t0 is not None:
        tm.end("conv2d", t0, 0.0, "tap_sum Cout%d k%dx%d B%d %dx%d (%d parts)" % (pk2.Cout, pk2.KH, pk2.KW, B, H, W, nparts))
    if cache is not None and profiler.ACTIVE is None:
        cache[0][cache[1]] = (d, out, targs, (pk1, pk2, in0, in1, out, e0, z))
    return out


class PackedMulti:
    """Weights of a MULTI-SOURCE convolution (accflow_conv_desc.nsrc, f16x3 mode): source s of the reduction is an ops.S16
    tensor of C[s] channels multiplied by weights[s] = (Cout, C[s], KH[s], KW[s]).  `geo[s]` = (step, oy, ox, KH, KW, padH,
    padW) is how the kernel reads that source (accflow_conv_src).  Built by from_cat (torch.cat([...], 1) feeding a conv:
    one source per member, no copy) or from_strided (a stride-2 convolution as stride-1 work over the input's four
    pixel-parity classes)."""
    __slots__ = ("wpatch16", "wscale16", "bias", "Cout", "CoutPad", "C", "geo", "out_hw", "flop_per_px", "split_c0")

    def __init__(self, weights, bias, geo, scale=None, split_c0=0):
        lib = _lib.load()
        n = len(weights)
        if not 1 <= n <= _lib.MAX_SRC or len(geo) != n:
            raise RuntimeError("PackedMulti: 1..%d sources" % _lib.MAX_SRC)
        ws = [_dense(w.detach().float().contiguous(), "weight") for w in weights]
        self.Cout = ws[0].shape[0]
        self.C = [w.shape[1] for w in ws]
        self.geo = [tuple(int(v) for v in g) for g in geo]
        for w, g in zip(ws, self.geo):
            if w.shape[0] != self.Cout or tuple(w.shape[2:]) != (g[3], g[4]):
                raise RuntimeError("PackedMulti: weight %s does not match its source geometry %s" % (tuple(w.shape), g))
        self.CoutPad = lib.accflow_conv_coutpad(self.Cout)
        arr = lambda v: (ctypes.c_int * n)(*v)  # noqa: E731
        C, KH, KW = arr(self.C), arr([g[3] for g in self.geo]), arr([g[4] for g in self.geo])
        dev = ws[0].device
        self.wpatch16 = torch.empty(lib.accflow_conv_multi_pack_elems(self.Cout, n, C, KH, KW), dtype=torch.int16, device=dev)
        self.wscale16 = torch.empty(self.CoutPad, dtype=torch.float32, device=dev)
        sc = _dense(scale.detach().float().contiguous(), "scale") if scale is not None else None
        wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
        _check(lib.accflow_conv_pack_multi16(wp, _p(sc), self.Cout, n, C, KH, KW, _p(self.wpatch16), _p(self.wscale16), _stream()),
               "accflow_conv_pack_multi16")
        self.bias = _dense(bias.detach().float().contiguous(), "bias") if bias is not None else None
        self.flop_per_px = 2.0 * self.Cout * sum(c * g[3] * g[4] for c, g in zip(self.C, self.geo))
        # accflow_conv_desc.split_c0: output rows >= split_c0 are a second convolution over source 0 alone
        self.split_c0 = int(split_c0)
        if self.split_c0:
            self.flop_per_px = 2.0 * (self.split_c0 * sum(c * g[3] * g[4] for c, g in zip(self.C, self.geo))
                                      + (self.Cout - self.split_c0) * self.C[0] * self.geo[0][3] * self.geo[0][4])

    @classmethod
    def from_cat(cls, weight, bias, splits, padding, scale=None):
        """conv(torch.cat(members, 1)) with `splits` = the members' channel counts, stride 1."""
        if isinstance(padding, (tuple, list)):
            pH, pW = int(padding[0]), int(padding[1])
        else:
            pH = pW = int(padding)
        Cout, Cin, KH, KW = weight.shape
        if sum(splits) != Cin:
            raise RuntimeError("PackedMulti.from_cat: splits %s do not add up to %d input channels" % (splits, Cin))
        ws, c0 = [], 0
        for c in splits:
            ws.append(weight[:, c0:c0 + c])
            c0 += c
        return cls(ws, bias, [(1, 0, 0, KH, KW, pH, pW)] * len(splits), scale)

    @classmethod
    def from_strided_with_projection(cls, weight, bias, padding, wproj, bproj, scale=None, scale_proj=None):
        """A residual block's stride-2 KxK convolution AND its 1x1 stride-2 projection of the same input (extractor.py:9,52:
        conv1 and downsample[0]) as ONE pack: rows [0, Cout) = the KxK convolution over the parity-class sources of
        from_strided, rows [Cout, Cout + Cproj) = the projection, which reads class (0, 0) - source 0, whose single tap (the
        KxK kernel's centre for pad = K // 2) sits on exactly the pixels (2Y, 2X) a 1x1 stride-2 convolution reads
        (accflow_conv_desc.split_c0 = Cout)."""
        p = int(padding[0] if isinstance(padding, (tuple, list)) else padding)
        Cout, Cin, KH, KW = weight.shape
        if tuple(wproj.shape[1:]) != (Cin, 1, 1) or KH != KW or p != KH // 2 or not (KH & 1):
            raise RuntimeError("from_strided_with_projection: an odd 'same' kernel and a 1x1 projection of the same input")
        ws, geo = cls._strided_sources(weight.detach().float(), p)
        if geo[0] != (2, 0, 0, 1, 1, 0, 0):
            raise RuntimeError("from_strided_with_projection: source 0 is not the (0, 0) parity class")
        Cp = wproj.shape[0]
        full = []
        for k, w in enumerate(ws):
            tail = wproj.detach().float() if k == 0 else torch.zeros((Cp, Cin) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
            full.append(torch.cat([w, tail], dim=0).contiguous())
        b = None
        if bias is not None or bproj is not None:
            z = lambda n: torch.zeros(n, dtype=torch.float32, device=weight.device)  # noqa: E731
            b = torch.cat([bias.float() if bias is not None else z(Cout), bproj.float() if bproj is not None else z(Cp)])
        sc = None
        if scale is not None or scale_proj is not None:
            o = lambda n: torch.ones(n, dtype=torch.float32, device=weight.device)  # noqa: E731
            sc = torch.cat([scale.float() if scale is not None else o(Cout), scale_proj.float() if scale_proj is not None else o(Cp)])
        return cls(full, b, geo, sc, split_c0=Cout)

    @staticmethod
    def _strided_sources(weight, p):
        Cout, Cin, KH, KW = weight.shape

        def axis(K):
            out = []
            for par in (0, 1):
                ks = [k for k in range(K) if (k - p) % 2 == par]
                if ks:
                    offs = [(k - p - par) // 2 for k in ks]
                    assert offs == list(range(offs[0], offs[0] + len(ks)))
                    out.append((par, ks, -offs[0]))
            return out

        ws, geo = [], []
        for py, kys, padH in axis(KH):
            for px, kxs, padW in axis(KW):
                ws.append(weight[:, :, kys[0]:kys[-1] + 1:2, kxs[0]:kxs[-1] + 1:2])
                geo.append((2, py, px, len(kys), len(kxs), padH, padW))
        return ws, geo

    @classmethod
    def from_strided(cls, weight, bias, padding, scale=None):
        """A stride-2 convolution (extractor.py:9,52) over ONE tensor as up to four parity-class sources: class (py, px)
        holds the taps (ky, kx) with (ky - pad) % 2 == py, (kx - pad) % 2 == px, read with pixel step 2 from origin
        (py, px); tap ky sits at patch row offset (ky - pad - py) / 2.
        (slices, not index lists: an index tensor is a host-to-device copy, which a stream capture refuses)"""
        p = int(padding[0] if isinstance(padding, (tuple, list)) else padding)
        ws, geo = cls._strided_sources(weight, p)
        return cls(ws, bias, geo, scale)


def conv2d_multi(pk, srcs, out=None, act=ACT_NONE, epi=EPI_STORE, e0=None, e1=None, out2=None, pre=None, out16=None,
                 fp32_out=True, want_stats=False, out_hw=None, lay=None):
    """Multi-source S16 convolution: srcs[s] = the ops.S16 tensor read as source s of `pk` (a PackedMulti; the same tensor
    may appear several times - the parity classes of a strided conv).  out_hw: output size (default: the first source's
    size divided by its step, rounded up).  Other arguments as conv2d; want_stats returns (out, ConvStats or None)."""
    lib = _lib.load()
    if current_mode() != CONV_F16X3:
        raise RuntimeError("conv2d_multi: S16 tensors need the f16x3 mode")
    if len(srcs) != len(pk.C):
        raise RuntimeError("conv2d_multi: %d sources for a %d-source pack" % (len(srcs), len(pk.C)))
    d = ConvDesc()
    B = srcs[0].shape[0]
    if out_hw is None:
        st = pk.geo[0][0]
        out_hw = ((srcs[0].shape[2] + st - 1) // st, (srcs[0].shape[3] + st - 1) // st)
    OH, OW = int(out_hw[0]), int(out_hw[1])
    d.nsrc = len(srcs)
    for k, (t, c, g) in enumerate(zip(srcs, pk.C, pk.geo)):
        if not isinstance(t, S16) or t.C != c or t.shape[0] != B:
            raise RuntimeError("conv2d_multi: source %d must be an ops.S16 of %d channels and batch %d" % (k, c, B))
        S = d.src[k]
        S.ptr, S.bs, S.C, S.Hs, S.Ws = t.ptr(), t.bs, c, t.shape[2], t.shape[3]
        S.step, S.oy, S.ox, S.KH, S.KW, S.padH, S.padW = g
    if lay is not None:
        d.src[0].reserved = int(lay) + 1      # tests / tuning: force a wave layout
    dev = srcs[0].device
    n_out = pk.Cout // 2 if epi == EPI_GRU_ZR else pk.Cout
    if out is None and (fp32_out or epi == EPI_GRU_ZR or out16 is None):
        out = torch.empty((B, n_out, OH, OW), dtype=torch.float32, device=dev)
    if out is not None:
        d.out_bs = _plane4(out, "out")
        if tuple(out.shape) != (B, n_out, OH, OW):
            raise RuntimeError("conv2d_multi: out shape %s != %s" % (tuple(out.shape), (B, n_out, OH, OW)))
        d.out = out.data_ptr()
    if out16 is not None:
        if tuple(out16.shape) != (B, n_out, OH, OW):
            raise RuntimeError("conv2d_multi: out16 shape %s != %s" % (out16.shape, (B, n_out, OH, OW)))
        d.out16, d.out16_bs = out16.ptr(), out16.bs
    d.B, d.H, d.W, d.OH, d.OW, d.Cout, d.CoutPad = B, OH, OW, OH, OW, pk.Cout, pk.CoutPad
    d.split_c0 = pk.split_c0
    d.KH = d.KW = d.stride = 1
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.act, d.epi, d.mode = act, epi, CONV_F16X3
    d.wpatch16, d.wscale16 = pk.wpatch16.data_ptr(), pk.wscale16.data_ptr()
    d.guard = _guard(dev).data_ptr()
    if _ksplit_on() and B * OH * OW <= KSPLIT_MAX_PIXELS and pk.Cout > 4 and not want_stats and not pk.split_c0:
        ws = _ksplit_ws(4 * B * pk.Cout * OH * OW, dev)
        d.kws, d.kws_elems = ws.data_ptr(), ws.numel()
    if isinstance(e0, S16):     # residual operand kept pre-split only (the encoders' block input)
        if tuple(e0.shape) != (B, pk.Cout, OH, OW):
            raise RuntimeError("conv2d_multi: S16 e0 shape %s != %s" % (e0.shape, (B, pk.Cout, OH, OW)))
        d.e0_bs, d.e0, d.e0_fmt = e0.bs, e0.ptr(), 1
        d.kws, d.kws_elems = None, 0
    elif e0 is not None:
        d.e0_bs, d.e0 = _plane4(e0, "e0"), e0.data_ptr()
    if e1 is not None:
        d.e1_bs, d.e1 = _plane4(e1, "e1"), e1.data_ptr()
    if out2 is not None:
        d.out2_bs, d.out2 = _plane4(out2, "out2"), out2.data_ptr()
    if pre is not None:
        if epi not in (EPI_GRU_ZR, EPI_GRU_Q) or tuple(pre.shape) != (B, pk.Cout, OH, OW):
            raise RuntimeError("conv2d_multi: `pre` is a (B, Cout, OH, OW) addend of the GRU epilogues")
        d.pre_bs, d.pre = _plane4(pre, "pre"), pre.data_ptr()
    stats = None
    if want_stats:
        if act != ACT_NONE or epi != EPI_STORE:
            raise RuntimeError("conv2d_multi: statistics are gathered for plain convolutions only (store, no activation)")
        slots = lib.accflow_conv_stat_slots(ctypes.byref(d))
        if slots > 0:
            stats = ConvStats(torch.empty((B, pk.Cout, slots, 3), dtype=torch.float32, device=dev), slots)
            d.stats, d.stat_slots = stats.partial.data_ptr(), slots
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("conv2d") else None
    _check(lib.accflow_conv2d_f32(ctypes.byref(d), _stream()), "accflow_conv2d_f32 (multi-source S16)")
    if t0 is not None:
        g0 = pk.geo[0]
        tm.end("conv2d", t0, pk.flop_per_px * B * OH * OW,
               "multi[%s] Cout%d step%d B%d %dx%d" % ("+".join("%dx%dx%d" % (c, g[3], g[4]) for c, g in zip(pk.C, pk.geo)),
                                                    pk.Cout, g0[0], B, OH, OW))
    ret = out if (out is not None and (fp32_out or out16 is None)) else out16
    return (ret, stats) if want_stats else ret


USE_DEFORM_S16_COLUMNS = os.environ.get("ACCFLOW_DEFORM_S16_COLUMNS", "1") == "1"   # (0: fp32 columns + sigmoid pass, A/B)


def deform_conv2d_s16(pk, x, offset, dmask, out16, mask_is_logit=False):
    """Modulated deformable convolution (torchvision.ops.deform_conv2d semantics, AccFlow_.py:104) with a PRE-SPLIT result:
    the deformed im2col columns, then the 1x1 matrix-core convolution over them writing the ops.S16 tensor `out16` only.
    Round 6: the columns themselves are written pre-split (accflow_deform_columns_s16: the 1x1 convolution stages them by LDS
    DMA), and with mask_is_logit the modulation's sigmoid (AccFlow_.py:103) is applied while sampling."""
    lib = _lib.load()
    if pk.zcols is None:
        raise RuntimeError("deform_conv2d_s16: needs a tap-major pack with the column form (stride 1)")
    B, C, H, W = x.shape
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("conv2d") else None
    if USE_DEFORM_S16_COLUMNS and C % 8 == 0 and (pk.KH * pk.KW * C) % 32 == 0:
        cols = S16.empty(B, pk.KH * pk.KW * C, H, W, x.device)
        _check(lib.accflow_deform_columns_s16(_p(x), _plane4(x, "x"), _p(offset), _plane4(offset, "offset"), _p(dmask),
                                              _plane4(dmask, "dmask"), int(bool(mask_is_logit)), ctypes.c_void_p(cols.ptr()), cols.bs,
                                              _p(_guard(x.device)), B, C, H, W, pk.KH, pk.KW, pk.padH, pk.padW, _stream()),
               "accflow_deform_columns_s16")
    else:
        if mask_is_logit:
            dmask = activation_(dmask, ACT_SIGMOID)
        cols = torch.empty((B, pk.KH * pk.KW * C, H, W), dtype=torch.float32, device=x.device)
        _check(lib.accflow_deform_columns_f32(_p(x), _plane4(x, "x"), _p(offset), _plane4(offset, "offset"), _p(dmask),
                                              _plane4(dmask, "dmask"), _p(cols), B, C, H, W, pk.KH, pk.KW, pk.padH, pk.padW,
                                              _stream()), "accflow_deform_columns_f32")
    if t0 is not None:
        tm.end("conv2d", t0, 0.0, "deform_columns C%d k%dx%d B%d %dx%d" % (C, pk.KH, pk.KW, B, H, W))
    return conv2d(pk.zcols, cols, out16=out16, fp32_out=False)


LOOKUP_BYTES_PER_PX = 4 * 100 * 4 + 8 + 324 * 4  # = 2904, SURVEY.md 8(d)
LOOKUP_S16_CHANNELS = 4 * 88   # S16 lookup output: 88 channels per level (81 taps in (row, column) order + 7 zeros)


def corr_pyramid_shapes(H8, W8, levels=4):
    return [(H8 >> l, W8 >> l) for l in range(levels)]


def corr_volume(fmap1, fmap2, mode=None):
    """-> list of 4 tensors (B*H8*W8, 1, Hl, Wl), the layout CorrBlock.corr_pyramid has."""
    lib = _lib.load()
    _plane4(fmap1, "fmap1"), _plane4(fmap2, "fmap2")
    fmap1, fmap2 = _dense(fmap1, "fmap1"), _dense(fmap2, "fmap2")
    B, C, H8, W8 = fmap1.shape
    P = H8 * W8
    lv = [torch.empty((B * P, 1, h, w), dtype=torch.float32, device=fmap1.device)
          for (h, w) in corr_pyramid_shapes(H8, W8)]
    md = current_mode() if mode is None else mode
    if md == CONV_F32:
        _check(lib.accflow_corr_volume_f32(_p(fmap1), _p(fmap2), _p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]),
                                           B, C, H8, W8, _stream()), "accflow_corr_volume_f32")
    else:
        ws = torch.empty(lib.accflow_corr_volume_ws_bytes(C, H8, W8), dtype=torch.uint8, device=fmap1.device)
        _check(lib.accflow_corr_volume_split_f32(_p(fmap1), _p(fmap2), _p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]),
                                                 _p(ws), md, B, C, H8, W8, _stream()), "accflow_corr_volume_split_f32")
    return lv


class DispPyramid:
    """The 4 pyramid levels in the displacement-indexed hot-path layout (see csrc/corr_disp.hip):
    levels[l] is (B, PB, Hl, Wl, 128), PB = ceil(H8*W8 / 128), with
    levels[l][b, p // 128, dy, dx, p % 128] = corr_pyramid[l][b*P + p, 0, y', x'],
    dy = (y' - (y1 >> l)) mod Hl, dx = (x' - (x1 >> l)) mod Wl, p = y1*W8 + x1 (entries of p >= P are padding)."""

    def __init__(self, levels, B, H8, W8):
        self.levels, self.B, self.H8, self.W8 = levels, B, H8, W8

    @staticmethod
    def _index(l, H8, W8, device):
        """(Hl*Wl, P) gather indices into a row-major plane: idx[dy*Wl + dx, p] = y'*Wl + x'."""
        Hl, Wl = H8 >> l, W8 >> l
        p = torch.arange(H8 * W8, device=device)
        y1l, x1l = (p // W8) >> l, (p % W8) >> l
        yy = (torch.arange(Hl, device=device)[:, None, None] + y1l[None, None, :]) % Hl
        xx = (torch.arange(Wl, device=device)[None, :, None] + x1l[None, None, :]) % Wl
        return (yy * Wl + xx).reshape(Hl * Wl, H8 * W8)

    def _unblocked(self, t, Hl, Wl):
        """(B, PB, Hl, Wl, 128) -> (B, Hl*Wl, P)"""
        P = self.H8 * self.W8
        return t.permute(0, 2, 3, 1, 4).reshape(self.B, Hl * Wl, -1)[:, :, :P]

    def to_rowmajor(self):
        """-> list of (B*P, 1, Hl, Wl) tensors, the reference's corr_pyramid (test / API helper, not the hot path)."""
        out = []
        P = self.H8 * self.W8
        for l, t in enumerate(self.levels):
            Hl, Wl = self.H8 >> l, self.W8 >> l
            idx = self._index(l, self.H8, self.W8, t.device)                           # [d, p] -> cell
            v = torch.empty((self.B, P, Hl * Wl), dtype=t.dtype, device=t.device)
            v.scatter_(2, idx.t()[None].expand(self.B, -1, -1), self._unblocked(t, Hl, Wl).transpose(1, 2))
            out.append(v.view(self.B * P, 1, Hl, Wl))
        return out

    @classmethod
    def from_rowmajor(cls, pyramid, B, H8, W8):
        """Permute a reference-layout pyramid into this layout with plain indexing (tests)."""
        P = H8 * W8
        PB = (P + 127) // 128
        lv = []
        for l, t in enumerate(pyramid):
            Hl, Wl = H8 >> l, W8 >> l
            idx = cls._index(l, H8, W8, t.device)
            g = torch.gather(t.reshape(B, P, Hl * Wl), 2, idx.t()[None].expand(B, -1, -1))  # [b, p, d]
            e = torch.zeros((B, Hl * Wl, PB * 128), dtype=t.dtype, device=t.device)
            e[:, :, :P] = g.transpose(1, 2)
            lv.append(e.view(B, Hl, Wl, PB, 128).permute(0, 3, 1, 2, 4).contiguous())
        return cls(lv, B, H8, W8)


def corr_disp_supported(H8, W8):
    return bool(_lib.load().accflow_corr_disp_supported(H8, W8))


def corr_volume_disp(fmap1, fmap2, mode=None):
    lib = _lib.load()
    _plane4(fmap1, "fmap1"), _plane4(fmap2, "fmap2")
    fmap1, fmap2 = _dense(fmap1, "fmap1"), _dense(fmap2, "fmap2")
    B, C, H8, W8 = fmap1.shape
    md = current_mode() if mode is None else mode
    if md == CONV_F32 or not lib.accflow_corr_disp_supported(H8, W8):
        raise RuntimeError("corr_volume_disp: needs a split conv mode (f16x3 / bf16x6 / bf16x3) and H8, W8 >= 8")
    guard = _guard(fmap1.device) if md == CONV_F16X3 else None
    PB = (H8 * W8 + 127) // 128
    # (the padding lanes p >= P of the last 128-pixel block are never written and never read: no zero fill - at 720x1280
    # that was 4.2 GB of memset per estimator call)
    lv = [torch.empty((B, PB, h, w, 128), dtype=torch.float32, device=fmap1.device) for (h, w) in corr_pyramid_shapes(H8, W8)]
    ws = torch.empty(lib.accflow_corr_volume_ws_bytes(C, H8, W8), dtype=torch.uint8, device=fmap1.device)
    _check(lib.accflow_corr_volume_disp_f32(_p(fmap1), _p(fmap2), _p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]), _p(ws), md,
                                            _p(guard), B, C, H8, W8, _stream()), "accflow_corr_volume_disp_f32")
    return DispPyramid(lv, B, H8, W8)


class CorrPacks:
    """Per-frame operand packs of the displaced correlation GEMM (accflow_corr_pack_f32): frame f of a (F, C, H8, W8)
    feature tensor split once; any number of (query frame, target frame) pairs can be correlated from them."""
    __slots__ = ("data", "F", "C", "H8", "W8", "mode")

    def __init__(self, data, F, C, H8, W8, mode):
        self.data, self.F, self.C, self.H8, self.W8, self.mode = data, F, C, H8, W8, mode


def corr_packs_supported(C, H8, W8, mode=None):
    md = current_mode() if mode is None else mode
    return md != CONV_F32 and C >= 16 and C % 16 == 0 and W8 % 2 == 0 and corr_disp_supported(H8, W8)


def corr_pack(fmaps, mode=None):
    lib = _lib.load()
    fmaps = _dense(fmaps, "fmaps")
    F, C, H8, W8 = fmaps.shape
    md = current_mode() if mode is None else mode
    if not corr_packs_supported(C, H8, W8, md):
        raise RuntimeError("corr_pack: needs a split conv mode, C % 16 == 0, even W8 and H8, W8 >= 8")
    data = torch.empty(F * lib.accflow_corr_pack_bytes(C, H8, W8), dtype=torch.uint8, device=fmaps.device)
    guard = _guard(fmaps.device) if md == CONV_F16X3 else None
    _check(lib.accflow_corr_pack_f32(_p(fmaps), _p(data), md, _p(guard), F, C, H8, W8, _stream()), "accflow_corr_pack_f32")
    return CorrPacks(data, F, C, H8, W8, md)


def corr_volume_disp_packed(packs, idx1, idx2):
    """DispPyramid of the pairs (query frame idx1[b], target frame idx2[b]) from per-frame packs."""
    lib = _lib.load()
    B = len(idx1)
    if B == 0 or len(idx2) != B or max(max(idx1), max(idx2)) >= packs.F or min(min(idx1), min(idx2)) < 0:
        raise RuntimeError("corr_volume_disp_packed: bad frame indices")
    H8, W8, dev = packs.H8, packs.W8, packs.data.device
    PB = (H8 * W8 + 127) // 128
    lv = [torch.empty((B, PB, h, w, 128), dtype=torch.float32, device=dev) for (h, w) in corr_pyramid_shapes(H8, W8)]
    guard = _guard(dev) if packs.mode == CONV_F16X3 else None
    a1, a2 = (ctypes.c_int * B)(*idx1), (ctypes.c_int * B)(*idx2)
    _check(lib.accflow_corr_volume_disp_packed_f32(_p(packs.data), packs.F, a1, a2, _p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]),
                                                   packs.mode, _p(guard), B, packs.C, H8, W8, _stream()),
           "accflow_corr_volume_disp_packed_f32")
    return DispPyramid(lv, B, H8, W8)


def corr_disp_pool(lvl0, H8, W8):
    """Levels 1..3 of a displaced level 0 (B, PB, H8, W8, 128) -> DispPyramid."""
    lib = _lib.load()
    lvl0 = _dense(lvl0, "lvl0")
    B, PB = lvl0.shape[:2]
    lv = [lvl0] + [torch.zeros((B, PB, h, w, 128), dtype=torch.float32, device=lvl0.device)
                   for (h, w) in corr_pyramid_shapes(H8, W8)[1:]]
    _check(lib.accflow_corr_disp_pool_f32(_p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]), B, H8, W8, _stream()),
           "accflow_corr_disp_pool_f32")
    return DispPyramid(lv, B, H8, W8)


def corr_lookup(pyramid, coords, out=None):
    if isinstance(pyramid, DispPyramid):
        return _corr_lookup_alt(pyramid, coords, out, "accflow_corr_lookup_disp_f32")
    return _corr_lookup_rowmajor(pyramid, coords, out)


def _corr_lookup_alt(pyr, coords, out, entry):
    lib = _lib.load()
    coords = _dense(coords, "coords")
    B, _, H8, W8 = coords.shape
    if (B, H8, W8) != (pyr.B, pyr.H8, pyr.W8):
        raise RuntimeError("corr_lookup: coords %s do not match the pyramid (%d,%d,%d)" % (tuple(coords.shape), pyr.B, pyr.H8, pyr.W8))
    if out is None:
        out = torch.empty((B, 324, H8, W8), dtype=torch.float32, device=coords.device)
    out_bs = _plane4(out, "out")
    lv = pyr.levels
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("corr_lookup") else None
    _check(getattr(lib, entry)(_p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]), _p(coords), _p(out), out_bs,
                               B, H8, W8, _stream()), entry)
    if t0 is not None:
        tm.end("corr_lookup", t0, LOOKUP_BYTES_PER_PX * B * H8 * W8)
    return out


def to_s16(src, dst16=None):
    """fp32 (B, C, H, W) (channel block dense) -> ops.S16 with the kernels' split (guard-checked)."""
    lib = _lib.load()
    sbs = _plane4(src, "src")
    B, C, H, W = src.shape
    if dst16 is None:
        dst16 = S16.empty(B, C, H, W, src.device)
    if tuple(dst16.shape) != (B, C, H, W):
        raise RuntimeError("to_s16: shape mismatch")
    _check(lib.accflow_to_s16_f32(_p(src), sbs, ctypes.c_void_p(dst16.ptr()), dst16.bs, _p(_guard(src.device)), B, C, H * W,
                                  _stream()), "accflow_to_s16_f32")
    return dst16


def _replay(cache, name):
    """Re-issue a launch whose argument list a previous call with the same `cache` = (dict, key) stored (same tensors: the
    refinement iterations); None when there is nothing to replay."""
    if cache is None or profiler.ACTIVE is not None:
        return None
    hit = cache[0].get(cache[1])
    if hit is None:
        return None
    _check(hit[0](*hit[1], _stream()), name + " (cached)")
    return hit[2]


def corr_lookup_s16(pyr, coords, out16, cache=None):
    """Displaced-layout lookup writing the S16 form consumed by convc1's S16 pack: 352 channels = 4 levels x 88, channel
    l*88 + j*9 + i = tap (i along x, j along y) of level l (reference channel l*81 + i*9 + j, raft/corr.py:34-45), the 7
    tail channels of each level zero."""
    r = _replay(cache, "accflow_corr_lookup_disp_s16")
    if r is not None:
        return r
    lib = _lib.load()
    if not isinstance(pyr, DispPyramid):
        raise RuntimeError("corr_lookup_s16: needs the displaced pyramid")
    coords = _dense(coords, "coords")
    B, _, H8, W8 = coords.shape
    if (B, H8, W8) != (pyr.B, pyr.H8, pyr.W8) or tuple(out16.shape) != (B, LOOKUP_S16_CHANNELS, H8, W8):
        raise RuntimeError("corr_lookup_s16: shape mismatch")
    lv = pyr.levels
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("corr_lookup") else None
    args = (_p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]), _p(coords), ctypes.c_void_p(out16.ptr()), out16.bs,
            _p(_guard(coords.device)), B, H8, W8)
    _check(lib.accflow_corr_lookup_disp_s16(*args, _stream()), "accflow_corr_lookup_disp_s16")
    if t0 is not None:
        tm.end("corr_lookup", t0, LOOKUP_BYTES_PER_PX * B * H8 * W8)
    elif cache is not None and tm is None:
        cache[0][cache[1]] = (lib.accflow_corr_lookup_disp_s16, args, out16, (pyr, coords))
    return out16


LOOKUP_FUSED_K = 336   # reduction length of the fused lookup -> convc1 kernel (10 x 32 + 16; accflow_corr_lookup_convc1_kpad)


def lookup_fused_weight(w):
    """convc1's (Cout, 324, 1, 1) weight re-indexed to the reduction order of accflow_corr_lookup_convc1_s16
    (include/accflow_hip.h): tap n = j*9 + i of level l (reference channel l*81 + i*9 + j, raft/corr.py:34-45) sits at
    k = 32*(n // 8) + 8*l + n % 8 for n < 80 and at k = 320 + l for n = 80; k = 324..335 are zero."""
    co, ci, kh, kw = w.shape
    if (ci, kh, kw) != (324, 1, 1):
        raise ValueError("lookup_fused_weight: a 1x1 convolution over 4 x 81 correlation channels")
    w4 = w.detach().float().reshape(co, 4, 9, 9).transpose(2, 3).reshape(co, 4, 81)     # [co][l][n = j*9 + i]
    out = torch.zeros((co, LOOKUP_FUSED_K), dtype=torch.float32, device=w.device)
    out[:, :320] = w4[:, :, :80].reshape(co, 4, 10, 8).permute(0, 2, 1, 3).reshape(co, 320)   # [co][c][l][t]
    out[:, 320:324] = w4[:, :, 80]
    return out.reshape(co, LOOKUP_FUSED_K, 1, 1)


def corr_lookup_convc1(pyr, coords, pk, out16=None, out=None, act=ACT_RELU, cache=None):
    """act(convc1(CorrBlock lookup)) in ONE kernel (accflow_corr_lookup_convc1_s16): the 4 x 81 taps never reach HBM.
    pk: PackedConv of lookup_fused_weight(convc1.weight); out16: ops.S16 (B, 256, H8, W8) and / or out: fp32."""
    r = _replay(cache, "accflow_corr_lookup_convc1_s16")
    if r is not None:
        return r
    lib = _lib.load()
    if not isinstance(pyr, DispPyramid):
        raise RuntimeError("corr_lookup_convc1: needs the displaced pyramid")
    if current_mode() != CONV_F16X3 or pk.wpatch16 is None or pk.Cin != LOOKUP_FUSED_K or pk.Cout != 256:
        raise RuntimeError("corr_lookup_convc1: needs the f16x3 mode and the fused pack of a 324 -> 256 1x1 convolution")
    coords = _dense(coords, "coords")
    B, _, H8, W8 = coords.shape
    if (B, H8, W8) != (pyr.B, pyr.H8, pyr.W8) or (out16 is None and out is None):
        raise RuntimeError("corr_lookup_convc1: shape mismatch")
    if out16 is not None and tuple(out16.shape) != (B, pk.Cout, H8, W8):
        raise RuntimeError("corr_lookup_convc1: out16 shape %s" % (out16.shape,))
    out_bs = 0
    if out is not None:
        out_bs = _plane4(out, "out")
        if tuple(out.shape) != (B, pk.Cout, H8, W8):
            raise RuntimeError("corr_lookup_convc1: out shape %s" % (tuple(out.shape),))
    lv = pyr.levels
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("lookup_convc1") else None
    args = (_p(lv[0]), _p(lv[1]), _p(lv[2]), _p(lv[3]), _p(coords), _p(pk.wpatch16), _p(pk.wscale16),
            _p(pk.bias) if pk.bias is not None else None,
            ctypes.c_void_p(out16.ptr()) if out16 is not None else None, out16.bs if out16 is not None else 0,
            _p(out) if out is not None else None, out_bs, int(act), _p(_guard(coords.device)), B, H8, W8, pk.Cout)
    _check(lib.accflow_corr_lookup_convc1_s16(*args, _stream()), "accflow_corr_lookup_convc1_s16")
    ret = out16 if out16 is not None else out
    if t0 is not None:
        # work = the reference's convc1 flop (324 input channels); detail carries the lookup side's algorithmic bytes
        tm.end("lookup_convc1", t0, 2.0 * 324 * pk.Cout * B * H8 * W8, "lookup+convc1 B%d %dx%d" % (B, H8, W8),
               work_exec=2.0 * LOOKUP_FUSED_K * pk.Cout * B * H8 * W8)
    elif cache is not None and tm is None:
        cache[0][cache[1]] = (lib.accflow_corr_lookup_convc1_s16, args, ret, (pyr, coords, pk, out16, out))
    return ret


def flow_from_coords_s16(coords1, dst0, dst1, stack16, motion16, motion_ch, is_flow=False, cache=None):
    """flow = coords1 - grid into the fp32 slices dst0 / dst1 (either may be None), its row-shifted 16-channel stack into
    the S16 tensor stack16 (convf1's 1x7 input) and the two flow channels into channels motion_ch, motion_ch + 1 of the
    S16 tensor motion16 (the tail of RAFT's motion features, update.py:96: cat[out, flow])."""
    if _replay(cache, "accflow_flow_from_coords_s16") is not None:
        return
    lib = _lib.load()
    coords1 = _dense(coords1, "coords1")
    B, _, H8, W8 = coords1.shape
    b0 = _plane4(dst0, "dst0") if dst0 is not None else 0
    b1 = _plane4(dst1, "dst1") if dst1 is not None else 0
    if tuple(stack16.shape) != (B, 16, H8, W8) or motion_ch % 2 or (
            motion16 is not None and (motion16.shape[0] != B or tuple(motion16.shape[2:]) != (H8, W8))):
        raise RuntimeError("flow_from_coords_s16: shape mismatch")
    args = (_p(coords1), _p(dst0), b0, _p(dst1), b1, ctypes.c_void_p(stack16.ptr()), stack16.bs,
            ctypes.c_void_p(motion16.ptr() if motion16 is not None else 0), motion16.bs if motion16 is not None else 0,
            int(motion_ch), _p(_guard(coords1.device)), int(bool(is_flow)), B, H8, W8)
    _check(lib.accflow_flow_from_coords_s16(*args, _stream()), "accflow_flow_from_coords_s16")
    if cache is not None and profiler.ACTIVE is None:
        cache[0][cache[1]] = (lib.accflow_flow_from_coords_s16, args, True, (coords1, dst0, dst1, stack16, motion16))


def _corr_lookup_rowmajor(pyramid, coords, out=None):
    lib = _lib.load()
    coords = _dense(coords, "coords")
    B, _, H8, W8 = coords.shape
    if out is None:
        out = torch.empty((B, 324, H8, W8), dtype=torch.float32, device=coords.device)
    out_bs = _plane4(out, "out")
    for t in pyramid:
        _dense(t, "pyramid level")
    tm = profiler.ACTIVE
    t0 = tm.begin() if tm is not None and tm.wants("corr_lookup") else None
    _check(lib.accflow_corr_lookup_f32(_p(pyramid[0]), _p(pyramid[1]), _p(pyramid[2]), _p(pyramid[3]),
                                       _p(coords), _p(out), out_bs, B, H8, W8, _stream()),
           "accflow_corr_lookup_f32")
    if t0 is not None:  # algorithmic bytes: 4 levels x 10x10 fp32 window + 8 B coords + 324 fp32 written per px
        tm.end("corr_lookup", t0, LOOKUP_BYTES_PER_PX * B * H8 * W8)
    return out


def convex_upsample(flow, mask, out=None):
    lib = _lib.load()
    fbs, mbs = _plane4(flow, "flow"), _plane4(mask, "mask")
    B, _, H8, W8 = flow.shape
    if mask.shape[1] != 576 or mask.shape[2:] != flow.shape[2:]:
        raise RuntimeError("convex_upsample: mask must be (B,576,H8,W8)")
    if out is None:
        out = torch.empty((B, 2, 8 * H8, 8 * W8), dtype=torch.float32, device=flow.device)
    elif tuple(out.shape) != (B, 2, 8 * H8, 8 * W8) or not out.is_contiguous():
        raise RuntimeError("convex_upsample: out must be a contiguous (B,2,8*H8,8*W8) tensor (a batch slice is fine)")
    _check(lib.accflow_convex_upsample_f32(_p(flow), fbs, _p(mask), mbs, _p(out), B, H8, W8, _stream()),
           "accflow_convex_upsample_f32")
    return out


def backwarp(img, flow, out=None):
    lib = _lib.load()
    ibs, fbs = _plane4(img, "image"), _plane4(flow, "flow")
    B, C, H, W = img.shape
    if tuple(flow.shape) != (B, 2, H, W):
        raise RuntimeError("backwarp: flow must be (N,2,H,W) matching image")
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=img.device)
    obs = _plane4(out, "out")
    _check(lib.accflow_backwarp_f32(_p(img), ibs, _p(flow), fbs, _p(out), obs, B, C, H, W, _stream()),
           "accflow_backwarp_f32")
    return out


def compose_flow(step, acc):
    """Flow a -> c from step = a -> b and acc = b -> c, both (B,2,H,W): step + backwarp(acc, step)."""
    lib = _lib.load()
    sbs, abs_ = _plane4(step, "step"), _plane4(acc, "acc")
    B, C, H, W = step.shape
    if C != 2 or tuple(acc.shape) != (B, 2, H, W):
        raise RuntimeError("compose_flow: both flows must be (B,2,H,W)")
    out = torch.empty((B, 2, H, W), dtype=torch.float32, device=step.device)
    _check(lib.accflow_compose_flow_f32(_p(step), sbs, _p(acc), abs_, _p(out), 2 * H * W, B, H, W, _stream()),
           "accflow_compose_flow_f32")
    return out


def get_occ(flow, i1, i2, binary=True, out=None, out16=None):
    """out16: an ops.S16 of 1 (binary) / C channels that receives the map pre-split INSTEAD of the fp32 tensor (the fusion
    chain: both maps feed convolutions only) - returns it."""
    lib = _lib.load()
    fbs, b1, b2 = _plane4(flow, "flow"), _plane4(i1, "I1"), _plane4(i2, "I2")
    B, C, H, W = i1.shape
    if out16 is not None:
        if tuple(out16.shape) != (B, 1 if binary else C, H, W):
            raise RuntimeError("get_occ: out16 shape %s" % (out16.shape,))
        _check(lib.accflow_get_occ_s16(_p(flow), fbs, _p(i1), b1, _p(i2), b2, ctypes.c_void_p(out16.ptr()), out16.bs,
                                       _p(_guard(i1.device)), B, C, H, W, int(bool(binary)), _stream()), "accflow_get_occ_s16")
        return out16
    if out is None:
        out = torch.empty((B, 1 if binary else C, H, W), dtype=torch.float32, device=i1.device)
    obs = _plane4(out, "out")
    _check(lib.accflow_get_occ_f32(_p(flow), fbs, _p(i1), b1, _p(i2), b2, _p(out), obs, B, C, H, W,
                                   int(bool(binary)), _stream()), "accflow_get_occ_f32")
    return out


def downflow8(flow):
    lib = _lib.load()
    _plane4(flow, "flow")
    flow = _dense(flow, "flow")
    B, C, H, W = flow.shape
    if H % 8 or W % 8:
        raise AssertionError("downflow8: H and W must be multiples of 8")
    out = torch.empty((B, C, H // 8, W // 8), dtype=torch.float32, device=flow.device)
    _check(lib.accflow_downflow8_f32(_p(flow), _p(out), B, C, H, W, _stream()), "accflow_downflow8_f32")
    return out


def instance_stats_finalize(stats, eps=1e-5, c0=0, C=None):
    """(B, C, 2) {mean, 1/sqrt(var + eps)} of every plane from a ConvStats (deterministic fixed-order merge); c0 / C: of the
    channels [c0, c0 + C) only (the two convolutions of a PackedMulti.from_strided_with_projection launch share one
    statistics tensor)."""
    lib = _lib.load()
    B, Ctot = stats.partial.shape[:2]
    C = Ctot - c0 if C is None else C
    mr = torch.empty((B, C, 2), dtype=torch.float32, device=stats.partial.device)
    _check(lib.accflow_instance_stats_finalize_sub_f32(_p(stats.partial), stats.slots, Ctot, int(c0), _p(mr), B, C, float(eps),
                                                       _stream()), "accflow_instance_stats_finalize_sub_f32")
    return mr


def instance_norm_proj(x, stats, proj, proj_stats, proj_c0, out16, eps=1e-5):
    """out16 = relu(norm(proj) + relu(norm(x))) (extractor.py:59-63 with a projected block input): x = conv2's raw output
    with its ConvStats; proj = the RAW projection, a channel slice (B, C, H, W) of the tensor the block's split_c0 launch
    wrote, proj_stats that launch's ConvStats and proj_c0 the slice's first channel in them."""
    lib = _lib.load()
    x = _dense(x, "x")
    B, C, H, W = x.shape
    pbs = _plane4(proj, "proj")
    if (tuple(proj.shape) != (B, C, H, W) or tuple(out16.shape) != (B, C, H, W) or tuple(stats.partial.shape[:2]) != (B, C)
            or proj_stats.partial.shape[0] != B or proj_c0 + C > proj_stats.partial.shape[1]):
        raise RuntimeError("instance_norm_proj: shape mismatch")
    mr = torch.empty(4 * B * C, dtype=torch.float32, device=x.device)
    _check(lib.accflow_instance_norm_apply_s16proj_f32(_p(x), _p(stats.partial), stats.slots, _p(proj), pbs, _p(proj_stats.partial),
                                                       proj_stats.slots, proj_stats.partial.shape[1], int(proj_c0), _p(mr),
                                                       ctypes.c_void_p(out16.ptr()), out16.bs, _p(_guard(x.device)), B, C, H * W,
                                                       float(eps), _stream()), "accflow_instance_norm_apply_s16proj_f32")
    return out16


def instance_norm(x, mode, res=None, eps=1e-5, out=None, stats=None, out16=None, fp32_out=True):
    """mode 0: norm(x); 1: relu(norm(x)); 2: relu(res + relu(norm(x))).  In-place when out is None.
    stats: the ConvStats of the convolution that produced x - one pass over x instead of three.
    out16 (needs stats): an ops.S16 that receives the pre-split copy of the result; fp32_out=False then skips the fp32 one."""
    lib = _lib.load()
    x = _dense(x, "x")
    B, C, H, W = x.shape
    if out16 is not None:
        if stats is None or tuple(stats.partial.shape[:2]) != (B, C) or tuple(out16.shape) != (B, C, H, W):
            raise RuntimeError("instance_norm: out16 needs the producing convolution's statistics and a matching S16 tensor")
        o32 = (x if out is None else _dense(out, "out")) if fp32_out else None
        mr = torch.empty(2 * B * C, dtype=torch.float32, device=x.device)
        if isinstance(res, S16):    # mode 2 with the residual kept pre-split only
            if mode != 2 or tuple(res.shape) != (B, C, H, W):
                raise RuntimeError("instance_norm: an S16 residual belongs to mode 2 and must match x")
            _check(lib.accflow_instance_norm_apply_s16res_f32(_p(x), _p(stats.partial), stats.slots, _p(mr),
                                                              ctypes.c_void_p(res.ptr()), res.bs, _p(o32),
                                                              ctypes.c_void_p(out16.ptr()), out16.bs, _p(_guard(x.device)), B, C,
                                                              H * W, float(eps), _stream()), "accflow_instance_norm_apply_s16res_f32")
            return o32 if fp32_out else out16
        if res is not None:
            _dense(res, "res")
        _check(lib.accflow_instance_norm_apply_s16_f32(_p(x), _p(stats.partial), stats.slots, _p(mr), _p(res), _p(o32),
                                                       ctypes.c_void_p(out16.ptr()), out16.bs, _p(_guard(x.device)), B, C, H * W,
                                                       float(eps), int(mode), _stream()), "accflow_instance_norm_apply_s16_f32")
        return o32 if fp32_out else out16
    if out is None:
        out = x
    if res is not None:
        _dense(res, "res")
    if stats is not None:
        if tuple(stats.partial.shape[:2]) != (B, C):
            raise RuntimeError("instance_norm: statistics do not belong to this tensor")
        mr = torch.empty(2 * B * C, dtype=torch.float32, device=x.device)
        _check(lib.accflow_instance_norm_apply_f32(_p(x), _p(stats.partial), stats.slots, _p(mr), _p(res), _p(out), B, C, H * W,
                                                   float(eps), int(mode), _stream()), "accflow_instance_norm_apply_f32")
        return out
    _check(lib.accflow_instance_norm_f32(_p(x), _p(res), _p(out), B, C, H * W, float(eps), int(mode), _stream()),
           "accflow_instance_norm_f32")
    return out


def split_tanh_relu(cnet, net, inp, hd, cd):
    lib = _lib.load()
    cnet = _dense(cnet, "cnet")
    B, _, H, W = cnet.shape
    nbs, ibs = _plane4(net, "net"), _plane4(inp, "inp")
    _check(lib.accflow_split_tanh_relu_f32(_p(cnet), _p(net), nbs, _p(inp), ibs, B, hd, cd, H * W, _stream()),
           "accflow_split_tanh_relu_f32")


def split_tanh_relu_indexed(cnet_items, idx, net, inp, hd, cd):
    """split_tanh_relu with output item b taken from item idx[b] of the contiguous (n_items, hd + cd, H, W) tensor."""
    lib = _lib.load()
    cnet_items = _dense(cnet_items, "cnet")
    n, _, H, W = cnet_items.shape
    B = len(idx)
    nbs, ibs = _plane4(net, "net"), _plane4(inp, "inp")
    if net.shape[0] != B or inp.shape[0] != B:
        raise RuntimeError("split_tanh_relu_indexed: one index per output item")
    arr = (ctypes.c_int * B)(*idx)
    _check(lib.accflow_split_tanh_relu_idx_f32(_p(cnet_items), n, arr, _p(net), nbs, _p(inp), ibs, B, hd, cd, H * W, _stream()),
           "accflow_split_tanh_relu_idx_f32")


def coords_grid(B, H8, W8, device, flow_init=None):
    lib = _lib.load()
    out = torch.empty((B, 2, H8, W8), dtype=torch.float32, device=device)
    if flow_init is not None:
        flow_init = _dense(flow_init.float().contiguous(), "flow_init")
    _check(lib.accflow_coords_grid_f32(_p(out), _p(flow_init), B, H8, W8, _stream()), "accflow_coords_grid_f32")
    return out


def flow_from_coords(coords1, dst0=None, dst1=None, stack16=None, is_flow=False):
    """flow = coords1 - grid into dst0 / dst1 ((B,2,h,w) slices) and, optionally, its row-shifted 16-channel stack
    (see accflow_flow_from_coords_f32; the input of the 7x7 flow convolution as a 1x7 one)."""
    lib = _lib.load()
    coords1 = _dense(coords1, "coords1")
    B, _, H8, W8 = coords1.shape
    b0 = _plane4(dst0, "dst0") if dst0 is not None else 0
    b1 = _plane4(dst1, "dst1") if dst1 is not None else 0
    if stack16 is not None and tuple(_dense(stack16, "stack16").shape) != (B, 16, H8, W8):
        raise RuntimeError("flow_from_coords: stack16 must be (B, 16, h, w)")
    _check(lib.accflow_flow_from_coords_f32(_p(coords1), _p(dst0), b0, _p(dst1), b1, _p(stack16), int(bool(is_flow)), B, H8, W8,
                                            _stream()),
           "accflow_flow_from_coords_f32")


def blend(f1, f2, m, out16=None):
    """out16: an ops.S16 (B, C, H, W) that receives the result pre-split INSTEAD of an fp32 tensor - returns it."""
    lib = _lib.load()
    f1, f2, m = _dense(f1, "f1"), _dense(f2, "f2"), _dense(m, "m")
    B, C, H, W = f1.shape
    if out16 is not None:
        if tuple(out16.shape) != (B, C, H, W):
            raise RuntimeError("blend: out16 shape %s" % (out16.shape,))
        _check(lib.accflow_blend_s16(_p(f1), _p(f2), _p(m), ctypes.c_void_p(out16.ptr()), out16.bs, _p(_guard(f1.device)), B, C,
                                     H * W, _stream()), "accflow_blend_s16")
        return out16
    out = torch.empty_like(f1)
    _check(lib.accflow_blend_f32(_p(f1), _p(f2), _p(m), _p(out), B, C, H * W, _stream()), "accflow_blend_f32")
    return out


def activation_(x, act):
    """In-place activation of a (possibly channel-sliced) tensor."""
    lib = _lib.load()
    xbs = _plane4(x, "x")
    B, C, H, W = x.shape
    _check(lib.accflow_activation_f32(_p(x), xbs, B, C, H * W, int(act), _stream()), "accflow_activation_f32")
    return x


def copy_into(src, dst):
    lib = _lib.load()
    sbs, dbs = _plane4(src, "src"), _plane4(dst, "dst")
    if src.shape != dst.shape:
        raise RuntimeError("copy_into: shape mismatch")
    B, C, H, W = src.shape
    _check(lib.accflow_copy_f32(_p(src), sbs, _p(dst), dbs, B, C, H * W, _stream()), "accflow_copy_f32")
    return dst


def gma_attention(qk, D, scale):
    """qk: (B, 2D, H, W) -> attn (B, 1, P, P)."""
    lib = _lib.load()
    qk = _dense(qk, "qk")
    B, _, H, W = qk.shape
    P = H * W
    attn = torch.empty((B, 1, P, P), dtype=torch.float32, device=qk.device)
    _check(lib.accflow_gma_attention_f32(_p(qk), _p(attn), B, D, P, float(scale), _stream()),
           "accflow_gma_attention_f32")
    return attn


def gma_aggregate(attn, v, fmap, gamma, out=None):
    lib = _lib.load()
    attn, v, fmap = _dense(attn, "attn"), _dense(v, "v"), _dense(fmap, "fmap")
    gamma = _dense(gamma.detach().float().contiguous(), "gamma")
    B, D, H, W = fmap.shape
    if out is None:
        out = torch.empty_like(fmap)
    obs = _plane4(out, "out")
    _check(lib.accflow_gma_aggregate_f32(_p(attn), _p(v), _p(fmap), _p(gamma), _p(out), obs, B, D, H * W, _stream()),
           "accflow_gma_aggregate_f32")
    return out


def gma_attention_t(qk, D, scale):
    """qk: (B, 2D, H, W) -> transposed attention attnT (B, P, P) with attnT[b, j, i] = attn[b, i, j]."""
    lib = _lib.load()
    qk = _dense(qk, "qk")
    B, _, H, W = qk.shape
    P = H * W
    attn_t = torch.empty((B, P, P), dtype=torch.float32, device=qk.device)
    md = current_mode()
    ws = (torch.empty(lib.accflow_gma_attention_ws_bytes(D, H, W), dtype=torch.uint8, device=qk.device)
          if md != CONV_F32 else None)
    _check(lib.accflow_gma_attention_t_f32(_p(qk), _p(attn_t), _p(ws), md, B, D, H, W, float(scale), _stream()),
           "accflow_gma_attention_t_f32")
    return attn_t


def gma_aggregate_t(attn_t, v, fmap, gamma, out=None, mode=None):
    """fmap + gamma * (attn @ v) from the transposed attention, on the split-bf16 matrix cores."""
    lib = _lib.load()
    attn_t, v, fmap = _dense(attn_t, "attnT"), _dense(v, "v"), _dense(fmap, "fmap")
    gamma = _dense(gamma.detach().float().contiguous(), "gamma")
    B, D, H, W = fmap.shape
    if out is None:
        out = torch.empty_like(fmap)
    obs = _plane4(out, "out")
    md = current_mode() if mode is None else mode
    ws = torch.empty(lib.accflow_gma_aggregate_ws_bytes(B, D, H * W), dtype=torch.uint8, device=fmap.device)
    guard = _guard(fmap.device) if md == CONV_F16X3 else None
    _check(lib.accflow_gma_aggregate_t_f32(_p(attn_t), _p(v), _p(fmap), _p(gamma), _p(out), obs, _p(ws), md, _p(guard), B, D, H, W,
                                           _stream()), "accflow_gma_aggregate_t_f32")
    return out


def gma_attention_s16(qk, D, scale):
    """qk: (B, 2D, H, W) -> the attention pre-split for the aggregation GEMM: ops.S16 of P = H*W channels j over the H x W
    pixels i per item (accflow_gma_attention_s16)."""
    lib = _lib.load()
    qk = _dense(qk, "qk")
    B, _, H, W = qk.shape
    P = H * W
    attn16 = S16.empty(B, P, H, W, qk.device)
    ws = torch.empty(lib.accflow_gma_attention_s16_ws_bytes(D, H, W), dtype=torch.uint8, device=qk.device)
    _check(lib.accflow_gma_attention_s16(_p(qk), ctypes.c_void_p(attn16.ptr()), _p(ws), _p(_guard(qk.device)), B, D, H, W,
                                         float(scale), _stream()), "accflow_gma_attention_s16")
    return attn16


def gma_aggregate_s16(attn16_item, v, fmap0, fmap_bs, gamma, out0, out_bs, out16_0, out16_bs, n, D, H, W):
    """fmap + gamma * (attn @ v) for n items sharing one S16 attention; *_0 = data pointers of the first item's slices,
    *_bs the strides (4-byte words) to the next item's (accflow_gma_aggregate_s16)."""
    lib = _lib.load()
    v = _dense(v, "v")
    gamma = _dense(gamma.detach().float().contiguous(), "gamma")
    ws = torch.empty(lib.accflow_gma_aggregate_s16_ws_bytes(n, D, H * W), dtype=torch.uint8, device=v.device)
    _check(lib.accflow_gma_aggregate_s16(ctypes.c_void_p(attn16_item), _p(v), ctypes.c_void_p(fmap0), fmap_bs, _p(gamma),
                                         ctypes.c_void_p(out0), out_bs, ctypes.c_void_p(out16_0), out16_bs, _p(ws),
                                         _p(_guard(v.device)), n, D, H, W, _stream()), "accflow_gma_aggregate_s16")
