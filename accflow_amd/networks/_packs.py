"""Lazily packed device-side weights for the HIP conv kernels.

A module keeps ordinary nn.Conv2d / nn.BatchNorm2d children purely as parameter containers (so that
`state_dict()` keys and shapes equal the reference's and its checkpoints load unchanged); the first
forward on a device packs them into the kernels' layout, and the pack is rebuilt whenever a parameter
is replaced or modified in place (load_state_dict, .to(), optimizer step).
"""
import torch

from .. import ops


def _sig(tensors):
    return tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors if t is not None)


def bn_fold(bn, conv_bias):
    """Eval-mode BatchNorm2d after a conv == per-channel scale of the conv weights + a new bias
    (reference: extractor.py:150-157 with the module in .eval(), test_cvo.py:14,20)."""
    inv = torch.rsqrt(bn.running_var.float() + bn.eps)
    scale = bn.weight.float() * inv if bn.weight is not None else inv
    bias = (conv_bias.float() if conv_bias is not None else 0.0) - bn.running_mean.float()
    bias = bias * scale
    if bn.bias is not None:
        bias = bias + bn.bias.float()
    return scale, bias


def _mark_ready(device):
    """Event behind the pack kernels just queued on the current stream (None without a GPU)."""
    if device.type != "cuda":
        return None
    if torch.cuda.is_current_stream_capturing():
        # Inside a stream capture (train.GraphedForwardBackward rebuilds the trainable packs as part of the graph) an event
        # recorded here could not be queried later; the graph's own dependencies order the pack kernels before every
        # consumer captured after them on this stream or on a stream forked from it afterwards.
        return None
    st = torch.cuda.current_stream(device)
    ev = torch.cuda.Event()
    ev.record(st)
    return [ev, st.cuda_stream, device]


def _wait_ready(entry):
    """A pack is written by kernels on the stream of its first use; a later use on ANOTHER stream (pair-group streams,
    the sequence pipeline's side stream, a caller's own stream) must be ordered behind them.  Costs one event query per
    hit until the pack kernels have completed, nothing afterwards."""
    r = entry[2]
    if r is None:
        return
    if r[0].query():
        entry[2] = None
        return
    cur = torch.cuda.current_stream(r[2])
    if cur.cuda_stream != r[1]:
        cur.wait_event(r[0])


class PackCache:
    def __init__(self):
        self._store = {}

    def clear(self):
        self._store.clear()

    def conv(self, key, conv, bn=None, scale=None, const_scale=None, C0=None, tap_major=False, rows_as_channels=False,
             scale_dep=None, lookup88=False, lookup_fused=False):
        """Pack one nn.Conv2d.  bn: fold an eval BatchNorm2d; scale: per-Cout tensor multiplier, or a callable returning
        it that is evaluated only on a cache miss; scale_dep: the PARAMETER a derived `scale` is computed from
        (ZeroConv2d: exp(3 * scale) is a fresh temporary on every call - the cache must be keyed on the parameter's
        (data_ptr, version), not on the temporary's, or an in-place update of the parameter leaves a stale pack and a
        different allocator block forces a repack under a consumer on another stream);
        const_scale: python float multiplier applied to weights and bias.
        lookup88: the 1x1 convolution over CorrBlock's 4 x 81 lookup channels (convc1, update.py:83) re-indexed for the
        S16 lookup output (ops.corr_lookup_s16): input channel l*88 + j*9 + i takes the weights of reference channel
        l*81 + i*9 + j, the 7 tail channels of each level are zero.
        lookup_fused: the same convolution re-indexed to the reduction order of the fused lookup -> convc1 kernel
        (ops.lookup_fused_weight, 336 entries).
        rows_as_channels: a KH x KW convolution of few input channels re-indexed as a 1 x KW convolution over
        Cin*KH row-shifted channels (padded to 16), w'[co][c*KH + ky][0][kx] = w[co][c][ky][kx] - see
        ops.flow_from_coords(stack16=...)."""
        deps = [conv.weight, conv.bias, scale_dep if scale_dep is not None else (None if callable(scale) else scale)]
        if callable(scale) and scale_dep is None:
            raise ValueError("PackCache.conv: a callable scale needs scale_dep (the parameter it derives from)")
        if bn is not None:
            deps += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        sig = _sig(deps) + (const_scale, C0, tap_major, rows_as_channels, lookup88, lookup_fused)
        key = (key, str(conv.weight.device))  # replicas (nn.DataParallel) share this object across devices
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            _wait_ready(hit)
            return hit[1]
        with torch.no_grad():
            w, b = conv.weight, conv.bias
            sc = None
            if bn is not None:
                sc, b = bn_fold(bn, b)
            if scale is not None:
                s = (scale() if callable(scale) else scale).reshape(-1).float()
                sc = s if sc is None else sc * s
                b = b * s if b is not None else None
            if const_scale is not None:
                cs = torch.full((w.shape[0],), float(const_scale), dtype=torch.float32, device=w.device)
                sc = cs if sc is None else sc * cs
                b = b * float(const_scale) if b is not None else None
            padding = conv.padding
            if lookup88:
                co, ci, kh, kw = w.shape
                if (ci, kh, kw) != (324, 1, 1):
                    raise ValueError("lookup88: a 1x1 convolution over 4 x 81 correlation channels")
                w4 = w.float().reshape(co, 4, 9, 9).transpose(2, 3)        # [co][l][i][j] -> [co][l][j][i]
                w2 = torch.zeros((co, 4, 88), dtype=torch.float32, device=w.device)
                w2[:, :, :81] = w4.reshape(co, 4, 81)
                w = w2.reshape(co, 352, 1, 1)
            if lookup_fused:
                w = ops.lookup_fused_weight(w)
            if rows_as_channels:
                co, ci, kh, kw = w.shape
                w2 = torch.zeros((co, 16, 1, kw), dtype=torch.float32, device=w.device)
                w2[:, :ci * kh, 0] = w.float().reshape(co, ci * kh, kw)
                w, padding = w2, (0, conv.padding[1])
            pk = ops.PackedConv(w, b, stride=conv.stride, padding=padding, scale=sc, C0=C0,
                                tap_major=tap_major)
        self._store[key] = [sig, pk, _mark_ready(conv.weight.device)]
        return pk

    def multi(self, key, conv, bn=None, scale=None, scale_dep=None, splits=None, strided=False, in_ranges=None, with_bias=True):
        """Pack one nn.Conv2d for the multi-source S16 kernel (ops.PackedMulti): `splits` = the channel counts of the
        tensors whose concatenation the conv reads (default: one source), or strided=True for a stride-2 convolution
        read as parity-class sources of ONE tensor (ops.PackedMulti.from_strided).  bn / scale / scale_dep as in conv().
        in_ranges: [(c0, c1), ...] - keep only these INPUT-channel ranges of the weight, one source each, in this order (a
        convolution over a concatenation evaluated as partial sums: AccPlus's members that do not depend on the
        accumulated flow are convolved for all steps at once, AccFlow.fuse_chain); with_bias=False: the partial sum that
        leaves the bias to the other part."""
        deps = [conv.weight, conv.bias, scale_dep if scale_dep is not None else (None if callable(scale) else scale)]
        if callable(scale) and scale_dep is None:
            raise ValueError("PackCache.multi: a callable scale needs scale_dep (the parameter it derives from)")
        if bn is not None:
            deps += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        sig = _sig(deps) + (tuple(splits) if splits else None, bool(strided), tuple(in_ranges) if in_ranges else None, bool(with_bias))
        key = (key, str(conv.weight.device))
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            _wait_ready(hit)
            return hit[1]
        with torch.no_grad():
            w, b = conv.weight.float(), (conv.bias if with_bias else None)
            if in_ranges:
                if splits or strided or bn is not None:
                    raise ValueError("PackCache.multi: in_ranges stands alone (it defines the sources)")
                w = torch.cat([w[:, a:b_] for a, b_ in in_ranges], dim=1).contiguous()
                splits = [b_ - a for a, b_ in in_ranges]
            sc = None
            if bn is not None:
                sc, b = bn_fold(bn, b)
            if scale is not None:
                s = (scale() if callable(scale) else scale).reshape(-1).float()
                sc = s if sc is None else sc * s
                b = b * s if b is not None else None
            stride = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
            if strided:
                if stride != 2:
                    raise ValueError("PackCache.multi: strided=True is for stride-2 convolutions")
                pk = ops.PackedMulti.from_strided(w, b, conv.padding, scale=sc)
            else:
                if stride != 1:
                    raise ValueError("PackCache.multi: a stride-%d convolution needs strided=True" % stride)
                pk = ops.PackedMulti.from_cat(w, b, list(splits) if splits else [w.shape[1]], conv.padding, scale=sc)
        self._store[key] = [sig, pk, _mark_ready(conv.weight.device)]
        return pk

    def multi_proj(self, key, conv, proj, bn=None, bn_proj=None):
        """A residual block's stride-2 3x3 convolution and its 1x1 stride-2 projection of the block input (extractor.py:9,
        52-53: conv1 and downsample[0]) as ONE multi-source pack (ops.PackedMulti.from_strided_with_projection): output
        rows [0, planes) = conv1, [planes, 2 planes) = the projection; bn / bn_proj: eval BatchNorms folded into each."""
        deps = [conv.weight, conv.bias, proj.weight, proj.bias]
        for b_ in (bn, bn_proj):
            if b_ is not None:
                deps += [b_.weight, b_.bias, b_.running_mean, b_.running_var]
        sig = _sig(deps) + (bn is not None, bn_proj is not None)
        key = (key, str(conv.weight.device))
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            _wait_ready(hit)
            return hit[1]
        with torch.no_grad():
            sc, b = (None, conv.bias) if bn is None else bn_fold(bn, conv.bias)
            scp, bp = (None, proj.bias) if bn_proj is None else bn_fold(bn_proj, proj.bias)
            pk = ops.PackedMulti.from_strided_with_projection(conv.weight.float(), b, conv.padding, proj.weight.float(), bp,
                                                              scale=sc, scale_proj=scp)
        self._store[key] = [sig, pk, _mark_ready(conv.weight.device)]
        return pk

    def multi_cat(self, key, convs):
        """Several stride-1 convolutions with the same geometry over ONE input as one multi-source-kernel pack with
        concatenated output channels (FlowDecoder's flow and mask heads both start with a 3x3 conv of f_fuse)."""
        deps = []
        for c in convs:
            deps += [c.weight, c.bias]
        sig = _sig(deps)
        key = (key, str(convs[0].weight.device))
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            _wait_ready(hit)
            return hit[1]
        with torch.no_grad():
            w = torch.cat([c.weight.float() for c in convs], dim=0).contiguous()
            b = torch.cat([c.bias.float() for c in convs], dim=0).contiguous()
            pk = ops.PackedMulti.from_cat(w, b, [w.shape[1]], convs[0].padding)
        self._store[key] = [sig, pk, _mark_ready(convs[0].weight.device)]
        return pk

    def conv_cat(self, key, convs, C0=None, in_slices=None, with_bias=True):
        """Pack several convs sharing one input as ONE conv with concatenated output channels
        (the z and r gates of a GRU half-step, update.py:47-48).  in_slices: list of (start, stop) INPUT-channel
        ranges to keep, in order (a convolution is linear in its input channels: the GRU convs are cut into the part
        fed by the iteration-invariant context features and the rest); with_bias=False drops the bias."""
        deps = []
        for c in convs:
            deps += [c.weight, c.bias]
        sig = _sig(deps) + (C0, tuple(in_slices) if in_slices else None, with_bias)
        key = (key, str(convs[0].weight.device))
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            _wait_ready(hit)
            return hit[1]
        with torch.no_grad():
            w = torch.cat([c.weight.float() for c in convs], dim=0)
            if in_slices:
                w = torch.cat([w[:, a:b] for a, b in in_slices], dim=1)
            w = w.contiguous()
            b = torch.cat([c.bias.float() for c in convs], dim=0).contiguous() if with_bias else None
            pk = ops.PackedConv(w, b, stride=convs[0].stride, padding=convs[0].padding, C0=C0)
        self._store[key] = [sig, pk, _mark_ready(convs[0].weight.device)]
        return pk


def span(items):
    """torch.cat(items, dim=0) WITHOUT a copy when the items are consecutive batch views of one tensor (the per-frame
    outputs of one encoder call, the frames of a sequence stacked once): the view over all of them; a real cat otherwise."""
    items = list(items)
    if len(items) == 1:
        return items[0]
    t0 = items[0]
    off, ok = t0.storage_offset(), True
    for t in items:
        ok &= (t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() and t.storage_offset() == off
               and t.dtype == t0.dtype and t.stride() == t0.stride() and t.shape[1:] == t0.shape[1:] and t.is_contiguous())
        off += t.shape[0] * (t.stride(0) if t.dim() else 0)
    if ok:
        B = sum(t.shape[0] for t in items)
        return t0.as_strided((B,) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())
    return torch.cat(items, dim=0)


def require_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "accflow_amd: this module runs only on an MI355X through libaccflow_hip (got a %s tensor); "
                "there is no CPU path in the product - the CPU restatement lives in oracle/ for tests." % t.device)
