"""RAFT pair-flow estimator on gfx950 kernels.

Interface of the reference's networks/raft/raft.py:25-146: `RAFT(args).forward(image1, image2,
iters=12, flow_init=None) -> (N, 2, H, W) float32`, attributes hidden_dim / context_dim / args,
`upsample_flow`, `freeze_bn`, identical state_dict.  Images are expected already scaled to [-1, 1]
(raft.py:97-98 are commented out in the reference).

Differences in execution, none in results beyond fp32 rounding:
  * fp32-equivalent arithmetic everywhere (the reference autocasts to fp16 on CUDA; its CPU path - the parity oracle - is
    fp32); in the f16x3 mode the refinement iteration's conv-to-conv tensors are held as the fp16 hi + lo pair of their
    fp32 value (ops.S16), which is what the matrix cores multiply either way;
  * the convex-upsampling mask head runs only in the last iteration: the reference evaluates it in all
    `iters` iterations but returns only the last flow_up (raft.py:142-146);
  * `estimate_pairs` lets a caller (AccFlow) encode each frame once and evaluate many (i, j) pairs in
    one batch; per-sample InstanceNorm / eval-BatchNorm make that exact.
"""
import argparse

import torch
import torch.nn as nn

from ... import ops
from .._packs import require_cuda
from . import corr as _corr
from .corr import CorrBlock
from .extractor import BasicEncoder
from . import update as _update
from .update import BasicUpdateBlock, UpdateWorkspace


import os

N_STREAMS = max(1, int(os.environ.get("ACCFLOW_STREAMS", "2")))
# bench.py's per-launch roofline pass sets this: the product path runs the lookup fused with convc1, so the north-star
# kernel ALONE (accflow_corr_lookup_disp_s16, same pyramid, same coords) is launched in addition there to be timed
PROFILE_STANDALONE_LOOKUP = False
USE_CORR_PACKS = os.environ.get("ACCFLOW_CORR_PACKS", "1") == "1"   # per-frame correlation operand packs (0: per pair, A/B)
_STREAMS = {}


# Schedule of the two encoders of estimate_pairs (round 6).  0: fnet, cnet, then the fork into pair groups (rounds 2-5).
# 1: cnet on the second pair-group stream underneath fnet.  2: fnet, the correlation operand packs, THEN cnet on the main
# stream while the pair-group streams - which wait for the packs only - already compute their correlation pyramids (a
# store-bound GEMM under a matrix-bound encoder); the groups wait for cnet where they first need it (_prepare_context).
# 3: both - cnet on the second group's stream from the start, the first group forks at the event behind the packs.
ENCODER_STREAMS = int(os.environ.get("ACCFLOW_ENCODER_STREAMS", "2"))   # (default 2: r06_ab_encoder_schedule.txt)
# Priority of the pair-group streams: -1 = above the caller's and the chain's.  With parallel.SequencePipeline's split mode the
# NEXT sequence's encoder launches (thousands of workgroups) run underneath the iterations (a few hundred per launch) and would
# otherwise hold every workgroup slot: 21.60 -> 21.49 ms per step; one sequence at a time pays 0.2 ms for it
# (profiles/r06_ab_group_priority.txt).  ONE set of streams on purpose: a second pair at another priority makes 7 streams on
# the 4 hardware queues, and streams that share a queue serialise (one sequence at a time: 23.8 -> 30.3 ms, same file).
GROUP_STREAM_PRIORITY = int(os.environ.get("ACCFLOW_GROUP_PRIORITY", "-1"))


def _side_streams(device, n):
    key = (str(device), n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=device, priority=GROUP_STREAM_PRIORITY) for _ in range(n)]
    return _STREAMS[key]


class RAFT(nn.Module):
    KEEP_CONTEXT_RUNS = False     # (RAFTGMA: items out of one image1 share an attention matrix and stay in one pair group)

    def __init__(self, args):
        super().__init__()
        self.args = args
        if getattr(args, "small", False):
            raise NotImplementedError("RAFT-small is not reachable from networks.build_flow_estimator "
                                      "(networks/__init__.py:8 fixes small=False)")
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        args.corr_levels = 4
        args.corr_radius = 4
        if "dropout" not in self.args:
            self.args.dropout = 0
        if "alternate_corr" not in self.args:
            self.args.alternate_corr = False
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=args.dropout)
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn="batch", dropout=args.dropout)
        self.update_block = BasicUpdateBlock(self.args, hidden_dim=hdim)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def initialize_flow(self, img):
        N, _, H, W = img.shape
        return (ops.coords_grid(N, H // 8, W // 8, img.device), ops.coords_grid(N, H // 8, W // 8, img.device))

    def upsample_flow(self, flow, mask):
        """[H/8, W/8, 2] -> [H, W, 2] convex combination (raft.py:81-92)."""
        require_cuda(flow, mask)
        return ops.convex_upsample(flow.float(), mask.float())

    # ------------------------------------------------------------------------------------------
    def _x_dim(self):
        return 256

    def _prepack(self):
        """All weight packs the refinement loop uses, built on the current (main) stream before the fork."""
        self.update_block.prepack()

    def _prepare_context(self, ws, cnet_feat, ids=None):
        """net = tanh(cnet[:, :128]), inp = relu(cnet[:, 128:]) into the workspace (raft.py:116-119).
        cnet_feat: the items' (b, 256, h, w) context-encoder outputs, or (frame-major tensor, item indices) - the pairs of
        a sequence share their image1's features, which are then gathered without a pair-major copy.
        ids: optional per-item keys; items with equal keys carry the SAME context features (same image1)."""
        if isinstance(cnet_feat, tuple):
            ops.split_tanh_relu_indexed(cnet_feat[0], cnet_feat[1], ws.net, ws.inp, self.hidden_dim, self.context_dim)
        else:
            ops.split_tanh_relu(cnet_feat, ws.net, ws.inp, self.hidden_dim, self.context_dim)
        if ws.s16:
            ops.to_s16(ws.net, ws.h16)

    def _lookup_and_flow(self, ws, corr_fn, coords1):
        if ws.s16 and corr_fn.supports_s16():
            # (coords1 is updated in place by the flow head: the same tensors in every iteration -> cached launches)
            ub = self.update_block
            if _update.FUSE_LOOKUP and ub.encoder.convc1.out_channels == 256:
                # relu(convc1(lookup)) in one kernel: the taps go through LDS into the matrix cores, not through HBM
                if PROFILE_STANDALONE_LOOKUP and ops.profiler.ACTIVE is not None:
                    corr_fn.lookup_s16(coords1, ws.corr16)    # (bench.py's roofline pass only: the north-star kernel alone)
                corr_fn.lookup_convc1(coords1, ub._packs.conv("c1f", ub.encoder.convc1, lookup_fused=True), ws.c1_16,
                                      cache=(ws.descs, "lookup"))
                ws.c1_fused = True
            else:
                corr_fn.lookup_s16(coords1, ws.corr16, cache=(ws.descs, "lookup"))
            ops.flow_from_coords_s16(coords1, ws.flow, ws.motion_flow if ws.x_dim > 256 else None, ws.stack16, ws.motion16, 126,
                                     cache=(ws.descs, "flow"))
            return
        if ws.s16:   # (row-major pyramid: fp32 lookup, converted at the boundary)
            corr_fn(coords1, out=ws.corr)
            ws.fill_s16_inputs(coords1, is_flow=False)
            return
        corr_fn(coords1, out=ws.corr)
        ops.flow_from_coords(coords1, dst0=ws.flow, dst1=ws.motion_flow, stack16=ws.flow16)

    def _iteration(self, ws, corr_fn, coords1, last):
        self._lookup_and_flow(ws, corr_fn, coords1)
        return self.update_block.step(ws, coords1, want_mask=last)

    @classmethod
    def _group_cuts(cls, B, n_groups, flip=False, ctx_ids=None):
        """Batch boundaries of the pair groups: floor halves, or ceil halves with `flip` (the pipeline's split mode alternates
        them); RAFTGMA never cuts between two items that share one attention matrix (and therefore ignores `flip`: 5 + 6 is its
        even split of AccFlow's 11 pairs)."""
        flip = flip and not cls.KEEP_CONTEXT_RUNS
        cuts = [(g * B + (n_groups - 1 if flip else 0)) // n_groups for g in range(n_groups + 1)]
        if ctx_ids is not None and cls.KEEP_CONTEXT_RUNS:
            for g in range(1, n_groups):
                while cuts[g] < cuts[g + 1] - 1 and ctx_ids[cuts[g]] == ctx_ids[cuts[g] - 1]:
                    cuts[g] += 1
        return cuts

    def _refine(self, fmap1, fmap2, cnet_feat, iters, flow_init, packed=None, ctx_ids=None, ready=None):
        """Correlation pyramid + `iters` refinement steps + convex upsampling for a batch of pairs.

        Batches of >= 4 pairs are processed as N_STREAMS independent groups on separate HIP streams: the pairs do
        not interact, and with two kernel sequences in flight the tail of one group's kernel (1 320 workgroups on
        768 resident slots = 1.7 rounds at B = 11) is back-filled by the other group's next kernel instead of
        idling the chip until the next launch."""
        if iters < 1:  # the reference would raise NameError on `flow_up`; be explicit
            raise ValueError("iters must be >= 1")
        # packed = (ops.CorrPacks, idx1, idx2): per-frame operand packs of the correlation GEMM + the frame of each
        # pair's queries / targets, instead of pair-major copies of the feature maps
        indexed = isinstance(cnet_feat, tuple)     # (frame-major features, per-item indices)
        B = len(cnet_feat[1]) if indexed else cnet_feat.shape[0]
        dev = (cnet_feat[0] if indexed else cnet_feat).device
        _, _, h, w = (cnet_feat[0] if indexed else cnet_feat).shape
        self._prepack()
        n_groups = N_STREAMS if B >= 4 else 1
        # `ready` may carry the event pair (packs ready, context features ready) and a `flip` flag of a caller that runs this
        # refinement "at home" on the first group stream while ITS stream goes on (parallel.SequencePipeline, split mode): the
        # groups then wait for those events only - never for each other - and the larger share of an odd pair count alternates
        # between the two streams from sequence to sequence, so that neither is the longer one every time.
        ctx_ready, flip = None, False
        if isinstance(ready, tuple):
            ready, ctx_ready, flip = ready
        cuts = self._group_cuts(B, n_groups, flip, ctx_ids)
        bounds = list(zip(cuts[:-1], cuts[1:]))
        main = torch.cuda.current_stream()
        streams = [main] if n_groups == 1 else _side_streams(dev, n_groups)
        # every group upsamples into its slice of ONE output tensor (allocated on the main stream before the fork)
        out_all = torch.empty((B, 2, 8 * h, 8 * w), dtype=torch.float32, device=dev)
        forked = None
        if ctx_ready is not None:    # (a group that never waits for `main` must not write out_all before main's earlier readers)
            forked = torch.cuda.Event()
            forked.record(main)
        state = []
        # ready (ENCODER_STREAMS = 2): an event on the main stream behind the correlation operand packs, with the context-feature
        # encoder queued after it - the groups build their pyramids underneath that encoder and join the main stream later
        early = ready is not None and packed is not None and n_groups > 1
        for (b0, b1), st in zip(bounds, streams):
            if st != main:
                if early:
                    st.wait_event(ready)
                else:
                    st.wait_stream(main)
            with torch.cuda.stream(st):
                if packed is not None:
                    corr_fn = CorrBlock.from_packs(packed[0], packed[1][b0:b1], packed[2][b0:b1])
                else:
                    corr_fn = CorrBlock(fmap1[b0:b1], fmap2[b0:b1], radius=self.args.corr_radius)
                if early and ctx_ready is not None:
                    st.wait_event(ctx_ready)  # the context features and the update block's packs (split mode: no flow_init)
                elif early:
                    st.wait_stream(main)      # the context features, the update block's packs, flow_init
                ws = UpdateWorkspace(b1 - b0, h, w, dev, hidden=self.hidden_dim, x_dim=self._x_dim())
                self._prepare_context(ws, (cnet_feat[0], cnet_feat[1][b0:b1]) if indexed else cnet_feat[b0:b1],
                                      ctx_ids[b0:b1] if ctx_ids is not None else None)
                fi = flow_init[b0:b1] if flow_init is not None else None
                coords1 = ops.coords_grid(b1 - b0, h, w, dev, flow_init=fi)
            state.append((st, corr_fn, ws, coords1))
        masks = [None] * n_groups
        for itr in range(iters):
            for g, (st, corr_fn, ws, coords1) in enumerate(state):
                with torch.cuda.stream(st):
                    masks[g] = self._iteration(ws, corr_fn, coords1, last=(itr == iters - 1))
        for g, (st, corr_fn, ws, coords1) in enumerate(state):
            with torch.cuda.stream(st):
                if forked is not None and early:
                    st.wait_event(forked)
                ops.flow_from_coords(coords1, dst0=ws.flow)
                ops.convex_upsample(ws.flow, masks[g], out=out_all[bounds[g][0]:bounds[g][1]])
            if st != main:
                main.wait_stream(st)
        return out_all

    @torch.no_grad()
    def forward(self, image1, image2, iters=12, flow_init=None):
        """f16x3 conv mode: every STAGE - feature encoder, context encoder, refinement - is recomputed in bf16x6 on its own
        if one of its values left the fp16 split's range (ops.with_range_guard; ops.guard_report() names what fell back)."""
        require_cuda(image1, image2)
        image1 = image1.float().contiguous()
        image2 = image2.float().contiguous()

        def run():
            fmap1, fmap2 = self.fnet([image1, image2])
            cnet_feat = self.cnet(image1)
            return self._refine_guarded(fmap1.contiguous(), fmap2.contiguous(), cnet_feat, iters, flow_init)

        out, tripped = ops.optimistic(run, image1.device)   # (one host synchronisation unless a stage has to fall back)
        return run() if tripped else out

    def _refine_guarded(self, *args, **kwargs):
        dev = next(self.parameters()).device
        return ops.with_range_guard(lambda: self._refine(*args, **kwargs), dev, name="%s.refine" % type(self).__name__)

    @torch.no_grad()
    def encode_frames(self, frames, fnet_ids, cnet_ids, features=None, after_fmap=None):
        """Per-frame encoder outputs {"fmap": {frame: (N,256,h,w)}, "cnet": {frame: (N,256,h,w)}} for the listed frame
        indices (exact to encode once and reuse: InstanceNorm / eval-BatchNorm are per-sample).  `features` is
        extended in place with what it lacks.  (Each encoder call is a guarded stage of its own: BasicEncoder.forward.)"""
        feats = features if features is not None else {}
        if feats.get("mode") != ops.current_mode():  # (inside an enclosing guard's bf16x6 retry: drop what f16x3 produced)
            feats.clear()
            feats.update({"fmap": {}, "cnet": {}, "mode": ops.current_mode()})
        jobs = [(key, enc, [f for f in sorted(set(ids)) if f not in feats[key]])
                for key, enc, ids in (("fmap", self.fnet, fnet_ids), ("cnet", self.cnet, cnet_ids))]
        # Both encoders have work and no stage guard of their own will read a flag (another mode, or inside a guard scope
        # whose one flag every kernel reports to): cnet runs on the second pair-group stream - idle until the refinement
        # starts - underneath fnet, the way the two pair groups run underneath each other (_refine): kernels of different
        # kinds (fnet's normalisation passes, cnet's convolutions) share the chip and back-fill each other's tails.
        dev = frames[0].device
        side = None
        if (ENCODER_STREAMS in (1, 3) and N_STREAMS > 1 and all(todo for _, _, todo in jobs)
                and (ops.current_mode() != ops.CONV_F16X3 or ops.inside_guard())):
            main = torch.cuda.current_stream(dev)
            side = _side_streams(dev, N_STREAMS)[-1]
            side.wait_stream(main)      # the frames / the weights' packs, and every earlier reader of this pool's blocks
        for key, enc, todo in jobs:
            if todo:
                if side is not None and key == "cnet":
                    with torch.cuda.stream(side):
                        outs = enc([frames[f].float().contiguous() for f in todo])
                else:
                    outs = enc([frames[f].float().contiguous() for f in todo])
                # all outputs of ONE encoder call = one frame-major tensor: packable per frame (fmap), gatherable (cnet)
                base = getattr(outs[0], "_base", None)
                whole = not feats[key] and base is not None and base.shape[0] == len(todo) * outs[0].shape[0]
                feats[key + "_base"] = (base, {f: k for k, f in enumerate(todo)}) if whole else None
                if key == "fmap":
                    feats.pop("corr_packs", None)
                feats[key].update(zip(todo, outs))
            if key == "fmap" and after_fmap is not None:
                after_fmap(feats)       # (estimate_pairs: what only needs the feature maps is queued in front of cnet)
        if side is not None:
            main.wait_stream(side)
            # cnet's outputs live in blocks of the side stream's allocator pool and are read on other streams from here on
            base = feats.get("cnet_base")
            for t in ([base[0]] if base is not None else []) + list(feats["cnet"].values()):
                t.record_stream(main)
        return feats

    @torch.no_grad()
    def estimate_pairs(self, frames, pairs, iters=12, flow_init=None, features=None):
        """frames: list of (N,3,H,W); pairs: list of (i, j) = flow from frame i to frame j.
        flow_init: optional (len(pairs)*N, 2, H/8, W/8) start flows, pair-major (raft.py:123-124 per pair);
        features: encoder outputs from encode_frames to reuse across calls (warm-start chaining).
        Returns (len(pairs)*N, 2, H, W), pair-major like torch.cat of per-pair calls."""
        require_cuda(*frames)
        N = frames[0].shape[0]
        feats = self.encode_pairs(frames, pairs, features)
        ready = feats.pop("fmap_event", None)
        home = feats.pop("refine_at_home", None)     # (ctx event, flip) of parallel.SequencePipeline's split mode
        if ready is not None and home is not None and flow_init is None:
            ready = (ready, home[0], home[1])
        return self._estimate_encoded(frames, pairs, feats, ready, iters, flow_init)

    @staticmethod
    def _packs_usable(feats, pairs):
        fb = feats.get("fmap_base")
        return (fb is not None and USE_CORR_PACKS and _corr.LAYOUT == "disp" and all(f in fb[1] for p in pairs for f in p)
                and ops.corr_packs_supported(fb[0].shape[1], fb[0].shape[2], fb[0].shape[3]))

    @torch.no_grad()
    def encode_pairs(self, frames, pairs, features=None):
        """The encoder half of estimate_pairs on the current stream: per-frame features (encode_frames) and, on the
        ENCODER_STREAMS = 2 / 3 schedules, the correlation operand packs with an event behind them (`fmap_event`).  A caller
        that keeps the result may hand it to estimate_pairs(features=) on ANOTHER stream that waited for this one
        (parallel.SequencePipeline: the next sequence's encoders then run underneath this one's refinement)."""
        def after_fmap(feats):
            # ENCODER_STREAMS = 2: the correlation operand packs in FRONT of cnet on the main stream and an event behind them -
            # all the pair-group streams need to start their pyramids.  Only where no stage guard of its own zeroes / reads the
            # thread's flag in between (another mode, or a guard scope whose one flag every kernel reports to).
            if (ENCODER_STREAMS not in (2, 3) or N_STREAMS < 2 or not self._packs_usable(feats, pairs)
                    or not (ops.current_mode() != ops.CONV_F16X3 or ops.inside_guard())):
                return
            cp = feats.get("corr_packs")
            if cp is not None and cp.mode == ops.current_mode() and "fmap_event" in feats:
                return                   # (encoded earlier, possibly on another stream: that event stands)
            if cp is None or cp.mode != ops.current_mode():
                feats["corr_packs"] = ops.corr_pack(feats["fmap_base"][0])
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(frames[0].device))
            feats["fmap_event"] = ev

        feats = self.encode_frames(frames, {i for p in pairs for i in p}, {i for i, _ in pairs}, features, after_fmap=after_fmap)
        self._prepack()      # (first call: the refinement loop's weight packs, on the stream the encoders ran on)
        return feats

    def _estimate_encoded(self, frames, pairs, feats, ready, iters, flow_init):
        N = frames[0].shape[0]
        cb = feats.get("cnet_base")
        if cb is not None and all(i in cb[1] for i, _ in pairs):
            # pair-major context features as a GATHER over the frame-major encoder output (no copy of them)
            cfeat = (cb[0], [cb[1][i] * N + n for i, _ in pairs for n in range(N)])
            n_items, hw8 = len(cfeat[1]), tuple(cb[0].shape[2:])
        else:
            cfeat = torch.cat([feats["cnet"][i] for i, _ in pairs], dim=0)
            n_items, hw8 = cfeat.shape[0], tuple(cfeat.shape[2:])
        assert n_items == N * len(pairs)
        if flow_init is not None:
            require_cuda(flow_init)
            if tuple(flow_init.shape) != (n_items, 2) + hw8:
                raise RuntimeError("estimate_pairs: flow_init must be (len(pairs)*N, 2, H/8, W/8)")
        ctx_ids = [(i, n) for i, _ in pairs for n in range(N)]  # pairs out of the same frame share context features
        fb = feats.get("fmap_base")
        if self._packs_usable(feats, pairs):
            # the feature maps stay frame-major: each frame is split ONCE into the correlation GEMM's operand pack
            # (7 packs for the 11 pairs of a 7-frame sequence; no pair-major copies)
            idx1 = [fb[1][i] * N + n for i, _ in pairs for n in range(N)]
            idx2 = [fb[1][j] * N + n for _, j in pairs for n in range(N)]

            first = [ready]

            def refine_packed():
                # (the operand packs are mode-specific: a bf16x6 retry of this stage splits the feature maps again - behind
                # cnet on the main stream, so only the first run may fork at the event)
                ev = first.pop() if first else None
                cp = feats.get("corr_packs")
                if cp is None or cp.mode != ops.current_mode():
                    cp, ev = ops.corr_pack(fb[0]), None
                    feats["corr_packs"] = cp
                return self._refine(None, None, cfeat, iters, flow_init, packed=(cp, idx1, idx2), ctx_ids=ctx_ids, ready=ev)

            return ops.with_range_guard(refine_packed, frames[0].device, name="%s.refine" % type(self).__name__)
        fmap1 = torch.cat([feats["fmap"][i] for i, _ in pairs], dim=0)
        fmap2 = torch.cat([feats["fmap"][j] for _, j in pairs], dim=0)
        return self._refine_guarded(fmap1, fmap2, cfeat, iters, flow_init, ctx_ids=ctx_ids)


def default_args():
    return argparse.Namespace(small=False, mixed_precision=True)
