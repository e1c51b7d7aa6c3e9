"""Multi-GPU execution of the accumulation path: one process per GPU (torchrun), weights resident on
every rank, work statically partitioned, ONE collective per batch.

This replaces the reference's only parallelism, nn.DataParallel (test_cvo.py:18,26: single process,
parameters re-broadcast on every forward, scatter/gather through GPU 0).  Two modes:

  * sequence-sharded (N sequences >= world size): every rank runs whole sequences; the final accumulated
    flows are gathered to the root with a single `gather` (RCCL over xGMI: <= 3.9 MB per sequence at
    480x1024, one point-to-point transfer per peer, no ring).
  * pair-sharded (fewer sequences than ranks): the 11 independent estimator pairs of a sequence are dealt
    round-robin; their 1/8-resolution flows (61 KB each) are all-gathered and the short fusion chain runs
    on the root.  For a STREAM of sequences the root rotates (run_pair_sharded_stream): every rank does
    the same number of pairs and chains, and a root's chain runs underneath its pairs of the next sequence.

The functions take callables so the partition / collective logic is testable on CPU with the gloo
backend (tests/test_parallel_gloo.py); nothing here touches the data path's arithmetic.
"""
import contextlib
import os

import torch
import torch.distributed as dist


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _collectives(group=None):
    """True when a process group exists: the collectives then run even at world size 1 (one code path whatever the
    number of ranks; a single-GPU box still exercises RCCL)."""
    return dist.is_available() and dist.is_initialized()


def block_partition(n_items, world_size, rank):
    """Contiguous, balanced: the first (n % world) ranks get one extra item."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def round_robin(n_items, world_size, rank):
    return list(range(rank, n_items, world_size))


def deal_pairs(pairs, world_size, keep_together=False):
    """Static deal of estimator pairs over ranks -> list (one entry per rank) of ascending indices into `pairs`.

    keep_together=False: pair k goes to rank k % world_size (balanced to within one pair).
    keep_together=True (GMA): the pairs out of one image1 - (i, i-1) and (i, 0) - stay on one rank, because they share
    one attention matrix (829 MB at 720x1280, built once per distinct image1 and read once per iteration for both pairs
    by the stacked aggregation GEMM); the groups are dealt longest-first to the least loaded rank (ties: lowest rank),
    which is deterministic and the same on every rank."""
    if keep_together:
        order, groups = [], {}
        for k, (i, _) in enumerate(pairs):
            if i not in groups:
                groups[i] = []
                order.append(i)
            groups[i].append(k)
        units = [groups[i] for i in order]
    else:
        units = [[k] for k in range(len(pairs))]
    ranks = [[] for _ in range(world_size)]
    for u in sorted(units, key=lambda u: (-len(u), u[0])):
        r = min(range(world_size), key=lambda r: (len(ranks[r]), r))
        ranks[r].extend(u)
    return [sorted(r) for r in ranks]


def deal_for_root(pairs, world_size, dst=0, keep_together=False):
    """deal_pairs with the root's share made the LIGHTEST one (swapped with the rank that got it): the root also runs the
    serial fusion chain, so with 11 pairs over 8 ranks it takes one pair instead of two.  Deterministic, the same on
    every rank."""
    deal = deal_pairs(pairs, world_size, keep_together)
    light = min(range(world_size), key=lambda r: (len(deal[r]), -r))
    if len(deal[light]) < len(deal[dst]):
        deal[dst], deal[light] = deal[light], deal[dst]
    return deal


def gather_to_root(local, dst=0, group=None):
    """local: (n_local, ...) tensor, same n_local on every rank -> list of per-rank tensors on dst, else None.
    A single gather collective."""
    ws, rank = world(group)
    if not _collectives(group):
        return [local]
    local = local.contiguous()
    bufs = [torch.empty_like(local) for _ in range(ws)] if rank == dst else None
    dist.gather(local, gather_list=bufs, dst=dst, group=group)
    return bufs


def run_sequence_sharded(run_sequence, sequences, dst=0, group=None):
    """sequences: the GLOBAL list (every rank passes the same list; only its shard is touched).
    run_sequence(seq) -> tensor (flow of the last frame to frame 0), or a SequencePipeline: the rank's shard then
    runs software-pipelined (fusion chain of one sequence underneath the estimator of the next).  Returns the flows
    in global order on `dst`, None elsewhere.  Requires len(sequences) % world == 0 (one gather, equal message
    sizes)."""
    ws, rank = world(group)
    n = len(sequences)
    if n % ws:
        raise ValueError("sequence-sharded mode needs len(sequences) %% world_size == 0 (got %d, %d); "
                         "use run_pair_sharded for fewer sequences than ranks" % (n, ws))
    mine = block_partition(n, ws, rank)
    if isinstance(run_sequence, SequencePipeline):
        outs = [run_sequence.submit(sequences[i]) for i in mine] + [run_sequence.flush()]
        if any(o is not None and len(o) == 0 for o in outs):
            raise ValueError("run_sequence_sharded: a sequence of fewer than 3 frames has no accumulated flow")
        local = torch.stack([o[-1] for o in outs if o is not None], dim=0)
    else:
        local = torch.stack([run_sequence(sequences[i]) for i in mine], dim=0)
    parts = gather_to_root(local, dst=dst, group=group)
    if parts is None:
        return None
    return [f for part in parts for f in part.unbind(0)]


def run_pair_sharded(estimate_small, fuse_chain, n_frames, pairs, dst=0, group=None, keep_together=False):
    """One sequence over several ranks.  estimate_small(list_of_pairs) -> (len, N, 2, h, w) 1/8-res flows of
    this rank's pairs; fuse_chain(dict pair -> flow) -> outputs, run on dst only.  One all_gather.
    keep_together: see deal_pairs (GMA: pairs that share an attention matrix stay on one rank)."""
    ws, rank = world(group)
    deal = deal_for_root(pairs, ws, dst, keep_together)
    mine = deal[rank]
    per_rank = max(len(d) for d in deal)
    local = estimate_small([pairs[i] for i in mine])
    if local.shape[0] < per_rank:  # pad so every rank contributes the same message size
        pad = torch.zeros((per_rank - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    if not _collectives(group):
        gathered = [local]
    else:
        gathered = [torch.empty_like(local) for _ in range(ws)]
        dist.all_gather(gathered, local.contiguous(), group=group)
    if rank != dst:
        return None
    by_pair = {}
    for r in range(ws):
        for slot, i in enumerate(deal[r]):
            by_pair[pairs[i]] = gathered[r][slot]
    return fuse_chain(by_pair)


def rotated_deal(pairs, world_size, k, keep_together=False):
    """Deal of sequence k in a STREAM of sequences (pair-sharded mode): -> (root, deal).  The root - the rank that runs
    the serial fusion chain - is rank k % world_size, and rank r takes the share rank (r - k) % world_size holds in the
    root-0 deal (deal_for_root: the root's share is the lightest).  Over any world_size consecutive sequences every rank
    holds every share and the root role exactly once: equal pair counts and one chain per rank, where a fixed root would do
    every chain while the other ranks idle behind their pairs."""
    base = deal_for_root(pairs, world_size, 0, keep_together)
    return k % world_size, [base[(r - k) % world_size] for r in range(world_size)]


def run_pair_sharded_stream(estimate_small, fuse_chain, pairs, sequences, group=None, keep_together=False, harvest=None,
                            on_result=None):
    """A stream of sequences, each spread over all ranks (run_pair_sharded per sequence) with a ROTATING root.
    sequences: the global list (every rank passes the same list).  estimate_small(seq, list_of_pairs, is_root) ->
    (len, N, 2, h, w) 1/8-res flows of this rank's pairs of `seq`, or (flows, aux) with aux a small 1-D tensor of the
    flows' dtype that travels IN the same all_gather (the f16x3 range-guard flag of this rank's pairs: the root then
    learns of a trip on any rank without a second collective and without a host synchronisation on any rank);
    fuse_chain(seq, dict pair -> flow[, list of every rank's aux]) runs on that sequence's root only and may return a
    pending handle (a chain launched on a side stream: it then executes underneath the root's pairs of the following
    sequences); harvest(handle) -> outputs resolves it (default: identity).  ONE all_gather per sequence, no other
    collective.  Returns {sequence index: outputs} for the sequences this rank was the root of (gather them with
    gather_to_root if one rank needs all).

    Bounded lag: a pending chain keeps its sequence's images, gathered flows, flags and a pinned host buffer
    alive (~60 MB per 7-frame 480x1024 sequence), so handles are NOT kept until the end of the stream: when sequence k has
    been launched, this rank resolves the sequences it rooted up to k - lag (lag = the world size, at least 2: the chain of
    a sequence that old finished while the following estimator calls were being issued, so harvest's wait on its event
    returns at once and the loop still runs without a stall) - at most two handles per rank are alive, a tripped sequence
    is recomputed `lag` sequences later instead of at the very end, and `sequences` may be any iterable (a data loader).
    on_result(k, outputs): if given, resolved outputs are handed over there instead of being collected (the returned dict
    is then empty): memory independent of the stream's length."""
    ws, rank = world(group)
    pending, results = {}, {}
    lag = ws * ((2 + ws - 1) // ws)

    def resolve(upto):
        for j in sorted(pending):
            if j > upto:
                break
            h = pending.pop(j)
            out = harvest(h) if harvest is not None else h
            if on_result is not None:
                on_result(j, out)
            else:
                results[j] = out

    for k, seq in enumerate(sequences):
        root, deal = rotated_deal(pairs, ws, k, keep_together)
        mine = deal[rank]
        per_rank = max(len(d) for d in deal)
        local = estimate_small(seq, [pairs[i] for i in mine], rank == root)
        aux = None
        if isinstance(local, tuple):
            local, aux = local
        if local.shape[0] < per_rank:
            pad = torch.zeros((per_rank - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], dim=0)
        shape, n_flow = tuple(local.shape), local.numel()
        payload = local.reshape(-1) if aux is None else torch.cat([local.reshape(-1), aux.reshape(-1).to(local.dtype)])
        if not _collectives(group):
            gathered = [payload]
        else:
            gathered = [torch.empty_like(payload) for _ in range(ws)]
            dist.all_gather(gathered, payload.contiguous(), group=group)
        if rank == root:
            by_pair = {}
            for r in range(ws):
                flows_r = gathered[r][:n_flow].view(shape)
                for slot, i in enumerate(deal[r]):
                    by_pair[pairs[i]] = flows_r[slot]
            if aux is None:
                pending[k] = fuse_chain(seq, by_pair)
            else:
                pending[k] = fuse_chain(seq, by_pair, [g[n_flow:] for g in gathered])
        resolve(k - lag)
    resolve(float("inf"))
    return results


# SequencePipeline: encoders on the caller's stream, refinement homed on the first pair-group stream (round 6; 0: the whole
# estimator on the caller's stream; 22.13 -> 21.76 ms per step same-box, profiles/r06_ab_pipeline_split.txt)
PIPELINE_SPLIT = os.environ.get("ACCFLOW_PIPELINE_SPLIT", "1") == "1"


class SequencePipeline:
    """Software pipeline over independent sequences on ONE GPU (what a data loader loop over sequences runs,
    test_cvo.py:60-101, at depth 1): the batch-1 fusion chain of sequence k - context encoder + 5 sequential AccPlus /
    decoder steps of 60-120 workgroups per launch, which cannot fill 256 CUs - is issued on a side stream and
    executes underneath the estimator of sequence k+1 on the main stream.  Results are those of `model(images)`
    (same kernels, same order within a sequence), returned one `submit` later.

        pipe = SequencePipeline(model)
        for images in sequences:
            outs = pipe.submit(images)      # outputs of the PREVIOUS sequence (None for the first)
        outs = pipe.flush()                 # outputs of the last one

    f16x3 range guard: each sequence reports to its own device flag (ops.guard_scope), copied to pinned host memory
    behind its chain; a tripped sequence is recomputed in bf16x6 at harvest, exactly as the module entry points do.
    The host never synchronises with the main stream.  Streams in use: the caller's, the estimator's two pair-group
    streams and ONE chain stream = 4, the number of hardware queues HIP multiplexes streams onto by default
    (GPU_MAX_HW_QUEUES); a fifth concurrently active stream would share a queue and serialise (measured: 30.3 instead
    of 26.7 ms per sequence, profiles/r02_ab_pipeline.txt) - call submit() from the default stream, as the loop of
    test_cvo.py does."""

    def __init__(self, model):
        self.model = model
        self.side = None
        self.pending = None
        self.count = 0

    @torch.no_grad()
    def _launch(self, images):
        from . import ops
        m = self.model
        images = list(images)
        dev = images[0].device
        if len(images) < 3 or getattr(m, "warm_start", False):   # nothing to overlap / seeded schedule: plain forward
            return (None, None, m(images=images), None)
        if self.side is None:
            from .networks.AccFlow_ import _context_stream
            self.side = _context_stream(dev)     # (the process's one side stream per device: 4 streams = 4 hardware queues)
        main = torch.cuda.current_stream(dev)
        if hasattr(m, "stack_frames"):
            images = m.stack_frames(images)      # (one copy instead of one torch.cat per encoder)
        N = images[0].shape[0]
        pairs = m.pair_schedule(len(images))
        guarded = ops.current_mode() == ops.CONV_F16X3
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if guarded else None
        scope = ops.guard_scope(flag) if guarded else contextlib.nullcontext()
        with scope:
            home, feats = main, None
            if PIPELINE_SPLIT and hasattr(m.ofe, "encode_pairs"):
                # The estimator in two halves: the encoders on the caller's stream, the refinement "at home" on the first
                # pair-group stream (which forks the second and joins it) - the caller's stream is then never blocked by a
                # refinement, so the NEXT sequence's encoders (big, traffic-heavy launches) run underneath this one's
                # iterations and fill the tail in which only the larger pair group is still iterating.  Streams in use: still
                # 4.  `feats` lives in the caller's allocator pool and is read on the group streams: kept until the harvest.
                from .networks.raft import raft as _raft
                if _raft.N_STREAMS > 1:
                    feats = m.ofe.encode_pairs(images, pairs)
                    enc_done = torch.cuda.Event()
                    enc_done.record(main)
                    home = _raft._side_streams(dev, _raft.N_STREAMS)[0]
                    home.wait_event(enc_done)
                    # the pair groups wait for the encoders' events only, never for each other, and take the larger share of
                    # the 11 pairs in turns (RAFT._refine)
                    self.count += 1
                    feats["refine_at_home"] = (enc_done, bool(self.count & 1))
            with torch.cuda.stream(home):
                small = m.estimate_small(images, pairs, features=feats)
                ready = torch.cuda.Event()
                ready.record(home)
            by_pair = {p: small[k * N:(k + 1) * N] for k, p in enumerate(pairs)}
            self.side.wait_event(ready)
            host = None
            from .networks.AccFlow_ import chain_in_pipeline
            with torch.cuda.stream(self.side), chain_in_pipeline():
                outs = m.fuse_chain(images, by_pair)
                if guarded:
                    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                    host.copy_(flag, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.side)
            # the outputs live in blocks of the SIDE stream's allocator pool but are consumed by the caller on `main`
            # (whatever stream that is): tell the allocator, so that freeing them while the caller's kernels still read
            # them cannot hand the block to the next chain (the harvest only synchronises the HOST with the chain)
            for o in outs:
                o.record_stream(main)
        # `small`, `images`, `flag` were allocated on the main stream and are read by the side stream: they stay
        # referenced here until the chain has finished
        return (done, host, outs, (images, small, flag, feats))

    def _harvest(self, p):
        from . import ops
        done, host, outs, keep = p
        if done is None:
            return outs
        done.synchronize()
        if host is not None and int(host.item()):
            from .networks.AccFlow_ import pipeline_chain_arithmetic
            with ops.conv_mode(ops.CONV_BF16X6), pipeline_chain_arithmetic():     # (every output of the pipeline: one arithmetic)
                outs = self.model(images=keep[0])
        return outs

    def submit(self, images):
        prev, self.pending = self.pending, self._launch(images)
        return self._harvest(prev) if prev is not None else None

    def flush(self):
        prev, self.pending = self.pending, None
        return self._harvest(prev) if prev is not None else None
