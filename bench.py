#!/usr/bin/env python3
"""Headline benchmark: AccFlow(RAFT) backward accumulation over 7-frame 480x1024 sequences on MI355X.

One "step" = one pass of the hot path over one batch of synthetic sequences per GPU (default one 7-frame
sequence = 11 estimator pair-evaluations + 5 fusion steps, BASELINE.json configs[2] - the configuration the
metric is quoted on).  Inputs and weights are resident in HBM before the timed region.  With N > 1 ranks (one
process per GPU, RCCL; started by torch.distributed.run, or by this script itself when WORLD_SIZE is not set:
`python bench.py --gpus 8` spawns its 8 ranks before making any GPU call) every rank processes its own sequences
(weak scaling) and the final accumulated flows are gathered to rank 0 with ONE gather per step, inside the timed
region.  `--shard pairs` is the strong-scaling mode: one sequence per step spread over the ranks
(AccFlow.forward_pair_sharded: pairs dealt over the ranks, one all_gather of the 1/8-res flows, chain on rank 0);
`--shard pairs-stream` the same over a stream of sequences with a rotating root (AccFlow.forward_pair_sharded_stream:
sequence k's chain on rank k mod N underneath that rank's pairs of the following sequences).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL on this pool's host driver

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (no sparsity)
PROF_STEPS = 2
STRICT_STEPS = 3
MFMAS_PER_PRODUCT = {"f32": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--seqs-per-gpu", type=int, default=1)
    ap.add_argument("--ofe", choices=["raft", "gma"], default="raft", help="pair estimator (gma + 720x1280 = configs[4])")
    ap.add_argument("--shard", choices=["sequences", "pairs", "pairs-stream"], default="sequences",
                    help="sequences (default, weak scaling): every rank runs its own sequences, one gather of the final flow per "
                         "step; pairs (strong scaling): ONE sequence per step over all ranks through AccFlow.forward_pair_sharded "
                         "(estimator pairs dealt over the ranks, one all_gather of the 1/8-res flows, fusion chain on rank 0); "
                         "pairs-stream: the same sharding over a stream of sequences with a ROTATING root "
                         "(AccFlow.forward_pair_sharded_stream: sequence k's chain on rank k %% world, underneath the next pairs)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the --gpus ranks from this process even for N = 1 (default: only when N > 1 and no WORLD_SIZE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the secondary bf16x6 measurement")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one sequence at a time (default: parallel.SequencePipeline - the fusion chain of step k runs "
                         "underneath the estimator of step k+1; all K steps complete inside the timed region)")
    ap.add_argument("--no-extra", action="store_true", help="skip the C2 (single pair), C5 (GMA 720x1280) and training-step side measurements")
    ap.add_argument("--busy-json", default=os.path.join(ROOT, "profiles", "conv_mfma_busy.json"),
                    help="matrix-pipe counters of the conv kernels from a rocprofv3 --pmc pass (optional)")
    ap.add_argument("--dump-kernels", default=None, help="write the per-conv-shape timing table to this file")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "lookup_traffic.json"),
                    help="per-launch HBM bytes of the lookup kernel from a rocprofv3 --pmc pass (optional)")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may really use: affinity mask, clipped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(iters, height, width, frames, budget_s=240):
    """The oracle (CPU restatement of the reference's PyTorch path) on the host cores, bounded sample of the same
    workload: after one warm-up pair-eval, 3 timed single pair-evals (pair 2->1 of sequence 1000: BASELINE configs[1])
    and ONE whole sequence (11 pair-evals + the fusion chain: configs[2], the bench's unit of work) - about 30 s of CPU
    work; `value` is the whole-sequence rate.  Runs in a child process (which never touches the GPU) so that a hard
    time budget can be enforced."""
    import subprocess
    cores = min(usable_cpus(), 32)  # torch CPU conv/gather kernels stop scaling (and thrash) far beyond this
    code = (
        "import sys, time, json, torch; sys.path.insert(0, %r)\n"
        "from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize\n"
        "from accflow_amd.networks import build_flow_estimator\n"
        "from accflow_amd.networks.AccFlow_ import AccFlow\n"
        "from oracle import accflow_oracle as O\n"
        "torch.set_num_threads(%d)\n"
        "model = AccFlow(build_flow_estimator('acc|raft'))\n"
        "sd = make_state_dict(model)\n"
        "ofe = {k[4:]: v for k, v in sd.items() if k.startswith('ofe.')}\n"
        "fr = [normalize(f) for f in make_sequence(1000, %d, %d, %d)]\n"
        "res = {'threads': torch.get_num_threads(), 'pair': [], 'seq': None}\n"
        "with torch.no_grad():\n"
        "    O.raft_forward(ofe, fr[2], fr[1], iters=%d)\n"
        "    for _ in range(3):\n"
        "        t0 = time.perf_counter(); O.raft_forward(ofe, fr[2], fr[1], iters=%d); res['pair'].append(time.perf_counter() - t0)\n"
        "    print(json.dumps(res), flush=True)\n"
        "    t0 = time.perf_counter(); O.accflow_forward(sd, fr, iters=%d); res['seq'] = time.perf_counter() - t0\n"
        "print(json.dumps(res), flush=True)\n" % (ROOT, cores, frames, height, width, iters, iters, iters))
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES="")
    out = ""
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s, env=env)
        out = r.stdout
    except subprocess.TimeoutExpired as e:  # keep what was printed before the budget ran out
        out = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    except Exception:
        pass
    d = None
    for line in out.strip().splitlines():
        try:
            d = json.loads(line)
        except Exception:
            pass
    if d is None:
        return {"value": None, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                "sample": "the oracle did not finish one pair-eval within %d s" % budget_s}
    pair = sorted(d["pair"])[len(d["pair"]) // 2]
    n_pairs = 3 + 2 * (frames - 3)
    what = ("oracle/accflow_oracle.py on torch CPU fp32: median of 3 single pair-evals (pair 2->1, batch-1 RAFT, %d iters, "
            "%dx%d) after 1 warm-up = %.2f s (%.3f frame-pairs/s)" % (iters, height, width, pair, 1.0 / pair))
    if d.get("seq"):
        return {"value": round(n_pairs / d["seq"], 4), "unit": "frame-pairs/s", "cores": d["threads"], "kind": "port",
                "single_pair_value": round(1.0 / pair, 4),
                "sample": "1 whole %d-frame sequence (%d pair-evals + %d fusion steps) in %.1f s; %s"
                          % (frames, n_pairs, frames - 2, d["seq"], what)}
    return {"value": round(1.0 / pair, 4), "unit": "frame-pairs/s", "cores": d["threads"], "kind": "port",
            "sample": what + "; the whole-sequence run did not finish within the %d s budget" % budget_s}


def extra_configs(a, dev):
    """Driver-visible side measurements of the other single-GPU configurations of BASELINE.json (not the headline):
    configs[1] = one RAFT pair at 480x1024, 12 iterations; configs[4] = AccFlow(GMA) 7 x 720x1280 on one GPU."""
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    out = {}

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    m = build_flow_estimator("raft")
    m.load_state_dict(make_state_dict(m), strict=True)
    m = m.to(dev).eval()
    fr = [normalize(f).to(dev) for f in make_sequence(1000, 2, 480, 1024)]
    t = timed(lambda: m(fr[1], fr[0], iters=12), 5)
    out["c2_raft_pair_480x1024_12it"] = {"ms_per_pair": round(1e3 * t, 3), "frame_pairs_per_s": round(1.0 / t, 2), "runs": 5,
                                         "config": "BASELINE.json configs[1]: RAFT direct, one 480x1024 pair, 12 GRU iters, batch 1"}
    del m, fr
    torch.cuda.empty_cache()
    g = AccFlow(build_flow_estimator("acc|gma"))
    g.load_state_dict(make_state_dict(g), strict=True)
    g = g.to(dev).eval()
    fr = [normalize(f).to(dev) for f in make_sequence(1000, 7, 720, 1280)]
    t = timed(lambda: g(images=fr), 3)
    out["c5_accflow_gma_7x720x1280"] = {"ms_per_step": round(1e3 * t, 3), "frame_pairs_per_s": round(11.0 / t, 2), "runs": 3,
                                        "config": "BASELINE.json configs[4] on ONE GPU: AccFlow(GMA) 7-frame 720x1280, 12 GRU iters "
                                                  "(11 pair-evals per step), one sequence at a time"}
    if not a.no_pipeline:
        from accflow_amd.parallel import SequencePipeline
        pipe = SequencePipeline(g)

        def piped():
            for _ in range(3):
                pipe.submit(fr)
            pipe.flush()
        t = timed(piped, 1) / 3
        out["c5_accflow_gma_7x720x1280"].update({"pipelined_ms_per_step": round(1e3 * t, 3),
                                                 "pipelined_frame_pairs_per_s": round(11.0 / t, 2)})
    del g, fr
    torch.cuda.empty_cache()
    # the training slice: one optimizer step at the reference's training shape (configs/AccRAFT-CVO.yml:
    # 7 frames of 256 x 256, batch 6 per GPU, AdamW, clip 1.0) - forward + hand-written backward of the fusion heads
    try:
        from accflow_amd import train
        tm = AccFlow(build_flow_estimator("acc|raft"))
        tm.load_state_dict(make_state_dict(tm), strict=True)
        tm = tm.to(dev).eval()
        fr = [normalize(f).to(dev) for f in make_sequence(11, 7, 256, 256, batch=6)]
        gen = torch.Generator().manual_seed(3)
        gts = [(3.0 * torch.randn(6, 2, 256, 256, generator=gen)).to(dev) for _ in range(5)]
        opt = torch.optim.AdamW(train.trainable_parameters(tm), lr=1.2e-4, weight_decay=1e-5, eps=1e-8)
        losses = []
        t_eager = timed(lambda: losses.append(train.train_step(tm, opt, fr, gts)[0]), 3)
        # the front end's default (accflow_amd/train_acc.py): forward + backward replayed from a HIP graph for the loader's
        # fixed shapes; optimizer / clipping stay eager
        gfb = train.GraphedForwardBackward(tm, fr, gts)
        t = timed(lambda: losses.append(train.train_step(tm, opt, fr, gts, graphed=gfb)[0]), 3)
        # roofline of the dominant backward kernel: per-launch HIP events around every weight-gradient GEMM of one eager step
        # (conv_wgrad_mfma_fast_kernel<3> / conv_wgrad_mfma_kernel<3>: 6 bf16 MFMAs per product -> ceiling = dense 16-bit MFMA peak / 6)
        from accflow_amd import profiler as _prof
        wt = _prof.KernelTimer(["conv_wgrad"])
        _prof.ACTIVE = wt
        try:
            train.train_step(tm, opt, fr, gts)
            torch.cuda.synchronize()
        finally:
            _prof.ACTIVE = None
        wg = wt.summary().get("conv_wgrad")
        out["train_step_accraft_7x256x256_b6"] = {"ms_per_step": round(1e3 * t, 3), "sequences_per_s": round(6.0 / t, 2), "runs": 3,
                                                  "ms_per_step_eager": round(1e3 * t_eager, 3),
                                                  "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
                                                  "forward_conv_mode": train.TRAIN_CONV_MODE,
                                                  "config": "train_acc.py step at configs/AccRAFT-CVO.yml's shape (frozen estimator, "
                                                            "heads' forward on the guarded fp16 split, backward bf16x6, AdamW; "
                                                            "forward + backward replayed from a HIP graph as the front end does by "
                                                            "default): DESIGN.md section 6b"}
        if wg:
            tf = wg["work"] / (wg["total_ms"] * 1e-3) / 1e12
            out["train_step_accraft_7x256x256_b6"]["roofline_wgrad"] = {
                "kernel": "conv_wgrad_mfma_fast_kernel<3> (stride-1 'same' shapes; conv_wgrad_mfma_kernel<3> for the rest)", "bound": "mfma",
                "achieved": round(tf, 1),
                "peak": round(BF16_MFMA_PEAK_TF / 6.0, 1), "unit": "TFLOP/s", "frac": round(tf / (BF16_MFMA_PEAK_TF / 6.0), 4),
                "launches_per_step": wg["launches"], "ms_per_step_in_kernel": round(wg["total_ms"], 3),
                "note": "algorithmic weight-gradient flop (2 Cout Cin KH KW per output pixel) / HIP-event time of every launch "
                        "of one eager training step; 6 bf16 MFMAs per product (3-term operands, fp32-equivalent)"}
        del gfb
        del tm, fr, gts, opt
        torch.cuda.empty_cache()
    except Exception as e:   # the side measurement must never take the headline down with it
        out["train_step_accraft_7x256x256_b6"] = {"error": repr(e)}
    return out


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.spawn):
        # `python bench.py --gpus N` without torchrun: this process has made no GPU call (importing torch does not
        # initialise HIP); it starts the N ranks as child processes, relays rank 0's JSON line and exits with their code
        from accflow_amd.launch import spawn_ranks
        argv = [os.path.abspath(__file__)] + [x for x in sys.argv[1:] if x != "--spawn"]
        raise SystemExit(spawn_ranks(argv, a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (a.gpus, world))
    if torch.cuda.device_count() <= local_rank:   # (device_count does not initialise the GPU)
        raise SystemExit("bench.py: rank %d needs GPU %d but this node exposes %d GPU(s) (no CPU path in the product)"
                         % (rank, local_rank, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in the product)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pairs_mode = a.shard in ("pairs", "pairs-stream")
    stream_mode = a.shard == "pairs-stream"
    rccl_ranks = 1
    if world > 1 or pairs_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            from accflow_amd.launch import free_port
            os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)              # RCCL really spans `world` ranks: every rank contributed a 1
        rccl_ranks = int(ones.item())
        try:   # RCCL writes its version banner to the C stdout: push it out now, so that the JSON line is the LAST line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if rccl_ranks != world:
            raise SystemExit("bench.py: all_reduce of ones = %d, expected %d" % (rccl_ranks, world))

    from accflow_amd import profiler
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    from accflow_amd.parallel import gather_to_root

    model = AccFlow(build_flow_estimator("acc|" + a.ofe))
    sd = make_state_dict(model)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.ofe_iters = a.iters
    grouped = dist.is_initialized()
    S = 1 if pairs_mode else a.seqs_per_gpu
    # sequences mode: every rank has its OWN sequences (seeds 1000 + rank*S ...); pairs mode: all ranks hold the SAME one
    frames_cpu = [normalize(f) for f in make_sequence(1000 + (0 if pairs_mode else rank * S), a.frames, a.height, a.width, batch=S)]
    frames = [f.to(dev) for f in frames_cpu]
    pairs_per_seq = len(model.pair_schedule(a.frames))

    from accflow_amd.parallel import SequencePipeline
    pipe = None if (a.no_pipeline or pairs_mode) else SequencePipeline(model)

    def step_local():
        if pairs_mode:   # one sequence over all ranks; outputs on rank 0, None elsewhere
            return model.forward_pair_sharded(frames, dst=0)
        return model(images=frames)

    def run_steps(n):
        """n steps (= n sequences per rank); every step's final flow goes to the root with one gather.  With the
        pipeline a step's outputs are harvested while the next step's estimator is already queued; the last step is
        flushed before returning, so all n steps are complete when the caller's fence returns.
        pairs mode: n sequences in all, each spread over the ranks (its one collective is the all_gather inside
        forward_pair_sharded)."""
        last = None
        if stream_mode and n > 0:   # n sequences, root of sequence k = rank k % world; every rank returns what it rooted
            got = model.forward_pair_sharded_stream([frames] * n)
            return got[max(got)] if got else None
        for k in range(n + (1 if pipe else 0)):
            if pipe:
                outs = pipe.submit(frames) if k < n else pipe.flush()
            else:
                outs = step_local()
            if outs is not None:
                last = outs
                if grouped and not pairs_mode:
                    gather_to_root(outs[-1], dst=0)
        return last

    def step():
        return run_steps(1)

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(a.warmup)
    fence()
    t0 = time.perf_counter()
    outs = run_steps(a.steps)
    fence()
    elapsed = time.perf_counter() - t0
    # Per-kernel HIP-event timing for the roofline objects: PROF_STEPS extra steps of the same workload right
    # after the timed region, with the pair groups on ONE stream - in the timed region two streams overlap
    # kernels, which makes a single kernel's event-to-event time meaningless.
    timer = profiler.KernelTimer(["corr_lookup", "conv2d", "lookup_convc1"]) if rank == 0 else None
    if rank == 0:
        import accflow_amd.networks.raft.raft as _raft
        saved = _raft.N_STREAMS
        _raft.N_STREAMS = 1
        # the product path runs the lookup FUSED with convc1 (csrc/corr_lookup_conv.hip); in this pass the north-star kernel
        # is launched alone as well - same pyramid, same coordinates - so that `roofline_lookup` stays a measurement of it
        _raft.PROFILE_STANDALONE_LOOKUP = True
        from accflow_amd.networks import AccFlow_ as _acc
        saved_ctx, _acc.CONTEXT_SIDE_STREAM = _acc.CONTEXT_SIDE_STREAM, False   # (per-launch events: ONE stream in this pass)
        profiler.ACTIVE = timer
        for _ in range(PROF_STEPS):
            model(images=frames)           # (rank-local: no collective, whatever the sharding mode)
        torch.cuda.synchronize()
        profiler.ACTIVE = None
        _raft.PROFILE_STANDALONE_LOOKUP = False
        _raft.N_STREAMS = saved
        _acc.CONTEXT_SIDE_STREAM = saved_ctx
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Secondary figure: latency-style schedule, one sequence at a time (model(images) in a loop)
    unpiped = None
    if pipe is not None:
        step_local()
        fence()
        ts = time.perf_counter()
        for _ in range(STRICT_STEPS):
            o1 = step_local()
            if grouped:
                gather_to_root(o1[-1], dst=0)
        fence()
        unpiped = (time.perf_counter() - ts) / STRICT_STEPS

    # Secondary figure (rank 0 only, no collectives): TWO sequences per step (22 pairs per estimator launch) - what batching
    # independent sequences, as test_cvo.py's loader does (batch 10), buys on this workload
    batched = None
    if rank == 0 and pipe is not None and S == 1 and not a.no_extra:
        two = [torch.cat([f, f2]) for f, f2 in zip(frames, [normalize(f).to(dev) for f in
                                                          make_sequence(5000, a.frames, a.height, a.width)])]
        pb = SequencePipeline(model)
        pb.submit(two)
        pb.flush()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(STRICT_STEPS):
            pb.submit(two)
        pb.flush()
        torch.cuda.synchronize()
        tb = (time.perf_counter() - ts) / STRICT_STEPS
        batched = {"sequences_per_step": 2, "ms_per_sequence": round(1e3 * tb / 2, 3), "steps": STRICT_STEPS,
                   "value_one_gpu": round(2 * pairs_per_seq / tb, 3)}
        del two, pb
    # Secondary figure (N = 1, sequences mode): the strong-scaling code path - AccFlow.forward_pair_sharded - on ONE rank
    pair_one = None
    if rank == 0 and world == 1 and not pairs_mode and not a.no_extra:
        model.forward_pair_sharded(frames)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(STRICT_STEPS):
            model.forward_pair_sharded(frames)
        torch.cuda.synchronize()
        pair_one = {"ms_per_step": round(1e3 * (time.perf_counter() - ts) / STRICT_STEPS, 3), "steps": STRICT_STEPS,
                    "note": "AccFlow.forward_pair_sharded with every pair on this rank (bench.py --shard pairs times it over N ranks)"}
        model.forward_pair_sharded_stream([frames] * 2)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        n_stream = max(STRICT_STEPS, 10)     # (the last sequence's chain has nothing to hide under: amortised over n)
        model.forward_pair_sharded_stream([frames] * n_stream)
        torch.cuda.synchronize()
        pair_one["stream_ms_per_step"] = round(1e3 * (time.perf_counter() - ts) / n_stream, 3)
        pair_one["stream_steps"] = n_stream
        pair_one["stream_note"] = ("AccFlow.forward_pair_sharded_stream (rotating root; on one rank: every chain on the side "
                                   "stream under the next sequence's pairs) - bench.py --shard pairs-stream over N ranks")
    if grouped:
        dist.barrier()

    # Secondary figure: the same workload with the unconditional fp32-equivalent arithmetic (bf16x6) - reported beside
    # the headline so that the cost of NOT using the fp16 operand split is on the same JSON line.
    strict = None
    from accflow_amd import ops as _ops
    if _ops.conv_mode_name() == "f16x3" and not a.no_strict:
        _ops.set_conv_mode("bf16x6")
        step()
        fence()
        ts = time.perf_counter()
        outs_strict = run_steps(STRICT_STEPS)
        fence()
        el = time.perf_counter() - ts
        _ops.set_conv_mode("f16x3")
        if grouped:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        strict = {"conv_mode": "bf16x6", "value": round((1 if pairs_mode else world) * S * pairs_per_seq * STRICT_STEPS / el, 3),
                  "ms_per_step": round(1e3 * el / STRICT_STEPS, 3), "steps": STRICT_STEPS}
        if rank == 0 and not a.no_parity:
            strict["parity"] = parity_vs_golden(outs_strict, a)

    if rank == 0:
        nseq = (1 if pairs_mode else world) * S * a.steps   # sequences completed inside the timed region, all ranks
        pair_evals = nseq * pairs_per_seq
        value = pair_evals / elapsed
        seq_s = nseq / elapsed
        ks = timer.summary()
        lk, cv, lf = ks.get("corr_lookup"), ks.get("conv2d"), ks.get("lookup_convc1")
        traffic = traffic_fused = None
        if os.path.exists(a.traffic_json):
            try:
                tj = json.load(open(a.traffic_json))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_fused = tj.get("fused", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "frame-pairs/s (estimator pair-evals/s), AccFlow(%s) %d-frame %dx%d backward accumulation"
                      % (a.ofe.upper(), a.frames, a.height, a.width),
            "value": round(value, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True,
            "scaling": "strong" if pairs_mode else "weak", "rccl_ranks": rccl_ranks,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("AccFlow(%s) %d-frame %dx%d, %d GRU iters, ONE sequence per step spread over all ranks "
                                    "(BASELINE.json configs[4] / the north-star's pair sharding when n_gpus=8)"
                                    % (a.ofe.upper(), a.frames, a.height, a.width, a.iters)) if pairs_mode else
                                   ("AccFlow(%s) %d-frame %dx%d, %d GRU iters, %d sequence(s)/GPU/step "
                                    "(BASELINE.json configs[2]; configs[3] when n_gpus=8)"
                                    % (a.ofe.upper(), a.frames, a.height, a.width, a.iters, S)),
                       "pair_evals_per_sequence": pairs_per_seq, "sequences_per_s": round(seq_s, 4),
                       "adjacent_pairs_per_s": round(seq_s * (a.frames - 1), 4),
                       "parallelism": ("pair-sharded, %d rank(s): the %d estimator pairs dealt over the ranks, 1 RCCL all_gather of "
                                       "the 1/8-res flows per step, fusion chain on %s" % (world, pairs_per_seq,
                                        "rank k % world for sequence k (side stream)" if stream_mode else "rank 0")) if pairs_mode
                                      else ("sequence-sharded, %d rank(s), 1 RCCL gather of the final flow per step" % world if grouped
                                            else "1 rank, no process group: no collective runs"),
                       "schedule": ("a stream of sequences, chains on side streams of their rotating roots" if stream_mode else
                                    "one sequence at a time" if (a.no_pipeline or pairs_mode) else
                                    "SequencePipeline depth 1: the batch-1 fusion chain of step k (no split-K there) on a side stream "
                                    "underneath the estimator of step k+1, whose encoders run on the caller's stream and whose refinement is "
                                    "homed on the first pair-group stream (so step k+2's encoders run underneath step k+1's iterations); the "
                                    "last step is flushed inside the timed region"),
                       "weights": "deterministic random init (no checkpoints offline)"},
        }
        if batched is not None:
            res["two_sequences_per_step"] = batched
        if pair_one is not None:
            res["pair_sharded_one_rank"] = pair_one
        if unpiped is not None:
            res["one_sequence_at_a_time"] = {"ms_per_step": round(1e3 * unpiped, 3), "steps": STRICT_STEPS,
                                             "value": round(world * S * pairs_per_seq / unpiped, 3),
                                             "note": "rank 0's clock, model(images) in a loop without the pipeline"}
        if cv:
            from accflow_amd import ops as _ops
            mode = _ops.conv_mode_name()
            tf = cv["work"] / (cv["total_ms"] * 1e-3) / 1e12
            tf_exec = cv["work_exec"] / (cv["total_ms"] * 1e-3) / 1e12
            # algorithmic fp32 conv flop; in the split-bf16 modes every product costs 3 / 6 bf16 MFMA flops, so
            # the ceiling for ALGORITHMIC flop/s is the dense bf16 MFMA peak divided by that factor
            peak = FP32_MFMA_PEAK_TF if mode == "f32" else BF16_MFMA_PEAK_TF / MFMAS_PER_PRODUCT[mode]
            res["dtype"] = ("f32" if mode == "f32" else
                            "f32 (direct-kernel operands split into 2 fp16 terms - activations kept pre-split in HBM as fp16 hi+lo, "
                            "4 B per element; other kernels 3 bf16 terms; f32 accumulate)"
                            if mode == "f16x3" else "f32 (operands split into bf16 terms, %s; f32 accumulate)" % mode)
            res["roofline"] = {"kernel": "implicit-GEMM conv kernels (%s, all instantiations)"
                                         % ("conv2d_f32_kernel" if mode == "f32" else
                                            "conv2d_direct_bf16s_kernel + conv2d_bf16s_kernel"), "conv_mode": mode,
                               "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(tf / peak, 4), "traffic": None,
                               # the same with the flop the launches really execute (the context third of the GRU gate convs is
                               # hoisted out of the iterations; convc1 runs over 352 re-indexed channels, convf1 over 16 x 7)
                               "achieved_executed": round(tf_exec, 2), "frac_executed": round(tf_exec / peak, 4),
                               # the headline's K steps run pipelined on 4 streams; THIS object is measured afterwards on ONE
                               # stream, one sequence at a time, so that a launch's event-to-event time is its own: its
                               # ms_per_step_in_kernel may exceed the headline's ms_per_step (kernels overlap there)
                               "schedule": "single stream, one sequence at a time, %d steps after the timed region "
                                           "(the headline overlaps 4 streams)" % PROF_STEPS,
                               "mfma_flops_executed_TFLOPs": round(tf * MFMAS_PER_PRODUCT[mode], 1),
                               "frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TF, 4),
                               "launches_per_step": cv["launches"] // PROF_STEPS,
                               "avg_launch_us": round(cv["avg_us"], 2),
                               "ms_per_step_in_kernel": round(cv["total_ms"] / PROF_STEPS, 3),
                               "measured": "HIP events around every launch, %d single-stream steps after the timed region"
                                           % PROF_STEPS,
                               "flop_accounting": "algorithmic = the reference's convolutions (SURVEY 8(d)): the GRU gate "
                                                  "convs count their full 384 input channels per iteration although the "
                                                  "context third is convolved once per pair and added in the epilogue"}
            if os.path.exists(a.busy_json):
                try:  # rocprofv3 --pmc pass of the same bench command (tools/collect_profiles.sh), dominant instantiation
                    bj = json.load(open(a.busy_json))["kernels"]
                    dom = max(bj.values(), key=lambda e: e["total_ms"])
                    res["roofline"]["pmc"] = {"file": os.path.relpath(a.busy_json, ROOT),
                                              "kernel": max(bj, key=lambda k: bj[k]["total_ms"]),
                                              "mfma_pipe_busy_frac": dom["mfma_pipe_busy_frac"],
                                              "effective_clock_GHz_under_profiler": dom["effective_clock_GHz"],
                                              "mfma_TFLOPs_executed": dom["mfma_TFLOPs_executed"],
                                              "note": "rocprofv3 --pmc pass of this command with one stream (counter "
                                                      "collection serialises launches: the clock reads higher than the "
                                                      "1.4-1.9 GHz in-kernel timestamps show in the loop, "
                                                      "profiles/r02_kprof_clock.txt)"}
                except Exception:
                    pass
        if lk:
            gbs = lk["work"] / (lk["total_ms"] * 1e-3) / 1e9
            from accflow_amd.networks.raft import corr as _corr
            lk_name = "corr_lookup_disp_kernel" if _corr.LAYOUT == "disp" else "corr_lookup_kernel"
            res["roofline_lookup"] = {"kernel": lk_name, "bound": "hbm", "achieved": round(gbs, 1),
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                      "traffic": traffic, "bytes_per_launch": int(lk["work"] / lk["launches"]),
                                      "avg_launch_us": round(lk["avg_us"], 2),
                                      "launches_per_step": lk["launches"] // PROF_STEPS,
                                      "schedule": "single stream, %d steps after the timed region" % PROF_STEPS,
                                      "in_product_path": (lf is None),
                                      "note": ("" if lf is None else "the timed region runs this lookup FUSED with convc1 (roofline_lookup_fused); "
                                               "this object times the stand-alone kernel, launched additionally in the roofline pass "
                                               "on the same pyramid and coordinates. ") +
                                              "bytes_per_launch = SURVEY 8(d)'s 2 904 B per query pixel and iteration; the S16 "
                                              "lookup writes 4 x 88 pre-split channels (1 408 B) instead of 324 fp32 (1 296 B), "
                                              "i.e. 3 016 B really move; the fraction depends on the flow's coherence "
                                              "(profiles/r06_lookup_sweep.txt: 0.76 at zero flow, 0.57 / 0.48 / 0.30 at sigma = 0.5 / 1 / "
                                              "2 px of i.i.d. noise; 0.36 at sigma = 1 px on top of a smooth 4-px field)"}
        if lf:
            # the fused kernel: the lookup's reads (4 x 100 fp32 + 8 B of coordinates per query pixel) + relu(convc1) written
            # pre-split (256 channels x 4 B) = 2 632 B per pixel, and convc1's 2 * 324 * 256 flop per pixel on the matrix cores
            nl = lf["launches"]
            px = lf["work"] / (2.0 * 324 * 256) / nl
            fb = (4 * 100 * 4 + 8 + 256 * 4) * px
            gbs_f = fb * nl / (lf["total_ms"] * 1e-3) / 1e9
            res["roofline_lookup_fused"] = {
                "kernel": "corr_lookup_convc1_ws_kernel (CorrBlock lookup + convc1 + ReLU, taps through LDS into the MFMA B operand)",
                "bound": "hbm", "achieved": round(gbs_f, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs_f / HBM_PEAK_GBS, 4), "traffic": traffic_fused, "bytes_per_launch": int(fb),
                "avg_launch_us": round(lf["avg_us"], 2), "launches_per_step": nl // PROF_STEPS,
                "conv_TFLOPs_algorithmic": round(lf["work"] / (lf["total_ms"] * 1e-3) / 1e12, 1),
                "replaces_us": (round(lk["avg_us"], 2) if lk else None),
                "note": "algorithmic bytes of the FUSED op: the 1 408 B per pixel the two-launch form writes and re-reads never "
                        "reach HBM and are not counted; replaces_us = the stand-alone lookup alone (convc1 on the direct kernel "
                        "took another ~72 us per launch, profiles/r05_lc1_ablation.txt).  Over the flow's coherence "
                        "(profiles/r06_lookup_sweep.txt, tools/lookup_sweep.py --fused): 0.40 at zero flow, 0.36 / 0.31 / 0.23 at sigma = "
                        "0.5 / 1 / 2 px of i.i.d. noise, 0.26 at sigma = 1 px on top of a smooth 4-px field; against the two launches it "
                        "replaces at sigma = 1 px: 107 vs 149 us (profiles/r06_lc1_bench.txt)"}
        if cv and unpiped is not None:
            # the roofline objects are measured on ONE stream after the timed region, so the headline's
            # own overlap is reported here: how the timed (pipelined, 4-stream) step relates to the same work unpipelined and
            # to the convolution kernels' single-stream sum
            res["schedule_efficiency"] = {
                "timed_ms_per_step": round(1e3 * elapsed / a.steps, 3),
                "one_sequence_at_a_time_ms": round(1e3 * unpiped, 3),
                "pipeline_gain": round(unpiped / (elapsed / a.steps), 4),
                "conv_kernels_single_stream_ms": round(cv["total_ms"] / PROF_STEPS, 3),
                "conv_share_of_timed_step": round(cv["total_ms"] / PROF_STEPS / (1e3 * elapsed / a.steps), 4),
                "note": "conv_share > 1 means the timed step overlaps kernels the single-stream pass runs back to back; the "
                        "remaining ~15 % of a step's kernel time (correlation GEMM, fused lookup, norm / sampling passes) is in "
                        "profiles/r06_kernel_stats_bench_1stream.txt"}
        if a.dump_kernels:
            os.makedirs(os.path.dirname(a.dump_kernels) or ".", exist_ok=True)
            rows = sorted(list(timer.by_detail("conv2d").items()) + list(timer.by_detail("lookup_convc1").items()),
                          key=lambda kv: -kv[1]["total_ms"])
            with open(a.dump_kernels, "w") as f:
                for k, d in rows:
                    f.write("%-44s launches/step %4d  ms/step %8.3f  TFLOP/s %7.2f\n" % (
                        k, d["launches"] // PROF_STEPS, d["total_ms"] / PROF_STEPS,
                        d["work"] / (d["total_ms"] * 1e-3) / 1e12))
        if not a.no_parity:
            res["parity"] = parity_vs_golden(outs, a)
        if strict is not None:
            res["strict_fp32_equivalent"] = strict
        if not a.no_cpu_baseline and world == 1 and not pairs_mode:  # reported at N = 1 only (the other ranks would idle at the barrier)
            res["cpu_baseline"] = cpu_baseline(a.iters, a.height, a.width, a.frames)
        if not a.no_extra and world == 1 and not pairs_mode and (a.ofe, a.height, a.width) == ("raft", 480, 1024):
            del outs
            torch.cuda.empty_cache()
            try:
                res["other_configs"] = extra_configs(a, dev)
            except Exception as e:  # never lose the headline line to a side measurement
                res["other_configs"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def parity_vs_golden(outs, a):
    """EPE of rank 0's 5 accumulated flows against the reference's own outputs on the same inputs
    (tests/golden/accflow_c3.npz - configs[2]; accflow_gma_c5.npz for --ofe gma at 720x1280 - configs[4]; every 8th pixel),
    mean over the outputs / max."""
    import numpy as np
    name = {("raft", 7, 480, 1024, 12): "accflow_c3.npz", ("gma", 7, 720, 1280, 12): "accflow_gma_c5.npz"}.get(
        (a.ofe, a.frames, a.height, a.width, a.iters))
    path = os.path.join(ROOT, "tests", "golden", name or "")
    if name is None or not os.path.exists(path):
        return None
    g = np.load(path)
    means, mx = [], 0.0
    for k, o in enumerate(outs):
        d = (o[:1, :, ::8, ::8].cpu() - torch.from_numpy(g["out%d" % k])).pow(2).sum(1).sqrt()
        means.append(float(d.mean()))
        mx = max(mx, float(d.max()))
    return {"reference": "tests/golden/%s (reference CPU fp32 path)" % name, "epe_mean_px": round(max(means), 6),
            "epe_max_px": round(mx, 6), "gate_px": 1e-3}


if __name__ == "__main__":
    main()
