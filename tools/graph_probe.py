"""Feasibility probe: capture one whole AccFlow(RAFT) sequence forward in a HIP graph (torch.cuda.graph) and replay it.
    python tools/graph_probe.py [--H 480 --W 1024 --frames 7]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=7)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from accflow_amd import ops
    from accflow_amd.data.synthetic import make_sequence, make_state_dict, normalize
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    model = AccFlow(build_flow_estimator("acc|raft"))
    model.load_state_dict(make_state_dict(model), strict=True)
    model = model.cuda().eval()
    frames = [normalize(f).cuda() for f in make_sequence(1000, a.frames, a.H, a.W)]
    frames2 = [normalize(f).cuda() for f in make_sequence(2000, a.frames, a.H, a.W)]
    with torch.no_grad():
        for _ in range(3):
            ref = model(frames)
        ref2 = model(frames2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model(frames)
        torch.cuda.synchronize()
        print("eager ms/seq %.3f" % ((time.perf_counter() - t0) * 1e3 / a.steps))
        static = [f.clone() for f in frames]
        g = torch.cuda.CUDAGraph()
        N = frames[0].shape[0]
        pairs = model.pair_schedule(len(frames))

        def body():
            handle = model.context_async(static)
            small = model.estimate_small(static, pairs)
            ctx = model.context_join(handle)
            return model.fuse_chain(static, {p: small[k * N:(k + 1) * N] for k, p in enumerate(pairs)}, ctx=ctx)
        flag = torch.zeros(1, dtype=torch.int32, device=static[0].device)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            with ops.guard_scope(flag):
                outs = body()
        torch.cuda.synchronize()
        print("captured")
        g.replay()
        torch.cuda.synchronize()
        print("replay 1 ok; max diff vs eager", max(float((o - r).abs().max()) for o, r in zip(outs, ref)), "guard", int(flag.item()))
        for s, f in zip(static, frames2):
            s.copy_(f)
        g.replay()
        torch.cuda.synchronize()
        print("replay new inputs: max diff", max(float((o - r).abs().max()) for o, r in zip(outs, ref2)))
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.replay()
        torch.cuda.synchronize()
        print("graph ms/seq %.3f" % ((time.perf_counter() - t0) * 1e3 / a.steps))
        # two instances on two streams: sequence k+1 underneath sequence k
        static_b = [f.clone() for f in frames]
        g2 = torch.cuda.CUDAGraph()
        static_a, static[:] = list(static), static_b
        flag2 = torch.zeros(1, dtype=torch.int32, device=static[0].device)
        torch.cuda.synchronize()
        with torch.cuda.graph(g2):
            with ops.guard_scope(flag2):
                outs2 = body()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for rounds in (1, 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(a.steps):
                with torch.cuda.stream(s1 if k % 2 == 0 else s2):
                    (g if k % 2 == 0 else g2).replay()
            torch.cuda.synchronize()
            print("two graphs / two streams ms/seq %.3f" % ((time.perf_counter() - t0) * 1e3 / a.steps))
        print("second instance diff", max(float((o - r).abs().max()) for o, r in zip(outs2, ref)))


if __name__ == "__main__":
    main()
