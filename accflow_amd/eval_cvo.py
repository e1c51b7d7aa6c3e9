#!/usr/bin/env python3
"""Evaluation harness with the reference's test_cvo.py semantics, on the HIP path.

Reproduces `preprocess` (test_cvo.py:32-50), `calc_occ_mask` (:53-78) and `cal_epe` (:81-101) and the CLI
(-d / -acc / -ofe / --acc_ckpt / --ofe_ckpt, :106-112).  Differences: no nn.DataParallel - with WORLD_SIZE > 1
(torchrun) batches are sharded over ranks and the per-sample EPEs gathered once at the end; checkpoints are
optional (without them the deterministic synthetic weights are used, so only throughput / parity are meaningful).
The reference's test_cvo.py itself also runs unmodified against this repo (same import surface).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from accflow_amd import ops  # noqa: E402


def preprocess(batch, device):
    out = {}
    for key, value in batch.items():
        value = value.to(device)
        if "flow" in key:
            value = value.split(2, dim=1)
            assert len(value) in [5, 6], len(value)
        elif "imgs" in key:
            value = (2 * (value / 255.0) - 1).split(3, dim=1)
            assert len(value) == 7, len(value)
        else:
            raise ValueError()
        out[key] = [v.contiguous() for v in value]
    return out


def _norm(x):
    return torch.pow(torch.sum(x ** 2, dim=1, keepdim=True), 0.5)


def calc_occ_mask(bflow, fflow):
    """FN0 and F0N in (N,2,H,W) -> (occ_bw, occ_fw), 1 = occluded (test_cvo.py:53-78; note the reference's
    `length_sq` returns the L2 norm, not its square)."""
    mag = _norm(fflow) + _norm(bflow)
    diff_fw = fflow + ops.backwarp(bflow.contiguous(), fflow.contiguous())
    diff_bw = bflow + ops.backwarp(fflow.contiguous(), bflow.contiguous())
    thr = 0.01 * mag + 0.5
    return (_norm(diff_bw) > thr).float(), (_norm(diff_fw) > thr).float()


def cal_epe(pred, label, occ_mask):
    diff = torch.norm(pred - label, p=2, dim=1, keepdim=True)
    epe_all = torch.mean(diff, dim=(1, 2, 3))
    epe_occ = torch.sum(diff * occ_mask, dim=(1, 2, 3)) / torch.sum(occ_mask, dim=(1, 2, 3))
    epe_vis = torch.sum(diff * (1 - occ_mask), dim=(1, 2, 3)) / torch.sum(1 - occ_mask, dim=(1, 2, 3))
    return epe_all, epe_occ, epe_vis


def _strip_module(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def build_model(acc, ofe, acc_ckpt, ofe_ckpt, device):
    from accflow_amd.data.synthetic import make_state_dict
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    est = build_flow_estimator(acc + "|" + ofe)
    model = AccFlow(est) if acc == "acc" else est
    ckpt = acc_ckpt if acc == "acc" else ofe_ckpt
    sd = _strip_module(torch.load(ckpt, map_location="cpu")) if ckpt else make_state_dict(model)
    model.load_state_dict(sd, strict=True)
    return model.to(device).eval()


def gather_metrics(alls, occs, viss, dev, world, rank, group=None):
    """Per-sample metrics of every rank on rank 0 as a (3, n_total) tensor.  The ranks may hold DIFFERENT numbers of samples
    (the last batch of a split rarely divides by the world size; a rank may hold none): the counts are all_reduced first, so
    that every rank leaves together when nothing was evaluated anywhere - no rank exits while the others sit in a
    collective - and the per-rank blocks are padded to the largest count for ONE gather of equal-sized messages."""
    import torch.distributed as dist
    n = sum(int(t.numel()) for t in alls)
    res = (torch.stack([torch.cat(alls), torch.cat(occs), torch.cat(viss)]) if n
           else torch.zeros((3, 0), dtype=torch.float32, device=dev))
    if world <= 1:
        if not n:
            raise SystemExit("eval_cvo: no sample evaluated (empty split or sequences shorter than 3 frames)")
        return res
    counts = torch.zeros(world, dtype=torch.int64, device=res.device)
    counts[rank] = n
    dist.all_reduce(counts, group=group)
    counts = [int(c) for c in counts.tolist()]
    if not sum(counts):
        raise SystemExit("eval_cvo: no sample evaluated on any rank (empty split or sequences shorter than 3 frames)")
    pad = torch.zeros((3, max(counts)), dtype=res.dtype, device=res.device)
    pad[:, :n] = res
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, parts, dst=0, group=group)
    if rank != 0:
        return res
    return torch.cat([p[:, :c] for p, c in zip(parts, counts)], dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", "-d", type=str, choices=["clean", "final"], default="clean")
    ap.add_argument("--acc", "-acc", type=str, choices=["acc", "direct"], default="acc")
    ap.add_argument("--acc_ckpt", type=str, default=None)
    ap.add_argument("--ofe", "-ofe", type=str, choices=["raft", "gma"], default="raft")
    ap.add_argument("--ofe_ckpt", type=str, default=None)
    ap.add_argument("--batch", type=int, default=10)   # test_cvo.py:114
    ap.add_argument("--result-dir", type=str, default=".",
                    help="where test_result_<split>_E6.txt is appended (test_cvo.py:163-165 writes to the working directory)")
    a = ap.parse_args()
    import torch.distributed as dist
    from accflow_amd.data.dataset import fetch_valid_dataloader
    from accflow_amd.parallel import SequencePipeline, block_partition
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    end = 6  # CVO-6 (test_cvo.py:116)
    model = build_model(a.acc, a.ofe, a.acc_ckpt, a.ofe_ckpt, dev)
    loader, _ = fetch_valid_dataloader(keys=["fflows", "bflows"], split=a.data, batch=a.batch)
    alls, occs, viss = [], [], []

    def account(FN0, labels):
        bflows, fflows = labels
        bmask, _ = calc_occ_mask(bflows[-1], fflows[-1])
        e_all, e_occ, e_vis = cal_epe(FN0, bflows[-1], bmask)
        alls.append(e_all), occs.append(e_occ), viss.append(e_vis)

    # batches are independent: the fusion chain of batch k runs underneath the estimator of batch k+1
    pipe = SequencePipeline(model) if a.acc == "acc" else None
    in_flight = []
    for index, batch in enumerate(loader):
        n = batch["imgs"].shape[0]
        mine = block_partition(n, world, rank)
        if not mine:
            continue
        batch = {k: v[mine[0]:mine[-1] + 1] for k, v in batch.items()}
        d = preprocess(batch, dev)
        imgs, bflows, fflows = d["imgs"][:end + 1], d["bflows"][:end - 1], d["fflows"][:end - 1]
        with torch.no_grad():
            if pipe is None:
                account(model(imgs[-1], imgs[0]), (bflows, fflows))
                continue
            in_flight.append((bflows, fflows))
            outs = pipe.submit(imgs)
        if outs:   # (None: nothing harvested yet; []: a sequence of fewer than 3 frames has no accumulated flow)
            account(outs[-1], in_flight.pop(0))
    if pipe is not None:
        outs = pipe.flush()
        if outs:
            account(outs[-1], in_flight.pop(0))
    res = gather_metrics(alls, occs, viss, dev, world, rank)
    avg = None
    if rank == 0:
        # test_cvo.py:157-166: plain torch.mean over the per-sample values (a sample without occluded pixels has
        # epe_occ = 0/0 = NaN and makes the occ average NaN, as in the reference), printed and APPENDED to the result file
        name = a.acc + "|" + a.ofe
        avg = (float(res[0].mean()), float(res[2].mean()), float(res[1].mean()))   # all, vis, occ
        print("Finish".center(50, "="))
        print("AVG EPE %s: " % name)
        print("all:%.4f vis:%.4f occ:%.4f" % avg)
        with open(os.path.join(a.result_dir, "test_result_%s_E%d.txt" % (a.data, end)), "a+") as f:
            f.write("AVG EPE %s: \n" % name)
            f.write("all:%.4f vis:%.4f occ:%.4f \n\n" % avg)
    if world > 1:
        dist.destroy_process_group()
    return avg


if __name__ == "__main__":
    main()
