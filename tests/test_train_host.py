"""Host logic of the training slice that needs no GPU: the gradient all-reduce of the data-parallel mode (gloo, world 2)
and the training data path (random crop, reference data/augmentor.py:9-21)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from accflow_amd.train import allreduce_grads
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 3), torch.nn.Conv2d(3, 1, 1))
    ps = list(net.parameters())
    for i, p in enumerate(ps):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    ps[3].grad = None if rank == 0 else ps[3].grad          # a parameter one rank holds no gradient for
    allreduce_grads(ps)
    ok = True
    for i, p in enumerate(ps):
        want = sum((r + 1) * (i + 1) for r in range(world)) / world
        if i == 3:
            want = sum((r + 1) * (i + 1) for r in range(1, world)) / world
        ok &= p.grad is not None and p.grad.shape == p.shape and bool(torch.allclose(p.grad, torch.full_like(p, want)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allreduce_grads_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == {0: True, 1: True}


def test_allreduce_grads_without_process_group_is_a_no_op():
    from accflow_amd.train import allreduce_grads
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    allreduce_grads([p])
    assert torch.equal(p.grad, torch.ones(3))


def test_flow_augmentor_crops_every_array_with_one_window():
    """data/augmentor.py:9-21: one random window for frames and flows."""
    import numpy as np
    from accflow_amd.data.dataset import FlowAugmentor
    np.random.seed(3)
    ys, xs = np.meshgrid(np.arange(40), np.arange(50), indexing="ij")
    s = {"imgs": np.stack([ys, xs, ys + xs], -1).astype(np.float32), "bflows": np.stack([ys * 100 + xs] * 10, -1).astype(np.float32)}
    out = FlowAugmentor(16)(dict(s))
    assert out["imgs"].shape == (16, 16, 3) and out["bflows"].shape == (16, 16, 10)
    y0, x0 = int(out["imgs"][0, 0, 0]), int(out["imgs"][0, 0, 1])
    assert 0 <= y0 < 40 - 16 and 0 <= x0 < 50 - 16
    assert out["bflows"][0, 0, 0] == y0 * 100 + x0 and out["bflows"][15, 15, 9] == (y0 + 15) * 100 + x0 + 15


def test_training_shards_are_disjoint_equal_and_reshuffled():
    from accflow_amd.data.dataset import _Shard
    parts = [list(_Shard(23, r, 4, seed=7)) for r in range(4)]
    assert all(len(p) == 23 // 4 for p in parts) and len({i for p in parts for i in p}) == 4 * (23 // 4)
    s = _Shard(23, 0, 4, seed=7)
    first = list(s)
    s.set_epoch(1)
    assert list(s) != first and list(_Shard(23, 0, 4, seed=7)) == first
    # a resume from the middle of an epoch skips at the index level, once: the next epoch is whole again
    s.set_epoch(0, skip=2)
    assert list(s) == first[2:] and len(s) == 23 // 4
    assert list(s) == first


def test_train_options_and_training_loader(tmp_path, monkeypatch):
    """The reference's YAML keys (configs/AccRAFT-CVO.yml) parse into the options train_acc.py reads; the training loader
    serves the batch-dict contract at the crop size (synthetic source here)."""
    from accflow_amd import train_acc
    from accflow_amd.data.dataset import fetch_train_dataloader
    cfg = tmp_path / "c.yml"
    cfg.write_text("exp_name: Acc+RAFT-cvo\ngpus: [0,1]\nepochs: 60\nlr: !!float 1.2e-4\nwdecay: !!float 1.0e-5\n"
                   "epsilon: !!float 1.0e-8\nmixed_precision: true\nbatch_per_gpu: 6\nloss_type: L1\nclip: 1.0\nadd_noise: true\n"
                   "log_freq: 100\nvalid_freq: 1000\nimage_size: [256, 256]\nflow_pretrained: checkpoints/raft-cvo.pth\n")
    o = train_acc.parse_options(str(cfg))
    assert (o.lr, o.wdecay, o.epsilon, o.batch_per_gpu, o.clip, o.image_size) == (1.2e-4, 1e-5, 1e-8, 6, 1.0, [256, 256])
    monkeypatch.setenv("ACCFLOW_SYNTHETIC", "1")
    monkeypatch.setenv("ACCFLOW_SYNTH_SAMPLES", "3")
    monkeypatch.delenv("ACCFLOW_CVO_LMDB", raising=False)
    loader, dst = fetch_train_dataloader(["bflows"], batch=2, crop_size=32, split="clean+final", rank=1, world=2)
    assert len(dst) == 6 and len(loader) == 1
    b = next(iter(loader))
    assert tuple(b["imgs"].shape) == (2, 21, 32, 32) and tuple(b["bflows"].shape) == (2, 10, 32, 32)
    pg = torch.nn.Parameter(torch.zeros(4))
    opt, sch = train_acc.fetch_optimizer(o, [pg], 1000)
    assert isinstance(opt, torch.optim.AdamW) and sch.total_steps == 1100


def test_train_acc_refuses_silent_synthetic_and_rotates(tmp_path, monkeypatch):
    """a run without the CVO training LMDB / the flow_pretrained checkpoint stops unless --synthetic asks for a
    smoke run (decided before any GPU or process-group call, by every rank); checkpoint rotation keeps what train_acc.py
    :296-302 keeps: once four .pth files exist (latest counts) the oldest numbered one goes."""
    import pytest
    from accflow_amd import train_acc
    monkeypatch.delenv("ACCFLOW_SYNTHETIC", raising=False)
    monkeypatch.delenv("ACCFLOW_CVO_LMDB", raising=False)
    cfg = tmp_path / "c.yml"
    cfg.write_text("exp_name: Acc+RAFT-x\nepochs: 1\nbatch_per_gpu: 2\nimage_size: [64, 64]\nflow_pretrained: none.pth\n")
    with pytest.raises(SystemExit) as e:
        train_acc.main(["-c", str(cfg), "--out", str(tmp_path)])
    assert "--synthetic" in str(e.value)
    d = tmp_path / "ck"
    d.mkdir()
    for name in ("latest", "000100", "000200", "000300"):
        (d / (name + ".pth")).write_bytes(b"x")
        (d / (name + ".state")).write_bytes(b"x")
    (d / "final.pth").write_bytes(b"x")
    train_acc.rotate_ckpts(str(d))
    assert sorted(p.name for p in d.iterdir() if p.name.endswith(".pth")) == ["000200.pth", "000300.pth", "final.pth", "latest.pth"]
    assert not (d / "000100.state").exists()
