"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the host mirror
keeps the reference's import surface / state_dict layout, the product refuses to run without the GPU, and the
scheduling helpers are right."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "accflow_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long)\s+(accflow_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from accflow_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), "libaccflow_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding missing for %s" % n
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.accflow_abi_version() == _lib.ABI_VERSION == 20
    assert lib.accflow_conv_kpad(3, 7, 7) == 160 and lib.accflow_conv_coutpad(126) == 128
    # the library must not drag in a second HIP runtime (it binds to the host process's)
    import subprocess
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip64" not in needed
    # ... and the other way round: the product library exports NOTHING under the accflow_ prefix that the header does not
    # declare (debug hooks such as round 5's accflow_debug_lc1_prof exist in tools builds only), and no mutable global
    # variable of its own ("no global mutable state"; static environment-derived constants are function-local)
    dyn = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in dyn.splitlines() if ln.split() and ln.split()[-1].startswith("accflow_")
                       and ln.split()[-2] in "TW"})
    assert exported == names, (sorted(set(exported) - set(names)), sorted(set(names) - set(exported)))
    syms = subprocess.run(["readelf", "-sW", "--dyn-syms", _lib.LIB_PATH], capture_output=True, text=True).stdout
    objs = {}
    for ln in syms.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] in ("OBJECT", "TLS") and f[4] == "GLOBAL" and f[6] != "UND":
            if not (f[7].startswith("__hip_cuid_") or "kernel" in f[7]):      # (hipcc's unit ids and kernel handles)
                objs[f[7]] = f[3]
    # the two per-THREAD routing pointers of the dry-run protocol (conv_common.h: set and cleared inside one call) are the
    # library's only global data symbols
    assert objs == {"accflow_tls_dry_route": "TLS", "accflow_tls_dry_slots": "TLS"}, objs


def _struct_fields(name):
    src = open(os.path.join(ROOT, "include", "accflow_hip.h")).read()
    body = src[src.index("typedef struct %s {" % name):src.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        for part in stmt.split(","):
            fields.append(re.findall(r"([A-Za-z0-9_]+)\s*(?:\[[A-Za-z0-9_]+\])?$", part.strip())[0])
    return fields


def test_conv_desc_matches_header_layout():
    from accflow_amd._lib import ConvDesc, ConvSrc, MAX_SRC
    assert _struct_fields("accflow_conv_desc") == [f[0] for f in ConvDesc._fields_]
    assert _struct_fields("accflow_conv_src") == [f[0] for f in ConvSrc._fields_]
    assert ctypes.sizeof(ConvDesc) % 8 == 0
    # the library reports the sizes it was compiled with: the ctypes mirror must match field for field (checked at load, too)
    from accflow_amd import _lib as L
    lib = L.load()
    assert lib.accflow_conv_desc_bytes() == ctypes.sizeof(ConvDesc) and lib.accflow_conv_src_bytes() == ctypes.sizeof(ConvSrc)
    # the kernel reads a source as 16 dwords of the kernarg segment (csrc/conv_s16m_kernel.h)
    assert ctypes.sizeof(ConvSrc) == 64 and ConvDesc.src.size == 64 * MAX_SRC


def test_import_surface_of_test_cvo():
    """test_cvo.py:5-8"""
    from data import dataset
    from networks import build_flow_estimator
    from networks.AccFlow_ import AccFlow
    from networks.utils import backwarp
    assert callable(backwarp) and callable(dataset.fetch_valid_dataloader)
    m = AccFlow(build_flow_estimator("acc|raft"))
    keys = set(m.state_dict())
    for k in ("ofe.fnet.conv1.weight", "ofe.cnet.layer2.0.downsample.1.running_var", "ofe.cnet.layer2.0.norm3.weight",
              "ofe.update_block.gru.convq2.bias", "ofe.update_block.mask.2.weight", "flow_encoder.conv3.bias",
              "flow_decoder.mask.2.weight", "context.layer3.1.conv2.weight", "accplus.conv2.4.scale",
              "accplus.conv2.4.conv.weight", "accplus.dconv.weight", "accplus.dconv.bias", "accplus.conv4.4.bias",
              "blending.mask.2.weight"):
        assert k in keys, k
    assert tuple(m.state_dict()["accplus.conv2.4.scale"].shape) == (1, 27, 1, 1)
    g = build_flow_estimator("gma")
    for k in ("att.to_qk.weight", "att.pos_emb.rel_ind", "att.pos_emb.rel_height.weight", "update_block.aggregator.gamma",
              "update_block.aggregator.to_v.weight", "update_block.gru.convz1.weight"):
        assert k in g.state_dict(), k
    assert tuple(g.state_dict()["update_block.gru.convz1.weight"].shape) == (128, 512, 1, 5)
    # DataParallel-prefixed checkpoints (test_cvo.py:18-19) load through the wrapper
    dp = torch.nn.DataParallel(m)
    dp.load_state_dict({"module." + k: v for k, v in m.state_dict().items()})


def test_product_has_no_cpu_path():
    from networks import build_flow_estimator
    from networks.utils import backwarp
    m = build_flow_estimator("raft").eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 128, 256), torch.zeros(1, 3, 128, 256))
    with pytest.raises(RuntimeError):
        backwarp(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))
    # nothing under accflow_amd/ (nor the drop-in shims) may import the oracle
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle)", re.M)
    for top in ("accflow_amd", "networks", "data"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(".py"):
                    assert not pat.search(open(os.path.join(dp, f)).read()), os.path.join(dp, f)


def test_pair_schedule_and_partitions():
    from accflow_amd.networks.AccFlow_ import AccFlow
    from accflow_amd.parallel import block_partition, round_robin
    assert AccFlow.pair_schedule(7) == [(2, 1), (2, 0), (1, 0), (3, 2), (3, 0), (4, 3), (4, 0), (5, 4), (5, 0),
                                        (6, 5), (6, 0)]
    assert len(AccFlow.pair_schedule(7)) == 11 and AccFlow.pair_schedule(2) == []
    for n in (0, 1, 7, 8, 11):
        for w in (1, 2, 3, 8):
            parts = [block_partition(n, w, r) for r in range(w)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
            rr = [round_robin(n, w, r) for r in range(w)]
            assert sorted(i for p in rr for i in p) == list(range(n))
    from accflow_amd.parallel import deal_pairs
    pairs = AccFlow.pair_schedule(7)
    for w in (1, 2, 3, 4, 8):
        flat = deal_pairs(pairs, w)
        assert flat == [list(range(r, 11, w)) for r in range(w)]                      # per-pair deal = round robin
        kept = deal_pairs(pairs, w, keep_together=True)
        assert sorted(k for d in kept for k in d) == list(range(11))
        owner = {}
        for r, d in enumerate(kept):
            for k in d:
                assert owner.setdefault(pairs[k][0], r) == r                          # one rank per image1
    assert max(map(len, deal_pairs(pairs, 8, keep_together=True))) == 2                # 6 groups of <= 2 over 8 ranks


def test_dataset_contract_and_determinism():
    from accflow_amd.data.dataset import fetch_valid_dataloader
    from accflow_amd.data.synthetic import gt_flow, make_sequence, make_state_dict
    os.environ["ACCFLOW_SYNTH_SAMPLES"] = "3"
    loader, ds = fetch_valid_dataloader(keys=["fflows", "bflows"], split="clean", batch=2)
    ds.size = (64, 96)
    ds.__init__(["fflows", "bflows"], "clean", 3, (64, 96))
    b = next(iter(loader))
    assert tuple(b["imgs"].shape) == (2, 21, 64, 96) and tuple(b["fflows"].shape) == (2, 10, 64, 96)
    assert 0.0 <= float(b["imgs"].min()) and float(b["imgs"].max()) <= 255.0
    a1, a2 = make_sequence(7, 3, 32, 48), make_sequence(7, 3, 32, 48)
    assert all(torch.equal(x, y) for x, y in zip(a1, a2))
    # forward and backward analytic flows are mutually consistent: p + F(0->i)(p) then + F(i->0) returns to p
    f, bk = gt_flow(0, 3, 64, 96), gt_flow(3, 0, 64, 96)
    assert float((f[:, 32, 48] + bk[:, 32 + int(round(float(f[1, 32, 48]))), 48 + int(round(float(f[0, 32, 48])))]).abs().max()) < 0.05
    from accflow_amd.networks import build_flow_estimator
    m = build_flow_estimator("raft")
    s1, s2 = make_state_dict(m), make_state_dict(m)
    assert all(torch.equal(s1[k], s2[k]) for k in s1)
    assert torch.equal(s1["cnet.layer2.0.norm3.weight"], s1["cnet.layer2.0.downsample.1.weight"])
    assert float(s1["cnet.norm1.running_var"].min()) > 0


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """Error behaviour of the C-ABI (include/accflow_hip.h: "return value 0 or a hipError_t ... nothing throws or
    aborts"): argument validation happens before any launch, so it can be exercised without a device."""
    from accflow_amd import _lib
    from accflow_amd._lib import ConvDesc
    lib = _lib.load()
    z = ctypes.c_void_p(0)
    assert lib.accflow_conv2d_f32(None, z) == 1
    d = ConvDesc()  # all-zero descriptor: no input, no weights
    assert lib.accflow_conv2d_f32(ctypes.byref(d), z) == 1
    assert lib.accflow_conv_pack_f32(z, z, 8, 8, 3, 3, 8, 0, z, z, z) == 1
    assert lib.accflow_conv_pack_bf16s(z, z, 8, 8, 3, 3, z, z) == 1
    assert lib.accflow_corr_volume_f32(z, z, z, z, z, z, 1, 256, 16, 32, z) == 1
    assert lib.accflow_corr_lookup_f32(z, z, z, z, z, z, 0, 1, 16, 32, z) == 1
    assert lib.accflow_convex_upsample_f32(z, 0, z, 0, z, 1, 16, 32, z) == 1
    assert lib.accflow_backwarp_f32(z, 0, z, 0, z, 0, 1, 3, 8, 8, z) == 1
    assert lib.accflow_downflow8_f32(z, z, 1, 2, 64, 64, z) == 1
    assert lib.accflow_instance_norm_f32(z, z, z, 1, 1, 16, ctypes.c_float(1e-5), 1, z) == 1
    assert lib.accflow_gma_attention_f32(z, z, 1, 128, 64, ctypes.c_float(0.1), z) == 1
    # frame indices of the packed correlation call are HOST arrays, validated against the frame count before any launch
    nz = ctypes.c_void_p(64)   # (never dereferenced: the index check comes first)
    ok_idx, bad_idx, neg_idx = (ctypes.c_int * 2)(0, 6), (ctypes.c_int * 2)(0, 7), (ctypes.c_int * 2)(-1, 0)
    for i1, i2 in ((bad_idx, ok_idx), (ok_idx, bad_idx), (neg_idx, ok_idx)):
        assert lib.accflow_corr_volume_disp_packed_f32(nz, 7, i1, i2, nz, nz, nz, nz, 4, z, 2, 256, 16, 32, z) == 1
    assert lib.accflow_corr_volume_disp_packed_f32(nz, 0, ok_idx, ok_idx, nz, nz, nz, nz, 4, z, 2, 256, 16, 32, z) == 1
    # ... and so are the item indices of the gathered context split
    one = (ctypes.c_int * 1)(3)
    assert lib.accflow_split_tanh_relu_idx_f32(nz, 3, one, nz, 0, nz, 0, 1, 128, 128, 64, z) == 1
    # sizes that are derivable without a device
    assert lib.accflow_conv_kpad(256, 3, 3) == 2304 and lib.accflow_conv_kpad(2, 7, 7) == 128
    assert lib.accflow_conv_coutpad(576) == 640
    assert lib.accflow_conv_patch_elems(256, 128, 3, 3) == 3 * 8 * 9 * 2 * 256 * 8


def test_ops_reject_non_cuda_and_bad_layouts():
    from accflow_amd import ops
    x = torch.zeros(1, 4, 8, 8)
    for fn in (lambda: ops.backwarp(x, torch.zeros(1, 2, 8, 8)), lambda: ops.downflow8(torch.zeros(1, 2, 64, 64)),
               lambda: ops.convex_upsample(torch.zeros(1, 2, 8, 8), torch.zeros(1, 576, 8, 8)),
               lambda: ops.instance_norm(x, 1), lambda: ops.corr_volume(x, x)):
        with pytest.raises(RuntimeError):
            fn()
    assert ops.conv_mode_name() in ("f32", "bf16x3", "bf16x6", "f16x3")
    with pytest.raises(KeyError):
        ops.set_conv_mode("fp8")


def test_reference_helper_surface():
    """networks/raft/utils/utils.py:7-28,66-80,90-93 of the reference: InputPadder, bilinear_sampler, upflow8 (API helpers
    beside the hot path; VERDICT r03 'missing' #4)."""
    from accflow_amd.networks.raft.utils.utils import InputPadder, bilinear_sampler, upflow8
    from oracle import accflow_oracle as O
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 436, 1024, generator=g)
    for mode, top in (("sintel", 2), ("kitti", 0)):
        p = InputPadder(x.shape, mode=mode)
        y, = p.pad(x)
        assert tuple(y.shape[-2:]) == (440, 1024) and torch.equal(p.unpad(y), x)
        assert torch.equal(y[..., :top, :], x[..., :1, :].expand(-1, -1, top, -1))          # replicate padding
    f = torch.randn(1, 2, 5, 7, generator=g)
    up = upflow8(f)
    assert tuple(up.shape) == (1, 2, 40, 56)
    assert torch.allclose(up[..., 0, 0], 8 * f[..., 0, 0]) and torch.allclose(up[..., -1, -1], 8 * f[..., -1, -1])  # align_corners
    img = torch.randn(1, 4, 9, 11, generator=g)
    coords = torch.rand(1, 9, 11, 2, generator=g) * torch.tensor([12.0, 10.0]) - 1.0     # some samples outside
    out, m = bilinear_sampler(img, coords, mask=True)
    grid = torch.stack(torch.meshgrid(torch.arange(9.0), torch.arange(11.0), indexing="ij")[::-1], 0)[None]
    want = O.backwarp(img, coords.permute(0, 3, 1, 2) - grid)                             # the oracle's bilinear-zeros
    assert torch.allclose(out, want, atol=1e-5)
    assert m.shape == (1, 9, 11, 1) and 0.0 < float(m.mean()) < 1.0


def test_fused_lookup_weight_order():
    """ops.lookup_fused_weight = the permutation include/accflow_hip.h documents for accflow_corr_lookup_convc1_s16: tap
    n = j*9 + i of level l (reference channel l*81 + i*9 + j, raft/corr.py:34-45) at k = 32*(n // 8) + 8*l + n % 8 for
    n < 80, at k = 320 + l for n = 80; k = 324..335 zero."""
    import torch
    from accflow_amd import ops
    w = torch.arange(2 * 324, dtype=torch.float32).reshape(2, 324, 1, 1) + 1.0
    f = ops.lookup_fused_weight(w).reshape(2, -1)
    assert f.shape[1] == ops.LOOKUP_FUSED_K == 336
    for l in range(4):
        for j in range(9):
            for i in range(9):
                n = j * 9 + i
                k = 32 * (n // 8) + 8 * l + n % 8 if n < 80 else 320 + l
                assert float(f[0, k]) == l * 81 + i * 9 + j + 1.0 and float(f[1, k]) == 324 + l * 81 + i * 9 + j + 1.0
    assert float(f[:, 324:].abs().max()) == 0.0


def test_tapgemm_channel_order_is_the_accumulator_layout():
    """ops.tapgemm_channel_order (ACCFLOW_EPI_TAPGEMM's second product): a permutation inside every 32-channel block such
    that K-step s, lane half h, slot e of the packed reduction is accumulator row 8 (2 s + e // 4) + 4 h + e % 4 - the rows
    a lane of a v_mfma_f32_32x32x16 result tile holds (8 i + 4 h + j)."""
    from accflow_amd.ops import tapgemm_channel_order
    order = tapgemm_channel_order(256)
    assert sorted(order.tolist()) == list(range(256))
    for blk in range(8):
        o = order[32 * blk:32 * blk + 32] - 32 * blk
        assert sorted(o.tolist()) == list(range(32))
        for s in range(2):
            for h in range(2):
                rows = o[16 * s + 8 * h:16 * s + 8 * h + 8].tolist()
                lane_rows = [8 * i + 4 * h + j for i in (2 * s, 2 * s + 1) for j in range(4)]   # what lane half h holds for i
                assert rows == lane_rows, (blk, s, h, rows)


def test_schedule_scopes_are_thread_local_and_restore():
    """Round-6 host switches that change arithmetic or schedules per scope: ops.ksplit_scope (split-K workspaces off for the
    convolutions of a pipelined fusion chain), AccFlow_.chain_in_pipeline / pipeline_chain_arithmetic (what sets it): nested
    scopes restore, an exception restores, another thread is unaffected."""
    import threading
    from accflow_amd import ops
    from accflow_amd.networks import AccFlow_ as A
    assert ops._ksplit_on() == ops.USE_KSPLIT
    with ops.ksplit_scope(False):
        assert not ops._ksplit_on()
        with ops.ksplit_scope(True):
            assert ops._ksplit_on() == ops.USE_KSPLIT
        assert not ops._ksplit_on()
        seen = []
        t = threading.Thread(target=lambda: seen.append(ops._ksplit_on()))
        t.start(); t.join()
        assert seen == [ops.USE_KSPLIT]
    assert ops._ksplit_on() == ops.USE_KSPLIT
    try:
        with ops.ksplit_scope(False):
            raise KeyError("x")
    except KeyError:
        pass
    assert ops._ksplit_on() == ops.USE_KSPLIT
    flag = lambda: getattr(A._CHAIN_TLS, "no_ksplit", False)
    assert not flag() and A._hoist_now() == (A.USE_CHAIN_HOIST in ("auto", "1", True))
    with A.chain_in_pipeline():
        assert flag() == (not A.PIPELINE_CHAIN_KSPLIT)
        if A.USE_CHAIN_HOIST == "auto":
            assert not A._hoist_now()
    assert not flag()
    with A.pipeline_chain_arithmetic():
        assert flag() == (not A.PIPELINE_CHAIN_KSPLIT)
        if A.USE_CHAIN_HOIST == "auto":
            assert A._hoist_now()              # (a plain forward keeps its own schedule: only the arithmetic is the pipeline's)
    assert not flag()


def test_pair_group_cuts():
    """RAFT._group_cuts: the split of a pair batch over the two group streams - floor / ceil halves (`flip`: the pipeline's split
    mode alternates them); RAFTGMA's cuts never separate items that share an attention matrix."""
    from accflow_amd.networks.AccFlow_ import AccFlow
    from accflow_amd.networks.gma.gma import RAFTGMA
    from accflow_amd.networks.raft.raft import RAFT
    assert RAFT._group_cuts(11, 2) == [0, 5, 11] and RAFT._group_cuts(11, 2, flip=True) == [0, 6, 11]
    assert RAFT._group_cuts(10, 2, flip=True) == [0, 5, 10] and RAFT._group_cuts(3, 1) == [0, 3]
    ids = [(i, 0) for i, _ in AccFlow.pair_schedule(7)]
    assert [i for i, _ in ids] == [2, 2, 1, 3, 3, 4, 4, 5, 5, 6, 6]
    assert RAFT._group_cuts(11, 2, flip=True, ctx_ids=ids) == [0, 6, 11]          # (RAFT: context features are per item)
    for flip in (False, True):
        c = RAFTGMA._group_cuts(11, 2, flip=flip, ctx_ids=ids)
        assert c == [0, 5, 11] and ids[c[1]] != ids[c[1] - 1]
    ids2 = [(0, 0)] * 3 + [(1, 0)] * 3                                               # a cut inside a run moves to its end
    assert RAFTGMA._group_cuts(6, 2, ctx_ids=ids2) == [0, 3, 6] and RAFTGMA._group_cuts(6, 2, ctx_ids=[(0, 0)] * 4 + [(1, 0)] * 2) == [0, 4, 6]
