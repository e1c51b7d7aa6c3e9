"""the CVO data layer without `lmdb` / legacy pyarrow - pure-Python LMDB reader, legacy pa.deserialize
decoder, the dataset classes with the reference's contract (data/dataset.py:23-108,146-161) - on LMDB files written
by tests/golden/make_cvo_fixture.py."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_cvo_fixture as FX  # noqa: E402

from accflow_amd.data import pa_legacy  # noqa: E402
from accflow_amd.data.lmdb_reader import LMDBError, ReadOnlyLMDB  # noqa: E402


def test_legacy_pyarrow_round_trip():
    rng = np.random.default_rng(0)
    a8 = rng.integers(0, 255, size=(5, 7, 21), dtype=np.uint8)
    a16 = rng.integers(0, 65535, size=(4, 6, 10), dtype=np.uint16)
    for obj in (a8, a16, [0, 1, 2, 3], ["imgs", "fflows"], [[1, 2], ["a"], 7], 12345678901, "x",
                rng.standard_normal((3, 2)).astype(np.float32)):
        back = pa_legacy.deserialize(FX.legacy_serialize(obj))
        if isinstance(obj, np.ndarray):
            assert back.dtype == obj.dtype and back.shape == obj.shape and np.array_equal(back, obj)
        else:
            assert back == obj
    buf = FX.legacy_serialize(a16)
    assert buf[:16] == np.array([0, 0, 1, 0], dtype="<i4").tobytes()   # tensors, sparse, ndarrays, buffers
    with pytest.raises(ValueError):
        pa_legacy.deserialize(b"\x00" * 8)


def test_lmdb_reader_small_deep_and_overflow(tmp_path):
    rng = np.random.default_rng(1)
    items = {("key%06d" % i).encode(): bytes(rng.integers(0, 255, size=int(rng.integers(0, 200)), dtype=np.uint8))
             for i in range(40000)}                                          # three levels of pages
    items[b"big_a"] = bytes(rng.integers(0, 255, size=300000, dtype=np.uint8))  # overflow runs
    items[b"big_b"] = bytes(rng.integers(0, 255, size=4080, dtype=np.uint8))    # exactly one overflow page
    items[b"edge"] = bytes(2030 - 8 - 4)                                        # largest inline node
    items[b""] = b"empty key"
    FX.write_lmdb(str(tmp_path / "t.lmdb"), items)
    env = ReadOnlyLMDB(str(tmp_path / "t.lmdb"))
    assert len(env) == len(items) and env.depth >= 3
    for k in list(items)[::97] + [b"big_a", b"big_b", b"edge", b"", b"key000000", b"key039999"]:
        assert env.get(k) == items[k], k
    assert env.get(b"nope") is None and env.get(b"key0000005") is None and env.get(b"zzz") is None
    walked = list(env.items())
    assert [k for k, _ in walked] == sorted(items) and all(items[k] == v for k, v in walked)
    env.close()
    empty = tmp_path / "e.lmdb"
    FX.write_lmdb(str(empty), {})
    e = ReadOnlyLMDB(str(empty))
    assert len(e) == 0 and e.get(b"a") is None and list(e.keys()) == []
    bad = tmp_path / "bad.mdb"
    bad.write_bytes(b"\0" * 8192)
    with pytest.raises(LMDBError):
        ReadOnlyLMDB(str(bad))
    with pytest.raises(LMDBError):
        ReadOnlyLMDB(str(tmp_path / "missing.lmdb"))


@pytest.mark.parametrize("psize", [4096, 16384])
@pytest.mark.parametrize("torn", [None, "root", "last_pg"])
def test_lmdb_reader_real_tool_artefacts(tmp_path, psize, torn):
    """What a data.mdb written by the real tools over several transactions contains and a bulk-written file does not:
    a populated free-list database with released pages (stale tree-page images) between the live ones, the current
    snapshot in EITHER meta page, a second meta page with a larger txnid that does not hold up (torn commit), overflow
    values spanning many pages (their run length in the first page's header only), a 16 KB page size."""
    rng = np.random.default_rng(5)
    items = {("k%05d" % i).encode(): bytes(rng.integers(0, 255, size=int(rng.integers(1, 900)), dtype=np.uint8))
             for i in range(3000)}
    items[b"big3"] = bytes(rng.integers(0, 255, size=3 * psize + 77, dtype=np.uint8))      # 4 overflow pages
    items[b"big_exact"] = bytes(rng.integers(0, 255, size=2 * psize - 16, dtype=np.uint8))  # exactly 2 pages incl. header
    items[b"big_plus1"] = bytes(rng.integers(0, 255, size=2 * psize - 15, dtype=np.uint8))  # one byte into a 3rd page
    items[b"huge"] = bytes(rng.integers(0, 255, size=1 << 20, dtype=np.uint8))
    for txnid in (6, 7):                      # good snapshot in meta page 0, then in meta page 1
        d = tmp_path / ("t%d" % txnid)
        info = FX.write_lmdb(str(d), items, psize=psize, free_pages=9, torn_meta=torn, txnid=txnid)
        assert len(info["freed"]) == 9 and info["depth"] >= 2
        env = ReadOnlyLMDB(str(d))
        assert env.psize == psize and env.txnid == txnid and env.free_db_entries == 1 and len(env) == len(items)
        assert bool(env.skipped_metas) == (torn is not None)
        for k in list(items)[::53] + [b"big3", b"big_exact", b"big_plus1", b"huge"]:
            assert env.get(k) == items[k], k
        assert [k for k, _ in env.items()] == sorted(items)
        assert env.get(b"stale") is None                       # released pages are never reached
        env.close()
    # both meta pages unusable -> a loud error, not garbage
    raw = bytearray((tmp_path / "t7" / "data.mdb").read_bytes())
    import struct
    struct.pack_into("<Q", raw, 1 * psize + 16 + 24 + 48 + 40, 3 * len(raw))     # main root of meta 1 -> nowhere
    if torn is None:
        struct.pack_into("<Q", raw, 0 * psize + 16 + 24 + 48 + 40, 3 * len(raw))
        struct.pack_into("<Q", raw, 0 * psize + 16 + 24 + 48 + 32, 5)            # (entries without a reachable root)
    (tmp_path / "broken.mdb").write_bytes(bytes(raw))
    with pytest.raises(LMDBError):
        ReadOnlyLMDB(str(tmp_path / "broken.mdb"))
    # an overflow run whose header claims fewer pages than the value needs
    good = bytearray((tmp_path / "t6" / "data.mdb").read_bytes())
    hit = 0
    for pg in range(2, len(good) // psize):
        own, _, fl, npg = struct.unpack_from("<QHHI", good, pg * psize)
        if own == pg and fl == FX.P_OVERFLOW and npg == 4:
            struct.pack_into("<I", good, pg * psize + 12, 2)
            hit += 1
    assert hit == 1
    (tmp_path / "short_run.mdb").write_bytes(bytes(good))
    env = ReadOnlyLMDB(str(tmp_path / "short_run.mdb"))
    with pytest.raises(LMDBError):
        env.get(b"big3")
    env.close()


def test_cvo_dataset_contract(tmp_path, monkeypatch):
    path, truth = FX.make_cvo(str(tmp_path), n_samples=3, size=128)
    monkeypatch.setenv("ACCFLOW_CVO_LMDB", str(tmp_path))      # a directory holding cvo_test.lmdb
    from data.dataset import CVO, CVO_sampler_lmdb, fetch_valid_dataloader   # the import surface of test_cvo.py:5
    smp = CVO_sampler_lmdb(False, ["fflows", "imgs"])
    assert len(smp) == 3 and smp.samples == [0, 1, 2]
    rec = smp.sample(1)
    assert list(rec) == ["fflows", "imgs"] and rec["imgs"].dtype == np.uint8 and rec["fflows"].dtype == np.float32
    assert np.array_equal(rec["imgs"], truth[1]["imgs"]) and np.array_equal(rec["fflows"], truth[1]["fflows"])
    with pytest.raises(AssertionError):
        CVO_sampler_lmdb(False, ["nope"])
    for split, key in (("clean", "imgs"), ("final", "imgs_blur")):
        loader, ds = fetch_valid_dataloader(keys=["fflows", "bflows"], split=split, batch=2)
        assert isinstance(ds, CVO) and len(ds) == 3
        batches = list(loader)
        assert [b["imgs"].shape[0] for b in batches] == [2, 1]          # drop_last=False
        b = batches[0]
        assert set(b) == {"imgs", "fflows", "bflows"}
        assert tuple(b["imgs"].shape) == (2, 21, 128, 128) and b["imgs"].dtype == torch.float32
        assert tuple(b["fflows"].shape) == (2, 10, 128, 128) and tuple(b["bflows"].shape) == (2, 10, 128, 128)
        assert torch.equal(b["imgs"][1], torch.from_numpy(truth[1][key]).permute(2, 0, 1).float())
        assert torch.equal(b["bflows"][0], torch.from_numpy(truth[0]["bflows"]).permute(2, 0, 1))
        # the uint16 code has 1/128 px resolution: decoded flows sit within 1/256 px of the analytic ground truth
        from accflow_amd.data.synthetic import gt_flow
        gt = torch.cat([gt_flow(i, 0, 128, 128) for i in range(2, 7)], 0)
        assert float((b["bflows"][0] - gt).abs().max()) <= 1 / 256 + 1e-6
    loader, ds = fetch_valid_dataloader(keys=["fflows", "bflows"], split="clean+final", batch=4)
    assert len(ds) == 6
    monkeypatch.setenv("ACCFLOW_CVO_LMDB", str(tmp_path / "nowhere"))
    with pytest.raises(FileNotFoundError):
        fetch_valid_dataloader(keys=["fflows"], split="clean", batch=1)


def test_cvo_record_sanity_and_eos_width(tmp_path):
    """A record of the wrong dtype is rejected by the sampler; the decoder finds the tensor whether the IPC stream ended
    in the 8-byte or the 4-byte (pyarrow < 0.15) end-of-stream marker."""
    path, truth = FX.make_cvo(str(tmp_path), n_samples=1, size=128)
    from accflow_amd.data.dataset import CVO_sampler_lmdb
    items = dict(ReadOnlyLMDB(path).items())
    items[b"00000_fflows"] = FX.legacy_serialize(truth[0]["fflows"].astype(np.float32))    # not the uint16 code
    FX.write_lmdb(str(tmp_path / "wrong.lmdb"), items)
    with pytest.raises(RuntimeError, match="expected an"):
        CVO_sampler_lmdb(False, ["fflows"], db_path=str(tmp_path / "wrong.lmdb")).sample(0)
    a = np.arange(5 * 7 * 21, dtype=np.uint8).reshape(5, 7, 21)
    buf = FX.legacy_serialize(a)
    eos = buf.find(b"\xff\xff\xff\xff\x00\x00\x00\x00")
    assert eos > 16
    # the same object with the old 4-byte end-of-stream marker: the tensor then starts at another 64-byte boundary
    # relative to the reader's position
    head, tail = buf[:eos], buf[eos + 8:]
    tensor = tail[tail.find(b"\xff\xff\xff\xff"):]
    old = head + b"\x00\x00\x00\x00"
    old += b"\0" * (-len(old) % 64) + tensor
    assert np.array_equal(pa_legacy.deserialize(buf), a)
    assert np.array_equal(pa_legacy.deserialize(old), a)


@pytest.mark.skipif(not os.environ.get("ACCFLOW_REAL_CVO_LMDB"), reason="opt-in: needs the real cvo_test.lmdb (not available offline)")
def test_real_cvo_lmdb_if_present():
    """Opt-in check against a file written by the real tools: ACCFLOW_REAL_CVO_LMDB=/path/to/cvo_test.lmdb."""
    from accflow_amd.data.dataset import CVO_sampler_lmdb
    smp = CVO_sampler_lmdb(False, ["imgs", "imgs_blur", "fflows", "bflows"], db_path=os.environ["ACCFLOW_REAL_CVO_LMDB"])
    assert len(smp) == 536                                        # data/README.md: CVO-test
    rec = smp.sample(smp.samples[0] if smp.samples else 0)
    assert rec["imgs"].shape == (512, 512, 21) and rec["fflows"].shape[2] == 10 and np.isfinite(rec["fflows"]).all()


def test_synthetic_fallback_announces_itself(monkeypatch, capsys):
    monkeypatch.delenv("ACCFLOW_CVO_LMDB", raising=False)
    monkeypatch.delenv("ACCFLOW_SYNTHETIC", raising=False)
    monkeypatch.setenv("ACCFLOW_SYNTH_SAMPLES", "2")
    from accflow_amd.data.dataset import SyntheticCVO, fetch_valid_dataloader
    _, ds = fetch_valid_dataloader(keys=["fflows", "bflows"], split="clean", batch=1)
    assert isinstance(ds, SyntheticCVO) and "NOT CVO" in capsys.readouterr().err
    monkeypatch.setenv("ACCFLOW_SYNTHETIC", "1")
    fetch_valid_dataloader(keys=["fflows", "bflows"], split="clean", batch=1)
    assert capsys.readouterr().err == ""


@pytest.mark.gpu
def test_eval_cvo_on_lmdb_fixture(tmp_path, monkeypatch, capsys):
    """The evaluation harness (test_cvo.py semantics) end to end on the LMDB data layer, as a PARITY test: 2 sequences
    of 7 x 128 x 128 through AccFlow(RAFT) on the GPU; the three EPE averages it prints and appends to
    test_result_clean_E6.txt (test_cvo.py:157-166) must equal what the oracle computes from the same LMDB records
    (oracle AccFlow forward + calc_occ_mask + cal_epe, fp32 CPU) within 1e-3 px."""
    from oracle import accflow_oracle as O
    from accflow_amd.data.synthetic import make_state_dict
    from accflow_amd.networks import build_flow_estimator
    from accflow_amd.networks.AccFlow_ import AccFlow
    _, truth = FX.make_cvo(str(tmp_path), n_samples=2, size=128)
    monkeypatch.setenv("ACCFLOW_CVO_LMDB", str(tmp_path / "cvo_test.lmdb"))
    monkeypatch.setattr(sys, "argv", ["eval_cvo.py", "-d", "clean", "-acc", "acc", "-ofe", "raft", "--batch", "2",
                                      "--result-dir", str(tmp_path)])
    from accflow_amd import eval_cvo
    got = eval_cvo.main()
    out = capsys.readouterr().out
    assert "AVG EPE acc|raft" in out and "all:" in out
    # the oracle on the records themselves (what the writer stored, decoded as data/dataset.py:60-67 does)
    sd = make_state_dict(AccFlow(build_flow_estimator("acc|raft")))
    e_all, e_occ, e_vis = [], [], []
    for rec in truth:
        frames = O.preprocess_images(torch.from_numpy(rec["imgs"]).permute(2, 0, 1).float()[None])   # 7 x (1,3,H,W)
        bfl = torch.from_numpy(rec["bflows"]).permute(2, 0, 1)[None].split(2, dim=1)[:5]
        ffl = torch.from_numpy(rec["fflows"]).permute(2, 0, 1)[None].split(2, dim=1)[:5]
        fn0 = O.accflow_forward(sd, frames)[-1]
        bmask, _ = O.calc_occ_mask(bfl[-1], ffl[-1])
        a, o, v = O.cal_epe(fn0, bfl[-1], bmask)
        e_all.append(a), e_occ.append(o), e_vis.append(v)
    want = tuple(float(torch.cat(x).mean()) for x in (e_all, e_vis, e_occ))   # all, vis, occ (the printed order)
    for name, g, w in zip(("all", "vis", "occ"), got, want):
        assert (g != g and w != w) or abs(g - w) <= 1e-3, (name, g, w)
    txt = open(tmp_path / "test_result_clean_E6.txt").read()
    assert txt == "AVG EPE acc|raft: \nall:%.4f vis:%.4f occ:%.4f \n\n" % got
    eval_cvo.main()                                                          # a second run APPENDS
    assert open(tmp_path / "test_result_clean_E6.txt").read().count("AVG EPE acc|raft") == 2
