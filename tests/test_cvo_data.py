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
    """The evaluation harness (test_cvo.py semantics) end to end on the LMDB data layer: 2 sequences of 7 x 128 x 128
    through AccFlow(RAFT) on the GPU; prints the three EPE averages like test_cvo.py:157-166."""
    FX.make_cvo(str(tmp_path), n_samples=2, size=128)
    monkeypatch.setenv("ACCFLOW_CVO_LMDB", str(tmp_path / "cvo_test.lmdb"))
    monkeypatch.setattr(sys, "argv", ["eval_cvo.py", "-d", "clean", "-acc", "acc", "-ofe", "raft", "--batch", "2"])
    from accflow_amd import eval_cvo
    eval_cvo.main()
    out = capsys.readouterr().out
    assert "AVG EPE acc|raft" in out and "all:" in out and "nan" not in out.split("all:")[1].split()[0]
