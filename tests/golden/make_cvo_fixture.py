"""Write a tiny CVO-style LMDB (the reference's validation data format, data/dataset.py:23-69, data/README.md) without the
`lmdb` package and without legacy pyarrow - test infrastructure for accflow_amd.data.{lmdb_reader,pa_legacy,dataset}.

    python tests/golden/make_cvo_fixture.py OUT_DIR [--samples 2] [--size 128]

Produces OUT_DIR/cvo_test.lmdb/data.mdb with the reference's key scheme
    __samples__ -> [0, 1, ...]     __valid_keys__ -> [...]     __keys__ -> [...]
    {index:05d}_{imgs|imgs_blur|fflows|bflows} -> ndarray (H, W, 21) uint8 / (H, W, 10) uint16, flows coded
    v = round(f * 128) + 2^15 (the inverse of dataset.py:65-67),
every value in pyarrow's legacy serialisation (see accflow_amd/data/pa_legacy.py for the layout), and returns the float
arrays that went in so that tests can check the whole chain.  Frames / flows come from the build's analytic synthetic
sequence generator (exact ground-truth flow).

Both writers follow the public format descriptions only (LMDB 0.9 `mdb.c` page / node / meta layout; Arrow IPC via the
installed pyarrow): nothing of the reference is involved, and no file written by the real tools exists offline.
"""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# ------------------------------------------------------------------------------------------------------------------
# legacy pyarrow serialisation (writer side)

PT_INT, PT_STRING, PT_LIST, PT_NDARRAY = 2, 5, 10, 15


def _union_of(values, ndarrays):
    """python list -> dense union array over the types present (ints, strs, lists, ndarrays)"""
    import pyarrow as pa
    ints, strs, lists, nds = [], [], [], []
    tids, offs = [], []
    for v in values:
        if isinstance(v, (bool, np.bool_)):
            raise TypeError("bool not needed by the fixture")
        if isinstance(v, (int, np.integer)):
            tids.append(PT_INT); offs.append(len(ints)); ints.append(int(v))
        elif isinstance(v, str):
            tids.append(PT_STRING); offs.append(len(strs)); strs.append(v)
        elif isinstance(v, (list, tuple)):
            tids.append(PT_LIST); offs.append(len(lists)); lists.append(list(v))
        elif isinstance(v, np.ndarray):
            tids.append(PT_NDARRAY); offs.append(len(nds)); nds.append(len(ndarrays)); ndarrays.append(v)
        else:
            raise TypeError(type(v))
    children, names, codes = [], [], []
    if ints:
        children.append(pa.array(ints, type=pa.int64())); codes.append(PT_INT)
    if strs:
        children.append(pa.array(strs, type=pa.string())); codes.append(PT_STRING)
    if lists:
        flat = [x for l in lists for x in l]
        inner = _union_of(flat, ndarrays)
        lo = np.cumsum([0] + [len(l) for l in lists]).astype(np.int32)
        children.append(pa.ListArray.from_arrays(pa.array(lo, type=pa.int32()), inner)); codes.append(PT_LIST)
    if nds:
        children.append(pa.array(nds, type=pa.int32())); codes.append(PT_NDARRAY)
    names = [str(c) for c in codes]
    return pa.UnionArray.from_dense(pa.array(tids, type=pa.int8()), pa.array(offs, type=pa.int32()), children, names, codes)


def legacy_serialize(obj):
    """bytes of `pa.serialize(obj).to_buffer()` for ints / strs / lists / numpy arrays."""
    import pyarrow as pa
    ndarrays = []
    union = _union_of([obj], ndarrays)
    batch = pa.RecordBatch.from_arrays([union], ["list"])
    sink = pa.BufferOutputStream()
    sink.write(struct.pack("<iiii", 0, 0, len(ndarrays), 0))
    w = pa.ipc.new_stream(sink, batch.schema)
    w.write_batch(batch)
    w.close()
    for a in ndarrays:
        sink.write(b"\0" * (-sink.tell() % 64))
        pa.ipc.write_tensor(pa.Tensor.from_numpy(np.ascontiguousarray(a)), sink)
    sink.write(b"\0" * (-sink.tell() % 64))
    return sink.getvalue().to_pybytes()


# ------------------------------------------------------------------------------------------------------------------
# LMDB writer (bulk load of sorted keys)

HDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
F_BIGDATA = 1
P_INVALID = 0xFFFFFFFFFFFFFFFF


def _even(n):
    return (n + 1) & ~1


def write_lmdb(path, items, psize=4096, free_pages=0, torn_meta=None, txnid=1):
    """items: dict bytes -> bytes.  Writes path/data.mdb (path is created).

    Options that reproduce what files written by the real tools over several transactions contain (hand-assembled pages):
      psize       page size (LMDB uses the OS page size: 4096 on x86, 16384 on Apple silicon / some ARM kernels)
      free_pages  n > 0: n released pages holding stale tree-page images are scattered between the live pages and listed in
                  a populated free-list database (dbs[0]: MDB_INTEGERKEY, key = the releasing txnid, value = {count, pgno...})
      torn_meta   None | "root": the OTHER meta page carries a larger txnid but points at a released page (a commit whose
                  meta page reached the disk without its data pages) | "last_pg": ... and a last page beyond the file
      txnid       the good snapshot's transaction id (its meta page is page txnid % 2, as mdb_env_write_meta alternates)"""
    PSIZE = psize
    NODEMAX = (((PSIZE - HDR) // 2) & ~1) - 2   # mdb.c: me_nodemax; larger leaf nodes move their data to overflow pages
    os.makedirs(path, exist_ok=True)
    pages = {}          # pgno -> bytes (PSIZE, or a multiple for overflow runs)
    next_pg = [2]
    freed = []
    rng = np.random.default_rng(12345)

    def add_stale():
        # a page some earlier transaction used and released: a stale leaf image (valid-looking header, old nodes)
        g = next_pg[0]
        next_pg[0] += 1
        stale = bytearray(rng.integers(0, 255, size=PSIZE, dtype=np.uint8).tobytes())
        struct.pack_into("<QHHHH", stale, 0, g, 0, P_LEAF, HDR + 2, PSIZE - 40)
        struct.pack_into("<H", stale, HDR, PSIZE - 40)
        struct.pack_into("<HHHH", stale, PSIZE - 40, 4, 0, 0, 5)
        stale[PSIZE - 32:PSIZE - 27] = b"stale"
        pages[g] = bytes(stale)
        freed.append(g)

    def alloc(n=1):
        if len(freed) < free_pages and next_pg[0] > 2 and rng.random() < 0.25:
            add_stale()
        p = next_pg[0]
        next_pg[0] += n
        return p

    def build_page(flags, nodes, pgno):
        """nodes: list of already encoded node byte strings (in key order)"""
        page = bytearray(PSIZE)
        upper = PSIZE
        ptrs = []
        for nd in nodes:
            upper -= _even(len(nd))
            page[upper:upper + len(nd)] = nd
            ptrs.append(upper)
        lower = HDR + 2 * len(nodes)
        assert lower <= upper, "page overflow"
        struct.pack_into("<QHHHH", page, 0, pgno, 0, flags, lower, upper)
        for i, p in enumerate(ptrs):
            struct.pack_into("<H", page, HDR + 2 * i, p)
        pages[pgno] = bytes(page)

    n_leaf = n_branch = n_over = 0
    level = []          # (first key, pgno) of the pages of the current level
    cur, used, first = [], HDR, None
    keys = sorted(items)

    def flush_leaf():
        nonlocal cur, used, first, n_leaf
        if cur:
            pg = alloc()
            build_page(P_LEAF, cur, pg)
            level.append((first, pg))
            n_leaf += 1
        cur, used, first = [], HDR, None

    for k in keys:
        v = items[k]
        if 8 + len(k) + len(v) > NODEMAX:
            npg = (HDR - 1 + len(v)) // PSIZE + 1
            opg = alloc(npg)
            assert opg not in pages
            run = bytearray(npg * PSIZE)
            struct.pack_into("<QHHI", run, 0, opg, 0, P_OVERFLOW, npg)
            run[HDR:HDR + len(v)] = v
            pages[opg] = bytes(run)
            n_over += npg
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", opg)
        else:
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        need = 2 + _even(len(node))
        if used + need > PSIZE:
            flush_leaf()
        if first is None:
            first = k
        cur.append(node)
        used += need
    flush_leaf()

    depth = 1 if level else 0
    while len(level) > 1:
        nxt, cur, used, first = [], [], HDR, None
        for i, (k, pg) in enumerate(level):
            kk = b"" if not cur else k               # the first key of a branch page is the implicit minimum
            node = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(kk)) + kk
            need = 2 + _even(len(node))
            if used + need > PSIZE:
                bp = alloc()
                build_page(P_BRANCH, cur, bp)
                nxt.append((first, bp))
                n_branch += 1
                cur, used, first = [], HDR, None
                node = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, 0)
                need = 2 + _even(len(node))
            if first is None:
                first = k
            cur.append(node)
            used += need
        bp = alloc()
        build_page(P_BRANCH, cur, bp)
        nxt.append((first, bp))
        n_branch += 1
        level = nxt
        depth += 1
    root = level[0][1] if level else P_INVALID
    while len(freed) < free_pages:       # (a tiny tree may not have interleaved enough of them)
        add_stale()
    free_db = (PSIZE, 0x08, 0, 0, 0, 0, 0, P_INVALID)
    if freed:
        # free-list database: one leaf, one record {key = txnid that released the pages (u64, MDB_INTEGERKEY),
        # value = IDL: count followed by the page numbers, descending as mdb_midl keeps them}
        fpg = next_pg[0]
        next_pg[0] += 1
        idl = struct.pack("<%dQ" % (len(freed) + 1), len(freed), *sorted(freed, reverse=True))
        key = struct.pack("<Q", max(1, txnid - 1))
        build_page(P_LEAF, [struct.pack("<HHHH", len(idl) & 0xFFFF, len(idl) >> 16, 0, len(key)) + key + idl], fpg)
        free_db = (PSIZE, 0x08, 1, 0, 1, 0, 1, fpg)
    last_pg = next_pg[0] - 1

    def meta(pgno, tx, free, main, last):
        page = bytearray(PSIZE)
        struct.pack_into("<QHHHH", page, 0, pgno, 0, P_META, 0, 0)
        struct.pack_into("<IIQQ", page, HDR, 0xBEEFC0DE, 1, 0, (last_pg + 1) * PSIZE)
        struct.pack_into("<IHHQQQQQ", page, HDR + 24, *free)                                       # FREE_DBI
        struct.pack_into("<IHHQQQQQ", page, HDR + 24 + 48, *main)                                  # MAIN_DBI
        struct.pack_into("<QQ", page, HDR + 24 + 96, last, tx)
        return bytes(page)

    good, other = txnid % 2, 1 - txnid % 2
    main_db = (0, 0, depth, n_branch, n_leaf, n_over, len(keys), root)
    empty_db = (0, 0, 0, 0, 0, 0, 0, P_INVALID)
    pages[good] = meta(good, txnid, free_db, main_db, last_pg)
    if torn_meta is None:     # the previous snapshot: what mdb_env_init_meta / an earlier commit left there
        pages[other] = meta(other, txnid - 1, (PSIZE, 0x08, 0, 0, 0, 0, 0, P_INVALID), empty_db, 1)
    elif torn_meta == "root":
        bad_root = freed[0] if freed else last_pg + 7
        torn = (0, 0, depth, n_branch, n_leaf, n_over, len(keys) + 3, bad_root)
        page = bytearray(meta(other, txnid + 1, free_db, torn, last_pg))
        if freed:             # make the released page look like anything but a tree page of that number
            pg = bytearray(pages[bad_root])
            struct.pack_into("<QHH", pg, 0, bad_root + 1, 0, P_OVERFLOW)
            pages[bad_root] = bytes(pg)
        pages[other] = bytes(page)
    elif torn_meta == "last_pg":
        pages[other] = meta(other, txnid + 1, free_db, main_db, last_pg + 1000)
    else:
        raise ValueError(torn_meta)
    with open(os.path.join(path, "data.mdb"), "wb") as f:
        for pg in sorted(pages):
            assert f.tell() == pg * PSIZE, (f.tell(), pg)
            f.write(pages[pg])
    return {"last_pg": last_pg, "freed": list(freed), "root": root, "depth": depth}


# ------------------------------------------------------------------------------------------------------------------
# the CVO-style content


def make_cvo(out_dir, n_samples=2, size=128, db_name="cvo_test.lmdb"):
    import torch
    from accflow_amd.data.synthetic import gt_flow, make_sequence
    H = W = size
    items, truth = {}, []
    fflow = torch.cat([gt_flow(0, i, H, W) for i in range(2, 7)], 0).permute(1, 2, 0).numpy()   # (H, W, 10)
    bflow = torch.cat([gt_flow(i, 0, H, W) for i in range(2, 7)], 0).permute(1, 2, 0).numpy()

    def code(f):  # inverse of dataset.py:65-67
        return np.clip(np.round(f * 128.0) + 2 ** 15, 0, 65535).astype(np.uint16)

    all_keys = []
    for s in range(n_samples):
        frames = make_sequence(7000 + s, 7, H, W, batch=1)                                       # 7 x (1,3,H,W) in [0,255]
        imgs = torch.cat([f[0] for f in frames], 0).permute(1, 2, 0).round().clamp(0, 255).numpy().astype(np.uint8)
        blur = (imgs.astype(np.float32) * 0.5 + np.roll(imgs, 1, axis=1).astype(np.float32) * 0.5).round().astype(np.uint8)
        rec = {"imgs": imgs, "imgs_blur": blur, "fflows": code(fflow), "bflows": code(bflow)}
        for k, v in rec.items():
            key = "{:05d}_{:s}".format(s, k)
            items[key.encode()] = legacy_serialize(v)
            all_keys.append(key)
        truth.append({"imgs": imgs, "imgs_blur": blur, "fflows": (code(fflow).astype(np.float32) - 2 ** 15) / 128.0,
                      "bflows": (code(bflow).astype(np.float32) - 2 ** 15) / 128.0})
    items[b"__samples__"] = legacy_serialize(list(range(n_samples)))
    items[b"__valid_keys__"] = legacy_serialize(["imgs", "imgs_blur", "fflows", "bflows"])
    items[b"__keys__"] = legacy_serialize(all_keys)
    path = os.path.join(out_dir, db_name)
    write_lmdb(path, items)
    return path, truth


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--samples", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    a = ap.parse_args()
    p, _ = make_cvo(a.out_dir, a.samples, a.size)
    print(p, os.path.getsize(os.path.join(p, "data.mdb")), "bytes")
