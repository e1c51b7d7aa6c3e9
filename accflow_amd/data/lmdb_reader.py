"""Read-only LMDB environment in pure Python (mmap + B+tree walk).

The reference opens `cvo_test.lmdb` with the `lmdb` C extension (data/dataset.py:36-43: readonly, lock=False) and
only ever calls `txn.get(key)` on the main database (:45, :64).  `lmdb` is not installable offline, so this module
implements exactly that operation on the on-disk format of LMDB 0.9 (`data.mdb`, format version 1; the format is
stable across py-lmdb 0.9x-1.4, the pinned 1.4.1 included - environment.yml:187):

  page    = 16-byte header {pgno u64, pad u16, flags u16, lower u16, upper u16 | overflow: pages u32 in lower/upper}
            + u16 node offsets mp_ptrs[(lower - 16) / 2] + nodes growing down from `upper`
  meta    = pages 0 and 1: header + {magic 0xBEEFC0DE, version 1, address, mapsize, dbs[2] x 48 B {pad (page size in
            dbs[0]), flags u16, depth u16, branch_pages, leaf_pages, overflow_pages, entries, root}, last_pg, txnid};
            the meta page with the larger txnid is current; dbs[1] is the main database
  node    = {lo u16, hi u16, flags u16, ksize u16, key bytes, data}: leaf data size = lo | hi << 16, data = inline
            bytes, or with F_BIGDATA (0x01) the u64 page number of an overflow run whose payload starts 16 bytes in;
            branch child page number = lo | hi << 16 | flags << 32, first key of a branch page is the implicit -inf
  order   = keys compare as byte strings (memcmp, shorter first on ties) - the default comparator

Not supported (and rejected loudly): named sub-databases, MDB_DUPSORT / integer-key databases, write transactions.
"""
import mmap
import os
import struct

P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
PAGEHDRSZ = 16
MDB_MAGIC = 0xBEEFC0DE
P_INVALID = 0xFFFFFFFFFFFFFFFF


class LMDBError(RuntimeError):
    pass


class ReadOnlyLMDB:
    """env = ReadOnlyLMDB(path); env.get(b"key") -> bytes | None; env.keys() iterates in key order; len(env)."""

    def __init__(self, path):
        if os.path.isdir(path):
            path = os.path.join(path, "data.mdb")
        if not os.path.isfile(path):
            raise LMDBError("LMDB data file not found: %s" % path)
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        if len(self._mm) < PAGEHDRSZ + 24 + 96 + 16:
            raise LMDBError("%s: too short for an LMDB meta page" % path)
        # Meta page 0 starts at file offset 0 and carries the page size (dbs[0].pad; mdb_env_read_header reads it there
        # before it can locate meta page 1).  The two meta pages are written alternately, one per commit, without a
        # checksum: mdb_env_pick_meta takes the larger txnid.  A copy taken while a writer was committing (or a crash
        # between the data sync and the meta sync) can leave that newer meta torn, so each candidate is validated against
        # the file - magic / version, last page inside the file, root inside the used pages and really a tree page - and
        # the older meta is used when the newer one does not hold up (what `mdb_copy` / MDB_PREVSNAPSHOT recover to).
        psize = self._probe_psize()
        if psize < 512 or psize > 32768 or psize & (psize - 1):   # mdb.c: MAX_PAGESIZE 0x8000
            raise LMDBError("%s: implausible page size %d in meta page 0" % (path, psize))
        self.psize = psize
        metas, why = [], []
        for pg in (0, 1):
            off = pg * psize + PAGEHDRSZ
            if off + 24 + 96 + 16 > len(self._mm):
                why.append("meta %d: beyond the end of the file" % pg)
                continue
            magic, version = struct.unpack_from("<II", self._mm, off)
            if magic != MDB_MAGIC:
                if pg == 0:
                    raise LMDBError("%s: bad LMDB magic 0x%08x in meta page 0" % (path, magic))
                why.append("meta %d: bad magic 0x%08x" % (pg, magic))
                continue
            if version != 1:
                raise LMDBError("%s: LMDB data format version %d is not supported (expected 1)" % (path, version))
            dbs = [struct.unpack_from("<IHHQQQQQ", self._mm, off + 24 + 48 * i) for i in range(2)]
            last_pg, txnid = struct.unpack_from("<QQ", self._mm, off + 24 + 96)
            bad = self._meta_problem(dbs, last_pg)
            if bad:
                why.append("meta %d (txnid %d): %s" % (pg, txnid, bad))
                continue
            metas.append((txnid, dbs, last_pg))
        if not metas:
            raise LMDBError("%s: no usable meta page (%s)" % (path, "; ".join(why)))
        txnid, dbs, last_pg = max(metas, key=lambda m: m[0])
        self.txnid, self.skipped_metas = txnid, why
        _, flags, self.depth, _, _, _, self.entries, self.root = dbs[1]
        if flags:  # MDB_REVERSEKEY / DUPSORT / INTEGERKEY / ...: the CVO files use the default byte-string keys
            raise LMDBError("%s: main database flags 0x%x (dupsort / integer / reverse keys) are not supported" % (path, flags))
        self.last_pg = last_pg
        # dbs[0] is the free-list database (txnid -> page numbers released by that transaction): pages it lists hold
        # stale content and are simply never reached from the main tree's root, so a reader has nothing to do with it
        self.free_db_entries = dbs[0][6]

    def _meta_problem(self, dbs, last_pg):
        """Why a meta page cannot be the current snapshot, or None."""
        npages = len(self._mm) // self.psize
        if last_pg < 1 or last_pg >= npages:
            return "last page %d outside the file (%d pages)" % (last_pg, npages)
        for name, db in (("free", dbs[0]), ("main", dbs[1])):
            root, depth, entries = db[7], db[2], db[6]
            if root == P_INVALID:
                if entries or depth:
                    return "%s database: no root but %d entries" % (name, entries)
                continue
            if root < 2 or root > last_pg:
                return "%s database: root page %d outside the used pages [2, %d]" % (name, root, last_pg)
            pgno, _, pflags = struct.unpack_from("<QHH", self._mm, root * self.psize)
            if pgno != root or not pflags & (P_BRANCH | P_LEAF) or pflags & (P_OVERFLOW | P_META):
                return "%s database: root page %d is not a tree page (header pgno %d, flags 0x%x)" % (name, root, pgno, pflags)
            if depth < 1 or depth > 64:
                return "%s database: depth %d" % (name, depth)
        return None

    def _probe_psize(self):
        # the page size lives in meta page 0 itself (dbs[0].pad), which always starts at file offset 0
        return struct.unpack_from("<I", self._mm, PAGEHDRSZ + 24)[0]

    def close(self):
        self._mm.close()
        self._f.close()

    def __len__(self):
        return self.entries

    # ---- pages / nodes ---------------------------------------------------------------------------
    def _page(self, pgno):
        if pgno > self.last_pg:
            raise LMDBError("page %d beyond the last used page %d" % (pgno, self.last_pg))
        off = pgno * self.psize
        own, _, flags, lower, upper = struct.unpack_from("<QHHHH", self._mm, off)
        if own != pgno or lower < PAGEHDRSZ or lower > upper or upper > self.psize:
            raise LMDBError("page %d: corrupt header (pgno %d, lower %d, upper %d)" % (pgno, own, lower, upper))
        return off, flags, (lower - PAGEHDRSZ) // 2

    def _node(self, poff, i):
        noff = poff + struct.unpack_from("<H", self._mm, poff + PAGEHDRSZ + 2 * i)[0]
        lo, hi, flags, ksize = struct.unpack_from("<HHHH", self._mm, noff)
        return noff, lo, hi, flags, ksize

    def _key(self, noff, ksize):
        return self._mm[noff + 8:noff + 8 + ksize]

    def _leaf_value(self, noff, lo, hi, flags, ksize):
        if flags & (F_SUBDATA | F_DUPDATA):
            raise LMDBError("sub-database / duplicate records are not supported")
        size = lo | (hi << 16)
        doff = noff + 8 + ksize
        if flags & F_BIGDATA:
            # the value lives in a run of mp_pages CONTIGUOUS pages (one header in the first page only): payload =
            # the `size` bytes after that header, crossing page boundaries without further headers (mdb.c: OVPAGES)
            pgno = struct.unpack_from("<Q", self._mm, doff)[0]
            if pgno < 2 or pgno > self.last_pg:
                raise LMDBError("F_BIGDATA node points at page %d outside the used pages" % pgno)
            ooff = pgno * self.psize
            opg, _, oflags, npages = struct.unpack_from("<QHHI", self._mm, ooff)
            if not oflags & P_OVERFLOW or opg != pgno:
                raise LMDBError("F_BIGDATA node does not point at an overflow page")
            need = (PAGEHDRSZ - 1 + size) // self.psize + 1
            if npages < need or pgno + npages - 1 > self.last_pg:
                raise LMDBError("overflow run at page %d: %d pages for a %d-byte value (needs %d; last page %d)"
                                % (pgno, npages, size, need, self.last_pg))
            doff = ooff + PAGEHDRSZ
        elif doff + size > (noff // self.psize + 1) * self.psize:
            raise LMDBError("inline value crosses its page: corrupt node")
        return self._mm[doff:doff + size]

    # ---- lookup ------------------------------------------------------------------------------------
    def get(self, key, default=None):
        if isinstance(key, str):
            key = key.encode()
        if self.root == P_INVALID:
            return default
        pgno = self.root
        for _ in range(64):
            poff, flags, n = self._page(pgno)
            if flags & P_LEAF2:
                raise LMDBError("LEAF2 (fixed-size dup) pages are not supported")
            if flags & P_BRANCH:
                # largest i with key_i <= key; node 0's key is the implicit minimum
                lo_i, hi_i = 0, n - 1
                while lo_i < hi_i:
                    mid = (lo_i + hi_i + 1) // 2
                    noff, _, _, _, ksize = self._node(poff, mid)
                    if self._key(noff, ksize) <= key:
                        lo_i = mid
                    else:
                        hi_i = mid - 1
                noff, lo, hi, nflags, _ = self._node(poff, lo_i)
                pgno = lo | (hi << 16) | (nflags << 32)
                continue
            if not flags & P_LEAF:
                raise LMDBError("unexpected page flags 0x%x in the tree" % flags)
            lo_i, hi_i = 0, n - 1
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) // 2
                noff, lo, hi, nflags, ksize = self._node(poff, mid)
                k = self._key(noff, ksize)
                if k == key:
                    return self._leaf_value(noff, lo, hi, nflags, ksize)
                if k < key:
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            return default
        raise LMDBError("tree deeper than 64 levels: corrupt file")

    def items(self):
        """(key, value) pairs in key order (depth-first walk)."""
        if self.root == P_INVALID:
            return
        stack = [self.root]
        while stack:
            pgno = stack.pop()
            poff, flags, n = self._page(pgno)
            if flags & P_BRANCH:
                children = []
                for i in range(n):
                    _, lo, hi, nflags, _ = self._node(poff, i)
                    children.append(lo | (hi << 16) | (nflags << 32))
                stack.extend(reversed(children))
            else:
                for i in range(n):
                    noff, lo, hi, nflags, ksize = self._node(poff, i)
                    yield self._key(noff, ksize), self._leaf_value(noff, lo, hi, nflags, ksize)

    def keys(self):
        for k, _ in self.items():
            yield k
