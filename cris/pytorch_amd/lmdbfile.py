"""Read-only reader of an LMDB environment file (`data.mdb`), from scratch - the container the reference's datasets live in
(`utils/dataset.py:112-133`: `lmdb.open(path, subdir=isdir(path), readonly=True, lock=False)`, `txn.get(b'__len__')`,
`txn.get(b'__keys__')`, `txn.get(keys[index])`; written by `tools/folder2lmdb.py:36-68`).  The `lmdb` package is a C extension
that is not available on the training image; the file format is simple enough to walk directly:

    page 0 / 1   meta pages (the one with the larger transaction id is current): magic 0xBEEFC0DE, data version 1, page size,
                 the two core databases (free list, MAIN) each as {pad, flags, depth, branch / leaf / overflow pages, entries, root}
    branch page  header (16 B: pgno, pad, flags, lower, upper) + uint16 node offsets; node = {lo, hi, flags, ksize, key}: the child
                 page number is lo | hi << 16 | flags << 32; the first node's key is empty (= minus infinity)
    leaf page    same layout; node = {lo, hi, flags, ksize, key, data}: data size = lo | hi << 16; with F_BIGDATA the data area
                 holds the number of an overflow page and the value lies contiguously from byte 16 of that page
    keys         ordered by memcmp, shorter key first on a tie (the default comparison; the reference uses no integer keys)

Layout follows LMDB 0.9.x's mdb.c structure definitions (MDB_page, MDB_node, MDB_meta, MDB_db) for a 64-bit little-endian writer,
which is what py-lmdb on x86-64 / aarch64 produces.  PARITY UNPINNED: neither the `lmdb` library nor an LMDB file exists offline,
so the reader is exercised against a writer restated from the same definitions (tests/lmdb_write.py) - it has not yet opened a
file written by the real library.  Not supported (and not used by the reference): named sub-databases, DUPSORT, integer keys,
big-endian or 32-bit files; each is detected and refused.

Everything is host-side Python over an mmap: a lookup touches `depth` pages (3 for a 100 k-record dataset) and returns a
memoryview-backed bytes copy of the value.
"""
import mmap
import os
import struct
from typing import Iterator, List, Optional, Tuple

MAGIC = 0xBEEFC0DE
DATA_VERSION = 1
PAGEHDRSZ = 16
NODESIZE = 8
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2, P_SUBP = 0x01, 0x02, 0x04, 0x08, 0x20, 0x40
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
P_INVALID = (1 << 64) - 1
# MDB_db flags that change the key order or the node layout
MDB_REVERSEKEY, MDB_DUPSORT, MDB_INTEGERKEY = 0x02, 0x04, 0x08

_META = struct.Struct("<IIQQ")                 # magic, version, address, mapsize
_DB = struct.Struct("<IHHQQQQQ")               # pad, flags, depth, branch, leaf, overflow pages, entries, root
_HDR = struct.Struct("<QHHHH")                 # pgno, pad, flags, lower, upper
_NODE = struct.Struct("<HHHH")                 # lo, hi, flags, ksize


class LmdbFormatError(ValueError):
    pass


def data_file(path: str) -> str:
    """`lmdb.open(path, subdir=os.path.isdir(path))`: a directory holds `data.mdb`, otherwise the path is the file"""
    return os.path.join(path, "data.mdb") if os.path.isdir(path) else path


class LmdbReader:
    """`with LmdbReader(path) as db: db.get(b'0')`; also `len(db)`, `key in db`, `db.items()` (key order), `db.keys()`."""

    def __init__(self, path: str):
        self.path = data_file(path)
        self._f = open(self.path, "rb")
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            self._f.close()
            raise LmdbFormatError("%s: too small for an LMDB environment" % self.path)
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            self._read_meta(size)
        except struct.error as e:                    # a header that runs past the end of a damaged file
            self.close()
            raise LmdbFormatError("%s: %s" % (self.path, e)) from None
        except Exception:
            self.close()
            raise

    # ---- meta ----------------------------------------------------------------------------------------------------------------
    def _meta_at(self, off: int):
        pgno, _, flags, _, _ = _HDR.unpack_from(self._m, off)
        magic, version, _, mapsize = _META.unpack_from(self._m, off + PAGEHDRSZ)
        if magic != MAGIC:
            raise LmdbFormatError("%s: bad magic %#x at byte %d (not an LMDB file, or a big-endian one)" % (self.path, magic, off))
        if version != DATA_VERSION:
            raise LmdbFormatError("%s: data version %d, this reader knows version %d" % (self.path, version, DATA_VERSION))
        if not flags & P_META:
            raise LmdbFormatError("%s: page at byte %d is not a meta page" % (self.path, off))
        o = off + PAGEHDRSZ + _META.size
        free_db = _DB.unpack_from(self._m, o)
        main_db = _DB.unpack_from(self._m, o + _DB.size)
        last_pg, txnid = struct.unpack_from("<QQ", self._m, o + 2 * _DB.size)
        return dict(psize=free_db[0], main=main_db, last_pg=last_pg, txnid=txnid, mapsize=mapsize)

    def _read_meta(self, size: int):
        m0 = self._meta_at(0)
        psize = m0["psize"]
        if psize < 512 or psize > 65536 or psize & (psize - 1):
            raise LmdbFormatError("%s: implausible page size %d" % (self.path, psize))
        if size < 2 * psize:
            raise LmdbFormatError("%s: truncated (no second meta page)" % self.path)
        m1 = self._meta_at(psize)
        meta = m1 if m1["txnid"] > m0["txnid"] else m0
        self.psize = psize
        self.txnid = meta["txnid"]
        _, flags, depth, nbranch, nleaf, novf, entries, root = meta["main"]
        if flags & (MDB_REVERSEKEY | MDB_DUPSORT | MDB_INTEGERKEY):
            raise LmdbFormatError("%s: main database flags %#x (reverse / duplicate / integer keys) are not supported" % (self.path, flags))
        self.depth, self.entries, self.root = depth, entries, root
        self.pages = dict(branch=nbranch, leaf=nleaf, overflow=novf, last=meta["last_pg"])
        if root != P_INVALID and (root + 1) * psize > size:
            raise LmdbFormatError("%s: root page %d lies beyond the end of the file (truncated copy?)" % (self.path, root))
        self._size = size

    # ---- pages ---------------------------------------------------------------------------------------------------------------
    def _page(self, pgno: int) -> Tuple[int, int, int]:
        """-> (byte offset, flags, number of nodes)"""
        off = pgno * self.psize
        if off + self.psize > self._size:
            raise LmdbFormatError("%s: page %d lies beyond the end of the file" % (self.path, pgno))
        got, _, flags, lower, _ = _HDR.unpack_from(self._m, off)
        if got != pgno:
            raise LmdbFormatError("%s: page %d carries page number %d" % (self.path, pgno, got))
        if lower < PAGEHDRSZ or lower > self.psize or (lower - PAGEHDRSZ) & 1:
            raise LmdbFormatError("%s: page %d has an implausible node-offset array (lower = %d)" % (self.path, pgno, lower))
        if flags & (P_LEAF2 | P_SUBP):
            raise LmdbFormatError("%s: page %d has flags %#x (fixed-size-key / sub-page layouts are not supported)" % (self.path, pgno, flags))
        return off, flags, (lower - PAGEHDRSZ) >> 1

    def _node(self, off: int, i: int):
        """node i of the page at byte `off` -> (lo, hi, flags, key offset, key size)"""
        ptr = struct.unpack_from("<H", self._m, off + PAGEHDRSZ + 2 * i)[0]
        if ptr < PAGEHDRSZ or ptr + NODESIZE > self.psize:
            raise LmdbFormatError("%s: node offset %d outside its page (corrupt file)" % (self.path, ptr))
        lo, hi, flags, ksize = _NODE.unpack_from(self._m, off + ptr)
        if ptr + NODESIZE + ksize > self.psize:
            raise LmdbFormatError("%s: key of %d bytes runs past its page (corrupt file)" % (self.path, ksize))
        return lo, hi, flags, off + ptr + NODESIZE, ksize

    def _key(self, off: int, i: int) -> bytes:
        _, _, _, ko, ks = self._node(off, i)
        return self._m[ko:ko + ks]

    def _value(self, off: int, i: int) -> bytes:
        lo, hi, flags, ko, ks = self._node(off, i)
        if flags & (F_SUBDATA | F_DUPDATA):
            raise LmdbFormatError("%s: node with flags %#x (sub-database / duplicates) is not supported" % (self.path, flags))
        dsize = lo | (hi << 16)
        if flags & F_BIGDATA:
            if (ko + ks - off) + 8 > self.psize:
                raise LmdbFormatError("%s: overflow reference runs past its page (corrupt file)" % self.path)
            ovf = struct.unpack_from("<Q", self._m, ko + ks)[0]
            o = ovf * self.psize
            if o + PAGEHDRSZ > self._size:
                raise LmdbFormatError("%s: overflow page %d lies beyond the end of the file" % (self.path, ovf))
            got, _, pflags, npages = struct.unpack_from("<QHHI", self._m, o)
            if got != ovf or not pflags & P_OVERFLOW or PAGEHDRSZ + dsize > npages * self.psize or o + PAGEHDRSZ + dsize > self._size:
                raise LmdbFormatError("%s: bad overflow page %d for a %d-byte value" % (self.path, ovf, dsize))
            return self._m[o + PAGEHDRSZ:o + PAGEHDRSZ + dsize]
        if (ko + ks - off) + dsize > self.psize:
            raise LmdbFormatError("%s: inline value of %d bytes runs past its page (corrupt file)" % (self.path, dsize))
        return self._m[ko + ks:ko + ks + dsize]

    @staticmethod
    def _child(lo: int, hi: int, flags: int) -> int:
        return lo | (hi << 16) | (flags << 32)

    # ---- lookups -------------------------------------------------------------------------------------------------------------
    def _leaf_for(self, key: bytes) -> Optional[Tuple[int, int]]:
        """the leaf page that would hold `key` -> (byte offset, number of nodes)"""
        if self.root == P_INVALID:
            return None
        pgno = self.root
        for _ in range(64):                        # far deeper than any real tree: a cycle in a corrupt file must not hang
            off, flags, n = self._page(pgno)
            if flags & P_LEAF:
                return off, n
            if not flags & P_BRANCH or n < 1:
                raise LmdbFormatError("%s: page %d is neither branch nor leaf" % (self.path, pgno))
            # node 0 stands for everything below node 1's key: the last node i >= 1 with key_i <= key, else 0
            a, b = 1, n                            # invariant: keys of nodes [1, a) are <= key, of [b, n) are > key
            while a < b:
                mid = (a + b) >> 1
                if self._key(off, mid) <= key:     # bytes compare = memcmp, shorter first on a tie: LMDB's default order
                    a = mid + 1
                else:
                    b = mid
            lo, hi, nflags, _, _ = self._node(off, a - 1)
            pgno = self._child(lo, hi, nflags)
        raise LmdbFormatError("%s: tree deeper than 64 levels (corrupt file)" % self.path)

    def get(self, key: bytes, default=None):
        """`txn.get(key)`"""
        key = bytes(key)
        leaf = self._leaf_for(key)
        if leaf is None:
            return default
        off, n = leaf
        a, b = 0, n
        while a < b:
            mid = (a + b) >> 1
            k = self._key(off, mid)
            if k == key:
                return self._value(off, mid)
            if k < key:
                a = mid + 1
            else:
                b = mid
        return default

    def __contains__(self, key) -> bool:
        return self.get(key) is not None

    def __len__(self) -> int:
        return self.entries

    def _walk(self, pgno: int, depth: int, seen: Optional[set] = None) -> Iterator[Tuple[int, int]]:
        seen = set() if seen is None else seen
        if depth > 64 or pgno in seen:               # a tree visits every page once: a repeat is a cycle in a corrupt file
            raise LmdbFormatError("%s: page %d reached twice or below level 64 (corrupt file)" % (self.path, pgno))
        seen.add(pgno)
        off, flags, n = self._page(pgno)
        if flags & P_LEAF:
            for i in range(n):
                yield off, i
        elif flags & P_BRANCH:
            for i in range(n):
                lo, hi, nflags, _, _ = self._node(off, i)
                yield from self._walk(self._child(lo, hi, nflags), depth + 1, seen)
        else:
            raise LmdbFormatError("%s: page %d is neither branch nor leaf" % (self.path, pgno))

    def items(self) -> Iterator[Tuple[bytes, bytes]]:
        """(key, value) pairs in key order (a cursor's `iternext`)"""
        if self.root != P_INVALID:
            for off, i in self._walk(self.root, 0):
                yield self._key(off, i), self._value(off, i)

    def keys(self) -> List[bytes]:
        return [self._key(off, i) for off, i in self._walk(self.root, 0)] if self.root != P_INVALID else []

    def stat(self) -> dict:
        """`env.stat()`"""
        return dict(psize=self.psize, depth=self.depth, branch_pages=self.pages["branch"], leaf_pages=self.pages["leaf"],
                    overflow_pages=self.pages["overflow"], entries=self.entries)

    # ---- lifetime ------------------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_m", None) is not None:
            self._m.close()
            self._m = None
        if getattr(self, "_f", None) is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
