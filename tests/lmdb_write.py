"""TEST INFRASTRUCTURE ONLY - a minimal bulk writer of LMDB environment files, restated from LMDB 0.9.x's mdb.c structure
definitions (64-bit little-endian): meta pages, leaf / branch pages with their node-offset arrays, overflow pages for values
that do not fit a node (the library's me_nodemax rule).  It exists because neither the `lmdb` package nor any LMDB file is
available offline; cris/pytorch_amd/lmdbfile.py is exercised against the files this writes.  One transaction, keys sorted by
memcmp, pages filled front to back like a sequence of appends."""
import struct

PAGEHDRSZ, NODESIZE = 16, 8
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
P_INVALID = (1 << 64) - 1
MAGIC, DATA_VERSION = 0xBEEFC0DE, 1


def _even(n):
    return (n + 1) & ~1


class _Page:
    def __init__(self, psize, flags):
        self.psize, self.flags = psize, flags
        self.nodes = []                         # bytes of each node, key order
        self.used = 0

    def fits(self, node):
        return PAGEHDRSZ + 2 * (len(self.nodes) + 1) + self.used + _even(len(node)) <= self.psize

    def add(self, node):
        self.nodes.append(node)
        self.used += _even(len(node))

    def render(self, pgno):
        buf = bytearray(self.psize)
        upper = self.psize
        ptrs = []
        for node in self.nodes:                 # nodes are placed from the top of the page downwards (mdb_node_add)
            upper -= _even(len(node))
            buf[upper:upper + len(node)] = node
            ptrs.append(upper)
        lower = PAGEHDRSZ + 2 * len(ptrs)
        assert lower <= upper
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, self.flags, lower, upper)
        struct.pack_into("<%dH" % len(ptrs), buf, PAGEHDRSZ, *ptrs)
        return bytes(buf)


def _meta(psize, pgno, txnid, main, last_pg, main_flags=0):
    depth, nbranch, nleaf, novf, entries, root = main
    buf = bytearray(psize)
    struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
    o = PAGEHDRSZ
    struct.pack_into("<IIQQ", buf, o, MAGIC, DATA_VERSION, 0, 1 << 30)
    o += 24
    struct.pack_into("<IHHQQQQQ", buf, o, psize, 0, 0, 0, 0, 0, 0, P_INVALID)           # FREE_DBI: md_pad carries the page size
    o += 48
    struct.pack_into("<IHHQQQQQ", buf, o, 0, main_flags, depth, nbranch, nleaf, novf, entries, root)
    o += 48
    struct.pack_into("<QQ", buf, o, last_pg, txnid)
    return bytes(buf)


def build_tree(items, psize, first_pgno):
    """items: {key: value} -> (pages {pgno: bytes}, (depth, branch, leaf, overflow pages, entries, root), next free pgno)"""
    pages = {}
    nxt = first_pgno
    nodemax = (((psize - PAGEHDRSZ) // 2) & ~1) - 2
    keys = sorted(items)                         # bytes order == memcmp, shorter first on a tie
    if not keys:
        return pages, (0, 0, 0, 0, 0, P_INVALID), nxt
    novf = 0
    level = []                                   # (first key, pgno) of the pages of the current level
    cur = _Page(psize, P_LEAF)
    first = None

    def flush(page, first_key):
        nonlocal nxt
        pgno = nxt
        nxt += 1
        pages[pgno] = page
        level.append((first_key, pgno))

    for k in keys:
        v = items[k]
        if NODESIZE + len(k) + len(v) > nodemax:
            npages = (PAGEHDRSZ - 1 + len(v)) // psize + 1
            ovf = nxt
            nxt += npages
            buf = bytearray(npages * psize)
            struct.pack_into("<QHHI", buf, 0, ovf, 0, P_OVERFLOW, npages)
            buf[PAGEHDRSZ:PAGEHDRSZ + len(v)] = v
            pages[ovf] = bytes(buf)
            novf += npages
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", ovf)
        else:
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        if not cur.fits(node):
            flush(cur, first)
            cur, first = _Page(psize, P_LEAF), None
        if first is None:
            first = k
        cur.add(node)
    flush(cur, first)
    nleaf, nbranch, depth = len(level), 0, 1
    while len(level) > 1:                        # branch levels, bottom up; node 0 of a branch page has an empty key
        below, level = level, []
        cur, first = _Page(psize, P_BRANCH), None
        for k, pgno in below:
            key = b"" if not cur.nodes else k
            node = struct.pack("<HHHH", pgno & 0xFFFF, (pgno >> 16) & 0xFFFF, pgno >> 32, len(key)) + key
            if not cur.fits(node):
                flush(cur, first)
                cur, first = _Page(psize, P_BRANCH), None
                node = struct.pack("<HHHH", pgno & 0xFFFF, (pgno >> 16) & 0xFFFF, pgno >> 32, 0)
            if first is None:
                first = k
            cur.add(node)
        flush(cur, first)
        nbranch += len(level)
        depth += 1
    root = level[0][1]
    rendered = {pg: (p.render(pg) if isinstance(p, _Page) else p) for pg, p in pages.items()}
    return rendered, (depth, nbranch, nleaf, novf, len(keys), root), nxt


def write_env(path, items, psize=4096, older=None, newer_first=False, main_flags=0):
    """write `items` as the current state of an environment; `older`: an earlier state kept in the other meta page (its pages
    stay in the file, as after a later write transaction); newer_first: the current meta goes to page 0 instead of page 1"""
    pages, main, nxt = build_tree(items, psize, 2)
    metas = {}
    if older is not None:
        old_pages, old_main, nxt = build_tree(older, psize, nxt)
        pages.update(old_pages)
    else:
        old_main = (0, 0, 0, 0, 0, P_INVALID)
    last = nxt - 1
    cur_slot = 0 if newer_first else 1
    metas[cur_slot] = _meta(psize, cur_slot, 7, main, last, main_flags)
    metas[1 - cur_slot] = _meta(psize, 1 - cur_slot, 6, old_main, last, main_flags)
    with open(path, "wb") as f:
        f.write(metas[0])
        f.write(metas[1])
        for pg in range(2, nxt):
            blob = pages.get(pg)
            if blob is None:
                continue                          # continuation of an overflow run (already written with its first page)
            f.seek(pg * psize)
            f.write(blob)
        f.truncate(nxt * psize)
    return main
