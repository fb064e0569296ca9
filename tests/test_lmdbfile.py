"""cris/pytorch_amd/lmdbfile.py (read-only LMDB container reader, reference utils/dataset.py:112-133) against files laid out by
tests/lmdb_write.py from LMDB 0.9.x's structure definitions.  Parity unpinned: no real LMDB library or file is available offline."""
import os
import pickle
import random
import struct

import pytest

from cris.pytorch_amd import lmdbfile
from cris.pytorch_amd.lmdbfile import LmdbFormatError, LmdbReader

from lmdb_write import write_env


def _rand_bytes(rng, n):
    return bytes(rng.getrandbits(8) for _ in range(n))


def test_small_environment_and_missing_keys(tmp_path):
    p = str(tmp_path / "a.mdb")
    items = {b"alpha": b"1", b"beta": b"", b"b": b"xyz" * 10}
    main = write_env(p, items)
    with LmdbReader(p) as db:
        assert len(db) == 3 and db.depth == 1 and db.stat()["leaf_pages"] == 1 == main[2]
        for k, v in items.items():
            assert db.get(k) == v and k in db
        assert db.get(b"beta") == b""                       # an empty value is a value
        assert db.get(b"gamma") is None and db.get(b"") is None and db.get(b"alph") is None and db.get(b"alphaa", 7) == 7
        assert list(db.items()) == sorted(items.items()) and db.keys() == sorted(items)


def test_empty_environment(tmp_path):
    p = str(tmp_path / "e.mdb")
    write_env(p, {})
    with LmdbReader(p) as db:
        assert len(db) == 0 and db.get(b"0") is None and list(db.items()) == []


def test_reference_dataset_layout_with_overflow_values(tmp_path):
    """the layout tools/folder2lmdb.py writes: ASCII decimal keys, pickled records whose image bytes take overflow pages,
    `__keys__` and `__len__`; read back the way RefDataset does"""
    rng = random.Random(0)
    n = 300
    recs = {}
    for i in range(n):
        recs[str(i).encode("ascii")] = pickle.dumps({"img": _rand_bytes(rng, rng.choice([900, 5000, 70000, 200000])), "mask": _rand_bytes(rng, 700),
                                                      "sents": ["the left one", "a dog"], "num_sents": 2, "seg_id": i}, protocol=5)
    keys = [str(i).encode("ascii") for i in range(n)]
    items = dict(recs)
    items[b"__keys__"] = pickle.dumps(keys, protocol=5)
    items[b"__len__"] = pickle.dumps(n, protocol=5)
    d = tmp_path / "train.lmdb"
    d.mkdir()                                            # lmdb.open(path, subdir=True): path/data.mdb
    main = write_env(str(d / "data.mdb"), items)
    assert main[3] > 0 and main[0] >= 2                  # overflow pages and at least one branch level
    with LmdbReader(str(d)) as db:
        assert pickle.loads(db.get(b"__len__")) == n
        got_keys = pickle.loads(db.get(b"__keys__"))
        assert got_keys == keys
        for i in rng.sample(range(n), 60):
            assert db.get(got_keys[i]) == recs[got_keys[i]]
        assert dict(db.items()) == items
        assert db.stat()["overflow_pages"] == main[3] and db.stat()["entries"] == n + 2


def test_three_level_tree_and_key_order(tmp_path):
    """60 000 short records: leaf pages -> branch pages -> root; lookups of present and absent keys at every boundary"""
    rng = random.Random(1)
    items = {("%d" % i).encode(): struct.pack("<I", i) + b"v" * (i % 7) for i in range(60000)}
    p = str(tmp_path / "big.mdb")
    main = write_env(p, items)
    assert main[0] == 3
    with LmdbReader(p) as db:
        assert db.depth == 3 and len(db) == 60000
        for i in rng.sample(range(60000), 2000):
            k = ("%d" % i).encode()
            assert db.get(k) == items[k]
        for k in (b"", b"/", b"0", b"00", b"59999", b"6", b"60000", b"9999", b"99999", b":", b"1\x00"):
            assert db.get(k) == items.get(k)
        it = db.items()
        prev = None
        count = 0
        for k, v in it:
            assert prev is None or prev < k
            assert items[k] == v
            prev = k
            count += 1
        assert count == 60000


@pytest.mark.parametrize("seed", range(6))
def test_random_keys_including_prefixes(tmp_path, seed):
    """random binary keys of 1-40 bytes, many of them prefixes of each other (memcmp order with the shorter key first), values of
    0-6000 bytes around the node / overflow limit, 512- to 16384-byte pages"""
    rng = random.Random(seed)
    psize = rng.choice([512, 1024, 4096, 16384])
    items = {}
    stems = [_rand_bytes(rng, rng.randint(1, 20)) for _ in range(30)]
    for _ in range(rng.randint(50, 1500)):
        k = rng.choice(stems)[:rng.randint(1, 20)] + _rand_bytes(rng, rng.randint(0, 20))
        items[k] = _rand_bytes(rng, rng.choice([0, 1, 7, 100, psize // 2 - 40, psize // 2, 3000, 6000]))
    p = str(tmp_path / "r.mdb")
    write_env(p, items, psize=psize)
    with LmdbReader(p) as db:
        assert db.psize == psize and len(db) == len(items)
        for k, v in items.items():
            assert db.get(k) == v
        for k in list(items)[:200]:
            for probe in (k + b"\x00", k[:-1], k[:-1] + bytes([(k[-1] + 1) & 255])):
                assert db.get(probe) == items.get(probe)
        assert db.keys() == sorted(items)


def test_the_meta_page_with_the_larger_transaction_id_is_current(tmp_path):
    new, old = {b"k": b"new", b"only-new": b"1"}, {b"k": b"old", b"only-old": b"2"}
    for newer_first in (False, True):
        p = str(tmp_path / ("m%d.mdb" % newer_first))
        write_env(p, new, older=old, newer_first=newer_first)
        with LmdbReader(p) as db:
            assert db.txnid == 7 and db.get(b"k") == b"new" and db.get(b"only-old") is None and db.get(b"only-new") == b"1"


def test_refuses_what_it_does_not_read(tmp_path):
    p = str(tmp_path / "x.mdb")
    write_env(p, {b"a": b"b"})
    blob = bytearray(open(p, "rb").read())
    bad = str(tmp_path / "bad.mdb")
    open(bad, "wb").write(b"\x00" * 16 + b"\xde\xc0\xef\xbe"[::-1] + bytes(blob[20:]))          # byte-swapped magic
    with pytest.raises(LmdbFormatError, match="magic"):
        LmdbReader(bad)
    open(bad, "wb").write(bytes(blob[:4096]))                                                   # second meta page missing
    with pytest.raises(LmdbFormatError, match="truncated"):
        LmdbReader(bad)
    open(bad, "wb").write(bytes(blob[:8192]))                                                   # the data pages cut off
    with pytest.raises(LmdbFormatError, match="beyond the end"):
        LmdbReader(bad)
    write_env(bad, {b"a": b"b"}, main_flags=0x04)                                               # MDB_DUPSORT
    with pytest.raises(LmdbFormatError, match="not supported"):
        LmdbReader(bad)
    v2 = bytearray(blob)
    struct.pack_into("<I", v2, 16 + 4, 2)
    struct.pack_into("<I", v2, 4096 + 16 + 4, 2)
    open(bad, "wb").write(bytes(v2))
    with pytest.raises(LmdbFormatError, match="version"):
        LmdbReader(bad)
    with pytest.raises(LmdbFormatError, match="too small"):
        open(bad, "wb").write(b"abc")
        LmdbReader(bad)
    assert lmdbfile.data_file(str(tmp_path)) == os.path.join(str(tmp_path), "data.mdb") and lmdbfile.data_file(p) == p


def test_lmdb_records_follow_the_reference_dataset_protocol(tmp_path):
    """records.LmdbRecords: __len__ / __keys__ / per-index record dicts, lazily opened, picklable (worker processes reopen)"""
    from cris.pytorch_amd.records import LmdbRecords
    rng = random.Random(3)
    n = 40
    recs = [{"img": _rand_bytes(rng, 30000), "mask": _rand_bytes(rng, 500), "cat": 3, "seg_id": 100 + i, "img_name": "x%d.jpg" % i,
             "num_sents": 1 + i % 3, "sents": ["sentence %d %d" % (i, j) for j in range(1 + i % 3)]} for i in range(n)]
    keys = [str(i).encode("ascii") for i in range(n)]
    items = {k: pickle.dumps(r, protocol=5) for k, r in zip(keys, recs)}
    items[b"__keys__"] = pickle.dumps(keys, protocol=5)
    items[b"__len__"] = pickle.dumps(n, protocol=5)
    p = str(tmp_path / "val.lmdb")
    write_env(p, items)                                      # subdir=False: the path is the file itself
    ds = LmdbRecords(p)
    assert ds._db is None and len(ds) == n
    assert ds[0] == recs[0] and ds[n - 1] == recs[n - 1] and ds.batch([5, 7]) == [recs[5], recs[7]]
    ds2 = pickle.loads(pickle.dumps(ds))
    assert ds2._db is None and ds2[11] == recs[11]
    ds.close()
    ds2.close()
    bad = str(tmp_path / "nokeys.lmdb")
    write_env(bad, {b"0": items[b"0"]})
    with pytest.raises(ValueError, match="__len__"):
        len(LmdbRecords(bad))


@pytest.mark.parametrize("seed", range(8))
def test_damaged_files_are_refused_or_read_never_hung_or_crashed(tmp_path, seed):
    """random byte damage in the meta, branch and leaf pages: opening, looking up every key and walking the tree either works
    or raises LmdbFormatError - no other exception type, no endless walk (a page reached twice is a cycle)"""
    rng = random.Random(100 + seed)
    items = {("%d" % i).encode(): _rand_bytes(rng, rng.choice([3, 40, 3000])) for i in range(400)}
    p = str(tmp_path / "ok.mdb")
    write_env(p, items, psize=512)
    blob = bytearray(open(p, "rb").read())
    npages = len(blob) // 512
    outcomes = set()
    for trial in range(60):
        bad = bytearray(blob)
        for _ in range(rng.randint(1, 6)):
            page = rng.choice([0, 1, rng.randrange(npages), npages - 1 - rng.randrange(min(40, npages))])
            pos = page * 512 + (rng.randrange(16) if rng.random() < 0.5 else rng.randrange(512))
            bad[pos] = rng.getrandbits(8)
        q = str(tmp_path / "bad.mdb")
        open(q, "wb").write(bytes(bad))
        try:
            with LmdbReader(q) as db:
                for k in list(items)[:50]:
                    db.get(k)
                n = 0
                for _ in db.items():
                    n += 1
                    assert n <= 100000
            outcomes.add("read")
        except LmdbFormatError:
            outcomes.add("refused")
    assert outcomes <= {"read", "refused"} and outcomes
