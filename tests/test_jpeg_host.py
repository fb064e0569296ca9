"""JPEG decoding without a GPU (reference call site utils/dataset.py:127-129):
  1. the oracle (oracle/jpeg_baseline.py) is PINNED to libjpeg-turbo as built into Pillow - bit exact on every case;
  2. the library's host half (marker parsing + Huffman decoding, csrc/jpeg.hip) gives the oracle's coefficients bit for bit;
  3. the arithmetic of the device half (csrc/jpeg_core.h - the header the HIP kernels include) compiled by g++ into a probe
     gives Pillow's pixels bit for bit;
  4. unsupported files are refused with a message, corrupt ones do not crash.
The kernels themselves are compared with the same references on the GPU (tests/test_jpeg_gpu.py)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT

from cris.pytorch_amd import hip, jpegdec  # noqa: E402
from oracle import jpeg_baseline as J  # noqa: E402
import jpeg_cases  # noqa: E402


@pytest.fixture(scope="module")
def files():
    pytest.importorskip("PIL")                       # the encoder of the test files and the pinning decoder
    return list(jpeg_cases.cases(big=(120, 160)))


@pytest.fixture(scope="module")
def probe():
    d = tempfile.mkdtemp()
    exe = os.path.join(d, "jpeg_core_probe")
    subprocess.check_call(["g++", "-O1", "-o", exe, os.path.join(ROOT, "tests", "jpeg_core_probe.cpp")])
    return exe


def test_oracle_is_pinned_to_libjpeg_turbo(files):
    for name, data in files:
        assert np.array_equal(J.decode(data), jpeg_cases.pil_decode(data)), name


def test_oracle_and_library_reproduce_the_committed_vectors(probe):
    """tests/golden/jpeg/vectors.npz (tests/golden/make_jpeg_golden.py): files + the pixels Pillow's libjpeg-turbo gave for them
    when the fixture was made - needs no Pillow at test time"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "jpeg", "vectors.npz"))
    names = [str(n) for n in g["names"]]
    assert len(names) >= 10
    datas = [g["jpg%d" % i].tobytes() for i in range(len(names))]
    infos, coef, offs = jpegdec.decode_coefficients(datas, threads=2)
    with tempfile.TemporaryDirectory() as td:
        for i, name in enumerate(names):
            ref = g["rgb%d" % i]
            assert np.array_equal(J.decode(datas[i]), ref), name
            I = infos[i]
            open(os.path.join(td, "i.bin"), "wb").write(bytes(I))
            coef[offs[i]:offs[i] + I.coef_count].numpy().tofile(os.path.join(td, "c.bin"))
            subprocess.check_call([probe, os.path.join(td, "i.bin"), os.path.join(td, "c.bin"), os.path.join(td, "o.rgb")])
            assert np.array_equal(np.fromfile(os.path.join(td, "o.rgb"), dtype=np.uint8).reshape(ref.shape), ref), name


def test_host_half_matches_oracle_coefficients(files):
    datas = [d for _, d in files]
    infos, coef, offs = jpegdec.decode_coefficients(datas, threads=4)
    for i, (name, data) in enumerate(files):
        h = J.parse(data)
        ref = J.entropy_decode_general(data, h) if h.multiscan else J.entropy_decode(data, h)
        I = infos[i]
        assert (I.width, I.height, I.ncomp, I.hmax, I.vmax, I.restart_interval) == (h.width, h.height, len(h.comps), h.hmax, h.vmax, h.restart_interval), name
        for c, comp in enumerate(h.comps):
            assert (I.blocks_w[c], I.blocks_h[c], I.down_w[c], I.down_h[c]) == (comp["bw"], comp["bh"], comp["dw"], comp["dh"]), name
            assert np.array_equal(np.array(I.quant[c][:], dtype=np.int64), h.qt[comp["tq"]]), name
            n = comp["bw"] * comp["bh"] * 64
            got = coef[offs[i] + I.coef_offset[c]: offs[i] + I.coef_offset[c] + n].numpy().reshape(comp["bh"], comp["bw"], 64)
            assert np.array_equal(got, ref[c]), (name, c)
    # one thread gives the same
    _, coef1, _ = jpegdec.decode_coefficients(datas, threads=1)
    for i in range(len(datas)):
        a, b = offs[i], offs[i] + infos[i].coef_count
        assert bool((coef1[a:b] == coef[a:b]).all())


def test_device_arithmetic_on_the_cpu_matches_pillow(files, probe):
    datas = [d for _, d in files]
    infos, coef, offs = jpegdec.decode_coefficients(datas, threads=4)
    with tempfile.TemporaryDirectory() as td:
        for i, (name, data) in enumerate(files):
            I = infos[i]
            open(os.path.join(td, "i.bin"), "wb").write(bytes(I))
            coef[offs[i]:offs[i] + I.coef_count].numpy().tofile(os.path.join(td, "c.bin"))
            subprocess.check_call([probe, os.path.join(td, "i.bin"), os.path.join(td, "c.bin"), os.path.join(td, "o.rgb")])
            got = np.fromfile(os.path.join(td, "o.rgb"), dtype=np.uint8).reshape(I.height, I.width, 3)
            assert np.array_equal(got, jpeg_cases.pil_decode(data)), name


def test_unsupported_and_corrupt_files_are_refused():
    pytest.importorskip("PIL")
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (16, 16, 4), dtype=np.uint8), "CMYK").save(b, "JPEG", quality=80)
    with pytest.raises(hip.HipLibraryError, match="1 or 3 components"):
        jpegdec.read_header(b.getvalue())
    assert jpegdec.read_header(jpeg_cases.encode(img, quality=80, progressive=True)).multiscan == 2
    with pytest.raises(hip.HipLibraryError, match="not a JPEG"):
        jpegdec.read_header(b"\x89PNG\r\n\x1a\n" + bytes(64))
    good = jpeg_cases.encode(img, quality=80, subsampling=2)
    with pytest.raises(hip.HipLibraryError):
        jpegdec.read_header(good[:100])                      # cut inside the tables
    # a file cut inside the entropy-coded data decodes (missing data reads as zero bits, like libjpeg) or is refused - no crash
    for cut in (len(good) // 2, len(good) - 3):
        try:
            jpegdec.decode_coefficients([good[:cut]], threads=1)
        except hip.HipLibraryError:
            pass
    # random bytes after a valid header
    junk = good[:jpegdec.read_header(good).scan_offset] + bytes(rng.integers(0, 256, 400, dtype=np.uint8))
    try:
        jpegdec.decode_coefficients([junk], threads=1)
    except hip.HipLibraryError:
        pass
    # the batch call names the failing image
    with pytest.raises(hip.HipLibraryError, match="image 1"):
        jpegdec.decode_coefficients([good, b"\xff\xd8\xff\xd9"], threads=2)


def test_corrupted_files_never_crash_the_host_decoder():
    """6000 mutations (flipped bytes anywhere, truncation, damaged entropy-coded data, inserted bytes) of sequential and
    progressive files: every one is either decoded (libjpeg, too, decodes what it can of a damaged scan) or refused with an
    error - the process survives and no call hangs"""
    pytest.importorskip("PIL")
    rng = np.random.default_rng(7)
    seeds = [d for _, d in jpeg_cases.cases(sizes=((37, 53), (16, 16), (9, 4)), qualities=(85,))]
    decoded = refused = 0
    for it in range(6000):
        src = bytearray(seeds[it % len(seeds)])
        mode = it % 4
        if mode == 0:
            for _ in range(rng.integers(1, 6)):
                src[rng.integers(2, len(src))] = rng.integers(0, 256)
        elif mode == 1:
            src = src[:rng.integers(4, len(src))]
        elif mode == 2:
            off = jpegdec.read_header(bytes(src)).scan_offset
            for _ in range(rng.integers(1, 20)):
                src[rng.integers(off, len(src))] = rng.integers(0, 256)
        else:
            p = int(rng.integers(2, len(src)))
            src[p:p] = bytes(rng.integers(0, 256, rng.integers(1, 8), dtype=np.uint8))
        try:
            jpegdec.decode_coefficients([bytes(src)], threads=1)
            decoded += 1
        except hip.HipLibraryError:
            refused += 1
    assert decoded + refused == 6000 and decoded > 500 and refused > 500


def test_exif_orientation_is_read_and_applied_like_pillow():
    """cv2.imdecode turns the decoded array by the EXIF orientation (all eight values); the tag is found in both TIFF byte orders,
    absent / damaged EXIF means upright.  Reference for the turn: PIL.ImageOps.exif_transpose (the same table as OpenCV's)."""
    pytest.importorskip("PIL")
    import io
    import torch
    from PIL import Image, ImageOps
    rng = np.random.default_rng(3)
    img = jpeg_cases._smooth(rng, 24, 40)
    plain = jpeg_cases.encode(img, quality=90, subsampling=0)
    assert jpegdec.exif_orientation(plain) == 1
    base = torch.from_numpy(np.array(jpeg_cases.pil_decode(plain)))
    for o in range(1, 9):
        ex = Image.Exif()
        ex[0x0112] = o
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=0, exif=ex.tobytes())
        data = b.getvalue()
        assert jpegdec.exif_orientation(data) == o
        want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(data))).convert("RGB"))
        got = jpegdec.apply_orientation(base, o).contiguous().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), o
        # the other byte order: rewrite the TIFF block by hand
        i = data.index(b"Exif\x00\x00") + 6
        tiff = data[i:]
        big = tiff[:2] == b"MM"
        other = (b"II*\x00\x08\x00\x00\x00\x01\x00\x12\x01\x03\x00\x01\x00\x00\x00" + bytes([o, 0, 0, 0]) + b"\x00\x00\x00\x00") if big else \
                (b"MM\x00*\x00\x00\x00\x08\x00\x01\x01\x12\x00\x03\x00\x00\x00\x01" + bytes([0, o, 0, 0]) + b"\x00\x00\x00\x00")
        seg = b"Exif\x00\x00" + other
        swapped = data[:2] + b"\xff\xe1" + (len(seg) + 2).to_bytes(2, "big") + seg + plain[2:]
        assert jpegdec.exif_orientation(swapped) == o
    assert jpegdec.exif_orientation(plain[:2] + b"\xff\xe1\x00\x0aExif\x00\x00zz" + plain[2:]) == 1      # damaged block


def test_property_random_files_decode_like_pillow(probe):
    """hypothesis: random content, size, quality, sampling, sequential / progressive, optimised tables - host half + the device
    arithmetic (jpeg_core.h via the g++ probe) == Pillow's libjpeg-turbo"""
    pytest.importorskip("PIL")
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp

    @settings(max_examples=120, deadline=None, derandomize=True)
    @given(hnp.arrays(np.uint8, st.tuples(st.integers(1, 40), st.integers(1, 40), st.just(3))), st.integers(1, 100), st.sampled_from([0, 1, 2]),
           st.booleans(), st.booleans())
    def check(img, quality, sub, progressive, optimize):
        data = jpeg_cases.encode(img, quality=quality, subsampling=sub, progressive=progressive, optimize=optimize)
        infos, coef, offs = jpegdec.decode_coefficients([data], threads=1)
        with tempfile.TemporaryDirectory() as td:
            open(os.path.join(td, "i.bin"), "wb").write(bytes(infos[0]))
            coef[offs[0]:offs[0] + infos[0].coef_count].numpy().tofile(os.path.join(td, "c.bin"))
            subprocess.check_call([probe, os.path.join(td, "i.bin"), os.path.join(td, "c.bin"), os.path.join(td, "o.rgb")])
            got = np.fromfile(os.path.join(td, "o.rgb"), dtype=np.uint8).reshape(infos[0].height, infos[0].width, 3)
        assert np.array_equal(got, jpeg_cases.pil_decode(data))

    check()
