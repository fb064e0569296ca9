"""Index arithmetic of the transposing-read weight-gradient kernel (csrc/wgrad.hip: wgrad_tile), emulated on
the CPU against the lane mapping of ds_read_b64_tr_b16 that was MEASURED on an MI355X (profiles/r01_ds_read_tr_probe.txt):
the DMA role (which 16-byte chunk each lane puts where, source-side swizzle), the fragment addresses of every lane and the
32x32x16 MFMA operand layout must together produce dW[n][k] = sum_m dY[m][n] * X[m][k] for a 128x128 tile."""
import os
import re

import numpy as np

from conftest import ROOT

MS = 32                                     # pixel rows per pipeline step (WG_MS of csrc/wgrad.hip)


def _measured_tr_semantics():
    """lane -> list of (supplying lane, element) for v[0..3], derived from probe case A (block g = 64 contiguous elements,
    lane t of a group supplied the address of elements 4t..4t+3)."""
    sem = {}
    for line in open(os.path.join(ROOT, "profiles", "r01_ds_read_tr_probe.txt")):
        m = re.match(r"lane\s+(\d+)\s+A:\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)", line)
        if m:
            lane = int(m.group(1))
            g = lane >> 4
            src = []
            for v in map(int, m.groups()[1:]):
                e = v - g * 64
                src.append((g * 16 + e // 4, e % 4))          # lane that supplied the address, element inside its 8 bytes
            sem[lane] = src
    assert len(sem) == 64
    return sem


def tr_read(lds_u16, addr_bytes, sem):
    """One wave-wide ds_read_b64_tr_b16: addr_bytes[lane] -> [64][4] values."""
    out = np.zeros((64, 4), dtype=lds_u16.dtype)
    for lane in range(64):
        for j, (src_lane, e) in enumerate(sem[lane]):
            a = addr_bytes[src_lane]
            assert a % 8 == 0
            out[lane, j] = lds_u16[a // 2 + e]
    return out


def fill_image(tile):
    """LDS image of one operand: tile [MS][128] -> u16-addressed buffer, by the kernel's DMA role."""
    img = np.zeros(MS * 128, dtype=tile.dtype)
    for wave in range(4):
        for i in range(MS // 16):
            for lane in range(64):
                rsub = lane >> 4
                row = (wave + 4 * i) * 4 + rsub
                row7 = ((wave & 1) << 2) + rsub
                assert row7 == (row & 7)
                cg = (lane & 15) ^ (row7 << 1)                 # global chunk fetched by this lane
                lds_byte = wave * 1024 + i * 4096 + lane * 16  # lane-linear DMA destination
                img[lds_byte // 2: lds_byte // 2 + 8] = tile[row, cg * 8: cg * 8 + 8]
    return img


def frag_offsets(lane, wsel):
    """offY / offX of the kernel for this lane: [fragment][q] byte offsets inside an operand image (slice 0)."""
    tl, gb, fh = lane & 15, (lane >> 4) & 1, lane >> 5
    off = [[0, 0], [0, 0]]
    for f in range(2):
        for q in range(2):
            r7 = 4 * q + (tl >> 2)
            row = 8 * fh + r7
            c = wsel * 64 + f * 32 + 16 * gb + 4 * (tl & 3)
            off[f][q] = row * 256 + ((((c >> 3) ^ (r7 << 1)) & 15) << 4) + (c & 7) * 2
    return off


def test_wgrad_tr_fragments_reproduce_the_tile_product():
    sem = _measured_tr_semantics()
    rng = np.random.default_rng(0)
    dY = rng.integers(-3, 4, size=(MS, 128)).astype(np.int64)            # [pixel][n]
    X = rng.integers(-3, 4, size=(MS, 128)).astype(np.int64)             # [pixel][k]
    imgY, imgX = fill_image(dY), fill_image(X)
    want = dY.T @ X                                                      # [n][k]
    got = np.zeros((128, 128), dtype=np.int64)
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        offY = [frag_offsets(l, wr) for l in range(64)]
        offX = [frag_offsets(l, wc) for l in range(64)]
        for ks in range(MS // 16):
            A = np.zeros((2, 64, 8), dtype=np.int64)                     # [fragment][lane][8 reduction elements]
            Bm = np.zeros((2, 64, 8), dtype=np.int64)
            for f in range(2):
                for q in range(2):
                    A[f][:, 4 * q:4 * q + 4] = tr_read(imgY, [offY[l][f][q] + ks * 4096 for l in range(64)], sem)
                    Bm[f][:, 4 * q:4 * q + 4] = tr_read(imgX, [offX[l][f][q] + ks * 4096 for l in range(64)], sem)
            for i in range(2):
                for j in range(2):
                    # v_mfma_f32_32x32x16: A lane l = row l&31, reduction 8*(l>>5)+e; B lane l = column l&31, same reduction
                    a = np.zeros((32, 16), dtype=np.int64)
                    b = np.zeros((16, 32), dtype=np.int64)
                    for l in range(64):
                        a[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = A[i][l]
                        b[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = Bm[j][l]
                    n0, k0 = wr * 64 + i * 32, wc * 64 + j * 32
                    got[n0:n0 + 32, k0:k0 + 32] += a @ b
    assert np.array_equal(got, want)


def test_wgrad_tr_bias_column_sums():
    """The bias-gradient path: thread t reads the 16-byte LDS slot t&15 of rows (t>>4) + 16j and attributes it to the global
    chunk (t&15) ^ (((t>>4)&7)<<1); 16 row groups are then added per column."""
    rng = np.random.default_rng(1)
    dY = rng.integers(-50, 50, size=(MS, 128)).astype(np.int64)
    img = fill_image(dY)
    red = np.zeros((16, 128), dtype=np.int64)
    for t in range(256):
        bcg = (t & 15) ^ ((((t >> 4) & 7) << 1) & 15)
        for j in range(MS // 16):
            a = (((t >> 4) + 16 * j) * 256 + (t & 15) * 16) // 2
            red[t >> 4, bcg * 8: bcg * 8 + 8] += img[a: a + 8]
    assert np.array_equal(red.sum(0), dY.sum(0))


def test_wgrad_tr_pixel_stepping_is_division_free_and_exact():
    """issue_step keeps (b, oh, ow) of each lane's rows incrementally (16 pixels between a lane's DMA instructions, MS between
    steps, each decomposed once into images + rows + pixels with single carries): must equal divmod for every geometry the
    path has, including images smaller than 16 pixels."""
    for (OH, OW, m_begin) in [(104, 104, 0), (13, 13, 128), (9, 9, 0), (1, 1, 256), (2, 3, 128), (26, 26, 1280), (4, 5, 0)]:
        OHW = OH * OW
        for ms in (32, 64):
            nd = ms // 16
            d16b = 16 // OHW
            d16q, d16r = divmod(16 - d16b * OHW, OW)
            dMSb = ms // OHW
            dMSq, dMSr = divmod(ms - dMSb * OHW, OW)
            for wave in range(4):
                for rsub in range(4):
                    m = m_begin + wave * 4 + rsub
                    rb, r = divmod(m, OHW)
                    roh, row_ = divmod(r, OW)
                    m_issue = m_begin
                    for _step in range(40):
                        b, oh, ow = rb, roh, row_
                        for i in range(nd):
                            mm = m_issue + (wave + 4 * i) * 4 + rsub
                            eb, er = divmod(mm, OHW)
                            assert (b, oh, ow) == (eb,) + divmod(er, OW), (OH, OW, ms, wave, rsub, _step, i)
                            b, oh, ow = b + d16b, oh + d16q, ow + d16r
                            if ow >= OW:
                                ow -= OW
                                oh += 1
                            if oh >= OH:
                                oh -= OH
                                b += 1
                        m_issue += ms
                        rb, roh, row_ = rb + dMSb, roh + dMSq, row_ + dMSr
                        if row_ >= OW:
                            row_ -= OW
                            roh += 1
                        if roh >= OH:
                            roh -= OH
                            rb += 1
