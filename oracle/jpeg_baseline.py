"""TEST INFRASTRUCTURE - CPU restatement of baseline JPEG decoding as the reference's loader gets it from OpenCV.

Reference call site: utils/dataset.py:127-129
    ori_img = cv2.imdecode(np.frombuffer(ref['img'], np.uint8), cv2.IMREAD_COLOR); img = cv2.cvtColor(ori_img, cv2.COLOR_BGR2RGB)
The arithmetic lives in a third-party dependency that is NOT under /root/reference: OpenCV's bundled libjpeg-turbo (the
reference pins no version: requirement.txt lists `opencv-python` without one).  What is restated here is libjpeg's published
decoder at its DEFAULT settings, which is what cv2.imdecode uses: sequential Huffman entropy decoding (ITU-T T.81 Annex F),
dct_method JDCT_ISLOW (jidctint.c: Loeffler-Ligtenberg-Moschytz, CONST_BITS 13, PASS1_BITS 2), do_fancy_upsampling (jdsample.c:
h2v1 / h2v2 triangle filters), YCbCr -> RGB with the 16-bit fixed-point tables of jdcolor.c.
PARITY PIN: cv2 is not installed here, but libjpeg-turbo itself is - inside Pillow (PIL 12.2.0, libjpeg-turbo, API 6.2):
tests/test_jpeg_host.py checks this file bit for bit against PIL's decoder on JPEGs of all supported samplings, qualities,
odd sizes, optimised Huffman tables and restart intervals, and against the committed vectors under tests/golden/jpeg/.
Progressive files (SOF2: spectral selection + successive approximation, T.81 Annex G / jdphuff.c) and sequential files with
one scan per component are restated too (entropy_decode_general).  Not restated (the product rejects them too): arithmetic
coding / 12-bit / lossless / CMYK files.  The EXIF orientation (OpenCV turns the decoded array on IMREAD_COLOR) is handled
and tested outside this file (jpegdec.apply_orientation against PIL.ImageOps.exif_transpose).

Only tests/, bench.py's cpu_baseline and selfcheck may import this module.  Pure Python / numpy: use small images.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62,
                   63])        # jpeg_natural_order: zigzag position -> row-major position


class JpegError(ValueError):
    pass


class Header:
    pass


def parse(data: bytes) -> Header:
    """markers up to and including SOS (T.81 Annex B)"""
    if data[:2] != b"\xff\xd8":
        raise JpegError("not a JPEG (no SOI)")
    h = Header()
    h.qt, h.dc, h.ac, h.restart_interval = {}, {}, {}, 0
    h.comps, h.progressive, h.multiscan = None, False, False
    pos = 2
    while True:
        while data[pos] != 0xFF:
            pos += 1
        while data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            raise JpegError("EOI before SOS")
        n = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2:pos + n]
        pos += n
        if m == 0xDB:                                   # DQT
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                tab = np.zeros(64, dtype=np.int64)
                for k in range(64):
                    if pq:
                        tab[ZIGZAG[k]] = (seg[i] << 8) | seg[i + 1]
                        i += 2
                    else:
                        tab[ZIGZAG[k]] = seg[i]
                        i += 1
                h.qt[tq] = tab
        elif m == 0xC4:                                 # DHT
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                nsym = sum(counts)
                syms = list(seg[i + 17:i + 17 + nsym])
                i += 17 + nsym
                (h.ac if tc else h.dc)[th] = _huff_table(counts, syms)
        elif m in (0xC0, 0xC1, 0xC2):                   # SOF0 / SOF1 (Huffman sequential), SOF2 (Huffman progressive)
            h.progressive = m == 0xC2
            if seg[0] != 8:
                raise JpegError("only 8-bit samples")
            h.height, h.width, nc = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            h.comps = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c]) for c in range(nc)]
        elif 0xC3 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise JpegError("unsupported JPEG process (SOF%d: lossless / arithmetic)" % (m - 0xC0))
        elif m == 0xDD:
            h.restart_interval = (seg[0] << 8) | seg[1]
        elif m == 0xDA:                                 # SOS
            ns = seg[0]
            if h.comps is None:
                raise JpegError("SOS before SOF")
            h.multiscan = h.progressive or ns != len(h.comps)          # more scans follow: entropy_decode_general
            for s in range(ns):
                cid, tabs = seg[1 + 2 * s], seg[2 + 2 * s]
                c = next(c for c in h.comps if c["id"] == cid)
                c["td"], c["ta"] = tabs >> 4, tabs & 15
            h.scan_start = pos
            break
    nc = len(h.comps)
    if nc not in (1, 3):
        raise JpegError("1 or 3 components")
    if nc == 1:
        h.comps[0]["h"] = h.comps[0]["v"] = 1          # a single-component scan is never interleaved (T.81 A.2.2)
    hmax, vmax = max(c["h"] for c in h.comps), max(c["v"] for c in h.comps)
    if nc == 3:
        ok = (h.comps[1]["h"], h.comps[1]["v"], h.comps[2]["h"], h.comps[2]["v"]) == (1, 1, 1, 1) and (hmax, vmax) in ((1, 1), (2, 1), (2, 2))
        if not ok:
            raise JpegError("supported samplings: 4:4:4, 4:2:2, 4:2:0")
    h.hmax, h.vmax = hmax, vmax
    h.mcus_x = -(-h.width // (8 * hmax))
    h.mcus_y = -(-h.height // (8 * vmax))
    for c in h.comps:
        c["bw"], c["bh"] = h.mcus_x * c["h"], h.mcus_y * c["v"]
        c["dw"] = -(-h.width * c["h"] // hmax)          # downsampled_width / height (jdmaster.c)
        c["dh"] = -(-h.height * c["v"] // vmax)
    return h


def _huff_table(counts, syms):
    """code -> symbol as a dict keyed by (length, code) (T.81 Annex C)"""
    table, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            table[(ln, code)] = syms[k]
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    def __init__(self, data, pos):
        self.d, self.pos, self.acc, self.n = data, pos, 0, 0

    def bit(self):
        if self.n == 0:
            b = self.d[self.pos] if self.pos < len(self.d) else 0
            self.pos += 1
            if b == 0xFF:
                b2 = self.d[self.pos] if self.pos < len(self.d) else 0
                if b2 == 0:
                    self.pos += 1                        # stuffed zero
                else:
                    self.pos -= 1                        # a marker: feed zeros (jdhuff.c does the same at a premature end)
                    b = 0
            self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, s):
        v = 0
        for _ in range(s):
            v = (v << 1) | self.bit()
        return v

    def symbol(self, table):
        code = 0
        for ln in range(1, 17):
            code = (code << 1) | self.bit()
            s = table.get((ln, code))
            if s is not None:
                return s
        raise JpegError("bad Huffman code")

    def restart(self):
        """byte-align and consume the RSTn marker"""
        self.n = 0
        while not (self.d[self.pos] == 0xFF and 0xD0 <= self.d[self.pos + 1] <= 0xD7):
            self.pos += 1
        self.pos += 2


def _extend(v, s):
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def entropy_decode(data: bytes, h: Header):
    """quantised coefficients per component: int16 [bh, bw, 64] in natural (row-major) order (T.81 Annex F.2)"""
    coefs = [np.zeros((c["bh"], c["bw"], 64), dtype=np.int16) for c in h.comps]
    br = _Bits(data, h.scan_start)
    pred = [0] * len(h.comps)
    left = h.restart_interval
    for my in range(h.mcus_y):
        for mx in range(h.mcus_x):
            if h.restart_interval and left == 0:
                br.restart()
                pred = [0] * len(h.comps)
                left = h.restart_interval
            for ci, c in enumerate(h.comps):
                dc, ac = h.dc[c["td"]], h.ac[c["ta"]]
                for v in range(c["v"]):
                    for u in range(c["h"]):
                        blk = coefs[ci][my * c["v"] + v, mx * c["h"] + u]
                        s = br.symbol(dc)
                        if s:
                            pred[ci] += _extend(br.bits(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = br.symbol(ac)
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                            k += 1
            left -= 1
    return coefs


# ---- multi-scan files: progressive (T.81 Annex G; libjpeg jdphuff.c) and sequential files with one scan per component -----
def _scan_blocks(h, scomps):
    """(component index, block row, block col) per MCU of a scan, with the restart-interval unit (one MCU)"""
    if len(scomps) == 1:                                              # non-interleaved: the component's REAL blocks, raster order
        ci = scomps[0]
        c = h.comps[ci]
        for by in range(-(-c["dh"] // 8)):
            for bx in range(-(-c["dw"] // 8)):
                yield [(ci, by, bx)]
    else:
        for my in range(h.mcus_y):
            for mx in range(h.mcus_x):
                yield [(ci, my * h.comps[ci]["v"] + v, mx * h.comps[ci]["h"] + u)
                       for ci in scomps for v in range(h.comps[ci]["v"]) for u in range(h.comps[ci]["h"])]


def entropy_decode_general(data: bytes, h: Header):
    """every scan of the file accumulated into the coefficient arrays: DC / AC first and refinement passes of a progressive
    file, or the per-component scans of a sequential one"""
    coefs = [np.zeros((c["bh"], c["bw"], 64), dtype=np.int16) for c in h.comps]
    dc, ac, ri = {}, {}, 0
    pos = 2
    while pos < len(data):
        while pos < len(data) and data[pos] != 0xFF:
            pos += 1
        while pos < len(data) and data[pos] == 0xFF:
            pos += 1
        if pos >= len(data):
            break
        m = data[pos]
        pos += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7 or m == 0x00:
            continue
        if m == 0xD9:
            break
        n = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2:pos + n]
        pos += n
        if m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                nsym = sum(counts)
                (ac if tc else dc)[th] = _huff_table(counts, list(seg[i + 17:i + 17 + nsym]))
                i += 17 + nsym
        elif m == 0xDD:
            ri = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            scomps, tabs = [], {}
            for k in range(ns):
                ci = next(i for i, c in enumerate(h.comps) if c["id"] == seg[1 + 2 * k])
                scomps.append(ci)
                tabs[ci] = (seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15)
            Ss, Se, Ah, Al = seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns] >> 4, seg[3 + 2 * ns] & 15
            if not h.progressive:
                Ss, Se, Ah, Al = 0, 63, 0, 0
            br = _Bits(data, pos)
            pred = {ci: 0 for ci in scomps}
            eobrun, left = 0, ri
            for mcu in _scan_blocks(h, scomps):
                if ri and left == 0:
                    br.restart()
                    pred = {ci: 0 for ci in scomps}
                    eobrun, left = 0, ri
                for ci, by, bx in mcu:
                    blk = coefs[ci][by, bx]
                    if not h.progressive:
                        s = br.symbol(dc[tabs[ci][0]])
                        if s:
                            pred[ci] += _extend(br.bits(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = br.symbol(ac[tabs[ci][1]])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                            k += 1
                    elif Ss == 0:
                        if Ah == 0:                                   # DC first pass (G.1.2.1)
                            s = br.symbol(dc[tabs[ci][0]])
                            if s:
                                pred[ci] += _extend(br.bits(s), s)
                            blk[0] = pred[ci] << Al
                        elif br.bit():                                # DC refinement: one more bit
                            blk[0] |= 1 << Al
                    elif Ah == 0:                                     # AC first pass (G.1.2.2)
                        if eobrun > 0:
                            eobrun -= 1
                            continue
                        k = Ss
                        while k <= Se:
                            rs = br.symbol(ac[tabs[ci][1]])
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                blk[ZIGZAG[k]] = _extend(br.bits(s), s) << Al
                            elif r == 15:
                                k += 15
                            else:
                                eobrun = (1 << r) + (br.bits(r) if r else 0) - 1
                                break
                            k += 1
                    else:                                             # AC refinement (G.1.2.3; jdphuff.c decode_mcu_AC_refine)
                        p1, m1 = 1 << Al, -1 << Al
                        k = Ss
                        if eobrun == 0:
                            while k <= Se:
                                rs = br.symbol(ac[tabs[ci][1]])
                                r, s = rs >> 4, rs & 15
                                if s:
                                    s = p1 if br.bit() else m1
                                elif r != 15:
                                    eobrun = (1 << r) + (br.bits(r) if r else 0)
                                    break
                                while k <= Se:
                                    z = ZIGZAG[k]
                                    if blk[z] != 0:
                                        if br.bit() and (int(blk[z]) & p1) == 0:
                                            blk[z] += p1 if blk[z] >= 0 else m1
                                    else:
                                        r -= 1
                                        if r < 0:
                                            break
                                    k += 1
                                if s:
                                    blk[ZIGZAG[k]] = s
                                k += 1
                        if eobrun > 0:
                            while k <= Se:
                                z = ZIGZAG[k]
                                if blk[z] != 0 and br.bit() and (int(blk[z]) & p1) == 0:
                                    blk[z] += p1 if blk[z] >= 0 else m1
                                k += 1
                            eobrun -= 1
                left -= 1
            # the reader stops in front of the marker that ends the scan (or ran over it by at most its look-ahead: none here)
            pos = br.pos
    return coefs


# ---- jidctint.c: jpeg_idct_islow -----------------------------------------------------------------------------------
CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift):
    """one LLM pass over axis -1 of int64 [..., 8]; pass 1 keeps 2 fraction bits, pass 2 removes all"""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    tmp0 = (v[..., 0] + v[..., 4]) << CONST_BITS
    tmp1 = (v[..., 0] - v[..., 4]) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F_1_175875602
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F_0_298631336, tmp1 * F_2_053119869, tmp2 * F_3_072711026, tmp3 * F_1_501321110
    z1, z2, z3, z4 = z1 * (-F_0_899976223), z2 * (-F_2_562915447), z3 * (-F_1_961570560) + z5, z4 * (-F_0_390180644) + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], axis=-1)
    return _descale(out, shift)


def _range_limit_idct(x):
    """IDCT_range_limit[x & RANGE_MASK] (jdmaster.c prepare_range_limit_table): clamp(x + 128, 0, 255) for -512 <= x < 512,
    with the table's wrap-around beyond"""
    i = x & 1023
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def idct_planes(h: Header, coefs):
    """dequantise + islow IDCT: uint8 sample planes [bh*8, bw*8] per component"""
    planes = []
    for c, cf in zip(h.comps, coefs):
        q = h.qt[c["tq"]].reshape(8, 8)
        blk = cf.astype(np.int64).reshape(c["bh"], c["bw"], 8, 8) * q                  # [by, bx, row, col]
        ws = _idct_1d(np.swapaxes(blk, -1, -2), CONST_BITS - PASS1_BITS)               # pass 1: columns  -> [.., col, row]
        ws = np.swapaxes(ws, -1, -2)                                                   #                      [.., row, col]
        px = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3)                                 # pass 2: rows
        px = _range_limit_idct(px)
        planes.append(px.transpose(0, 2, 1, 3).reshape(c["bh"] * 8, c["bw"] * 8))
    return planes


# ---- jdsample.c (fancy upsampling) + jdcolor.c -----------------------------------------------------------------------
def _h2v1_fancy(p):
    """[rows, n] -> [rows, 2n]: 3/4 nearer + 1/4 further, rounding 1 / 2 alternately; edge columns copied"""
    p = p.astype(np.int64)
    n = p.shape[1]
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    even = (3 * p + left + 1) >> 2
    odd = (3 * p + right + 2) >> 2
    even[:, 0] = p[:, 0]
    odd[:, -1] = p[:, -1]
    out = np.empty((p.shape[0], 2 * n), dtype=np.int64)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out


def _h2v2_fancy(p):
    """[r, n] -> [2r, 2n]: 9/16, 3/16, 3/16, 1/16; rows beyond the component's real rows are its edge rows (jdmainct.c context
    rows), edge columns use the column sum itself"""
    p = p.astype(np.int64)
    r, n = p.shape
    up = np.concatenate([p[:1], p[:-1]], axis=0)
    dn = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * r, 2 * n), dtype=np.int64)
    for v, other in ((0, up), (1, dn)):
        cs = 3 * p + other                                             # thiscolsum
        last = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
        out[v::2, 0::2] = (3 * cs + last + 8) >> 4
        out[v::2, 1::2] = (3 * cs + nxt + 7) >> 4
    return out


def _ycc_to_rgb(y, cb, cr):
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    half = 1 << 15
    r = y + ((91881 * cr + half) >> 16)
    g = y + ((-22554 * cb + half - 46802 * cr) >> 16)
    b = y + ((116130 * cb + half) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def reconstruct(h: Header, planes):
    """sample planes -> RGB uint8 [H, W, 3] (IMREAD_COLOR + COLOR_BGR2RGB)"""
    H, W = h.height, h.width
    if len(h.comps) == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], axis=-1)
    y = planes[0][:H, :W]
    ch = []
    for c, p in zip(h.comps[1:], planes[1:]):
        p = p[:c["dh"], :c["dw"]]                                   # the component's real samples only
        fancy = c["dw"] > 2                                         # jdsample.c jinit_upsampler: fancy only if downsampled_width > 2
        if (h.hmax, h.vmax) == (2, 2):
            p = _h2v2_fancy(p) if fancy else np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)
        elif (h.hmax, h.vmax) == (2, 1):
            p = _h2v1_fancy(p) if fancy else np.repeat(p, 2, axis=1)
        ch.append(p[:H, :W])
    return _ycc_to_rgb(y, ch[0], ch[1])


def decode(data: bytes):
    h = parse(data)
    coefs = entropy_decode_general(data, h) if h.multiscan else entropy_decode(data, h)
    return reconstruct(h, idct_planes(h, coefs))
