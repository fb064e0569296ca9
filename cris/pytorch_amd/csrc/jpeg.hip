// Baseline JPEG decoding for the input pipeline (include/cris_hip.h "Baseline JPEG decoding"; reference call site
// utils/dataset.py:127-129: cv2.imdecode + cvtColor on the raw file bytes of an LMDB record).
//
// Host part (this file, plain C++): marker parsing and Huffman entropy decoding (ITU-T T.81 Annex B / F; progressive files:
// Annex G - DC / AC first and refinement scans accumulated into the same coefficient blocks) into quantised coefficient
// blocks - bit-serial, data-dependent, one image per host thread.  Device part: everything that is arithmetic.
//   jpeg_idct_kernel   one 8x8 block per 8 lanes (64 lanes = 8 blocks): lane (b, c) dequantises and transforms COLUMN c of
//                      block b (pass 1), the 8x8 workspace goes through LDS, lane (b, r) transforms ROW r (pass 2) and
//                      writes its 8 samples with one 8-byte store.  Reads 128 B of coefficients per block (the quantisation
//                      tables sit in the descriptor), writes 64 B: HBM-bound, 3 B/pixel in + 1.5 B/pixel out at 4:2:0.
//   jpeg_color_kernel  one output pixel per lane: luma sample + the two chroma samples through libjpeg's "fancy" triangle
//                      filters (neighbouring chroma samples come from L2: each is read by ~4 lanes), YCbCr -> RGB, 3 bytes out.
// The per-block / per-pixel arithmetic is jpeg_core.h (also compiled by g++ for the CPU-side check against the oracle).
#include "common.h"
#include "../../../include/cris_hip.h"
#include "jpeg_core.h"
#include <string.h>
#include <thread>
#include <vector>
#include <atomic>
#include <string>

namespace {

const unsigned char kNatural[64 + 16] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,
                                         7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38,
                                         31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                         63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // (run overshoot of a corrupt stream lands here)

struct HuffTable {
    bool present = false;
    // 9-bit look-ahead: (length << 8) | symbol for codes of <= 9 bits, 0 otherwise
    unsigned short look[512];
    int maxcode[18];           // largest code of each length (-1 if none), [17] = sentinel
    int valoff[17];            // symbol index of the first code of a length minus that code
    unsigned char syms[256];
};

struct Tables {
    unsigned short qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    int comp_tq[3], comp_td[3], comp_ta[3];
};

#define JERR(...)                            \
    do {                                     \
        cris_set_error(__VA_ARGS__);         \
        return -1;                           \
    } while (0)

int build_huff(HuffTable& t, const unsigned char* counts, const unsigned char* syms, int nsym) {
    memset(t.look, 0, sizeof(t.look));
    memcpy(t.syms, syms, nsym);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        t.valoff[len] = k - code;
        for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
            if (len <= 9) {
                const int lo = code << (9 - len), n = 1 << (9 - len);
                if (lo + n > 512) return -1;
                for (int j = 0; j < n; ++j) t.look[lo + j] = (unsigned short)((len << 8) | syms[k]);
            }
        }
        t.maxcode[len] = counts[len - 1] ? code - 1 : -1;
        if (code > (1 << len)) return -1;                 // over-subscribed
        code <<= 1;
    }
    t.maxcode[17] = 0x7FFFFFFF;
    t.present = true;
    return 0;
}

// markers up to SOS: fills info (geometry, quantisation tables) and, when `tb` is given, the entropy tables
int parse_header(const unsigned char* d, size_t n, cris_jpeg_info* info, Tables* tb) {
    Tables local;
    Tables& T = tb ? *tb : local;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) JERR("cris_jpeg: not a JPEG file (no SOI)");
    memset(info, 0, sizeof(*info));
    size_t pos = 2;
    bool have_sof = false, progressive = false, first_scan_all = true;
    int comp_id[3] = {0, 0, 0};
    for (;;) {
        while (pos < n && d[pos] != 0xFF) ++pos;
        while (pos < n && d[pos] == 0xFF) ++pos;
        if (pos >= n) JERR("cris_jpeg: no SOS marker");
        const int m = d[pos++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) JERR("cris_jpeg: EOI before SOS");
        if (pos + 2 > n) JERR("cris_jpeg: truncated segment");
        const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
        if (len < 2 || pos + len > n) JERR("cris_jpeg: truncated segment");
        const unsigned char* s = d + pos + 2;
        const size_t sl = len - 2;
        pos += len;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > sl) JERR("cris_jpeg: bad DQT");
                for (int k = 0; k < 64; ++k) {
                    T.qt[tq][kNatural[k]] = pq ? (unsigned short)((s[i] << 8) | s[i + 1]) : s[i];
                    i += pq ? 2 : 1;
                }
                T.qt_present[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) JERR("cris_jpeg: bad DHT");
                const int tc = s[i] >> 4, th = s[i] & 15;
                int nsym = 0;
                for (int k = 0; k < 16; ++k) nsym += s[i + 1 + k];
                if (tc > 1 || th > 3 || nsym > 256 || i + 17 + nsym > sl) JERR("cris_jpeg: bad DHT");
                if (build_huff(tc ? T.ac[th] : T.dc[th], s + i + 1, s + i + 17, nsym)) JERR("cris_jpeg: invalid Huffman table");
                i += 17 + nsym;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (m == 0xC2) progressive = true;
            if (sl < 6 || s[0] != 8) JERR("cris_jpeg: only 8-bit samples are supported");
            info->height = (s[1] << 8) | s[2];
            info->width = (s[3] << 8) | s[4];
            info->ncomp = s[5];
            if ((info->ncomp != 1 && info->ncomp != 3) || sl < (size_t)(6 + 3 * info->ncomp)) JERR("cris_jpeg: 1 or 3 components are supported (got %d)", info->ncomp);
            if (info->width <= 0 || info->height <= 0) JERR("cris_jpeg: empty image");
            if ((long)info->width * info->height > (1L << 28)) JERR("cris_jpeg: image larger than 2^28 pixels (%d x %d)", info->width, info->height);
            for (int c = 0; c < info->ncomp; ++c) {
                comp_id[c] = s[6 + 3 * c];
                info->comp_h[c] = s[7 + 3 * c] >> 4;
                info->comp_v[c] = s[7 + 3 * c] & 15;
                T.comp_tq[c] = s[8 + 3 * c];
                if (T.comp_tq[c] > 3) JERR("cris_jpeg: bad quantisation table index");
            }
            have_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            JERR("cris_jpeg: unsupported JPEG process SOF%d (lossless / arithmetic coding): decode this file on the CPU", m - 0xC0);
        } else if (m == 0xDD) {
            if (sl < 2) JERR("cris_jpeg: bad DRI");
            info->restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof) JERR("cris_jpeg: SOS before SOF");
            if (sl < 1 || s[0] < 1 || s[0] > info->ncomp || sl < (size_t)(1 + 2 * s[0] + 3)) JERR("cris_jpeg: bad SOS");
            first_scan_all = s[0] == info->ncomp;
            for (int k = 0; k < s[0]; ++k) {
                int c = -1;
                for (int q = 0; q < info->ncomp; ++q)
                    if (comp_id[q] == s[1 + 2 * k]) c = q;
                if (c < 0 || (first_scan_all && c != k)) JERR("cris_jpeg: scan components out of frame order");
                T.comp_td[c] = s[2 + 2 * k] >> 4;
                T.comp_ta[c] = s[2 + 2 * k] & 15;
                if (T.comp_td[c] > 3 || T.comp_ta[c] > 3) JERR("cris_jpeg: bad Huffman table index");
            }
            info->scan_offset = (long)pos;
            break;
        }
    }
    if (info->ncomp == 1) info->comp_h[0] = info->comp_v[0] = 1;       // a one-component scan is not interleaved (T.81 A.2.2)
    info->hmax = info->comp_h[0];
    info->vmax = info->comp_v[0];
    if (info->ncomp == 3) {
        const bool chroma11 = info->comp_h[1] == 1 && info->comp_v[1] == 1 && info->comp_h[2] == 1 && info->comp_v[2] == 1;
        const bool ok = chroma11 && ((info->hmax == 1 && info->vmax == 1) || (info->hmax == 2 && info->vmax == 1) || (info->hmax == 2 && info->vmax == 2));
        if (!ok) JERR("cris_jpeg: supported chroma samplings are 4:4:4, 4:2:2 and 4:2:0 (luma %dx%d)", info->comp_h[0], info->comp_v[0]);
    }
    info->mcus_x = (info->width + 8 * info->hmax - 1) / (8 * info->hmax);
    info->mcus_y = (info->height + 8 * info->vmax - 1) / (8 * info->vmax);
    long co = 0, po = 0;
    int nb = 0;
    for (int c = 0; c < info->ncomp; ++c) {
        if (!T.qt_present[T.comp_tq[c]]) JERR("cris_jpeg: missing quantisation table");
        memcpy(info->quant[c], T.qt[T.comp_tq[c]], sizeof(info->quant[c]));
        info->blocks_w[c] = info->mcus_x * info->comp_h[c];
        info->blocks_h[c] = info->mcus_y * info->comp_v[c];
        info->down_w[c] = (info->width * info->comp_h[c] + info->hmax - 1) / info->hmax;
        info->down_h[c] = (info->height * info->comp_v[c] + info->vmax - 1) / info->vmax;
        info->coef_offset[c] = co;
        info->plane_offset[c] = po;
        const long blocks = (long)info->blocks_w[c] * info->blocks_h[c];
        co += blocks * 64;
        po += blocks * 64;
        nb += (int)blocks;
    }
    info->coef_count = co;
    info->plane_bytes = po;
    info->total_blocks = nb;
    // 1: the coefficients are spread over several scans (progressive, or a sequential file with one scan per component):
    // decode_general walks all of them; 0: one interleaved sequential scan (decode_scan)
    info->multiscan = (progressive || !first_scan_all) ? (progressive ? 2 : 1) : 0;
    if (tb && !info->multiscan)
        for (int c = 0; c < info->ncomp; ++c)
            if (!T.dc[T.comp_td[c]].present || !T.ac[T.comp_ta[c]].present) JERR("cris_jpeg: missing Huffman table");
    return 0;
}

// bit reader over the entropy-coded segment: 0xFF00 -> 0xFF, a marker ends the data (zeros are fed beyond it, like jdhuff.c)
struct BitReader {
    const unsigned char* d;
    size_t n, pos;
    unsigned long long acc = 0;     // bits left-aligned at bit (cnt-1)
    int cnt = 0;
    bool hit_marker = false;
    BitReader(const unsigned char* d_, size_t n_, size_t pos_) : d(d_), n(n_), pos(pos_) {}
    inline void fill() {
        while (cnt <= 56) {
            unsigned b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    const unsigned b2 = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (b2 == 0) pos += 2;
                    else { hit_marker = true; b = 0; }
                } else {
                    ++pos;
                }
            }
            acc = (acc << 8) | b;
            cnt += 8;
        }
    }
    inline unsigned peek(int k) { return (unsigned)((acc >> (cnt - k)) & ((1u << k) - 1)); }
    inline void skip(int k) { cnt -= k; }
    inline int bits(int k) {
        if (cnt < k) fill();
        const int v = (int)peek(k);
        cnt -= k;
        return v;
    }
    inline int symbol(const HuffTable& t) {
        if (cnt < 16) fill();
        const unsigned short e = t.look[peek(9)];
        if (e) {
            cnt -= e >> 8;
            return e & 255;
        }
        int code = 0, len;
        for (len = 10; len <= 16; ++len) {           // (never peeks more than 16 bits: cnt >= 16 after fill())
            code = (int)peek(len);
            if (code <= t.maxcode[len]) break;
        }
        if (len > 16) return -1;
        cnt -= len;
        return t.syms[(code + t.valoff[len]) & 255];
    }
    // byte-align, find the next RSTn and step over it
    inline bool restart() {
        cnt = 0; acc = 0;
        hit_marker = false;
        while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) {
            if (d[pos] == 0xFF && d[pos + 1] != 0 && d[pos + 1] != 0xFF) return false;       // another marker: give up
            ++pos;
        }
        if (pos + 1 >= n) return false;
        pos += 2;
        return true;
    }
};

inline int huff_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

int decode_scan(const unsigned char* d, size_t n, const cris_jpeg_info& I, const Tables& T, short* coef) {
    memset(coef, 0, (size_t)I.coef_count * sizeof(short));
    BitReader br(d, n, (size_t)I.scan_offset);
    int pred[3] = {0, 0, 0};
    int left = I.restart_interval;
    for (int my = 0; my < I.mcus_y; ++my) {
        for (int mx = 0; mx < I.mcus_x; ++mx) {
            if (I.restart_interval && left == 0) {
                // the reader may have run ahead of the marker by whole bytes only up to the marker itself (fill() stops there)
                if (!br.restart()) JERR("cris_jpeg: missing restart marker");
                pred[0] = pred[1] = pred[2] = 0;
                left = I.restart_interval;
            }
            for (int c = 0; c < I.ncomp; ++c) {
                const HuffTable& dc = T.dc[T.comp_td[c]];
                const HuffTable& ac = T.ac[T.comp_ta[c]];
                for (int v = 0; v < I.comp_v[c]; ++v) {
                    for (int u = 0; u < I.comp_h[c]; ++u) {
                        short* blk = coef + I.coef_offset[c] + ((long)(my * I.comp_v[c] + v) * I.blocks_w[c] + (mx * I.comp_h[c] + u)) * 64;
                        int s = br.symbol(dc);
                        if (s < 0 || s > 15) JERR("cris_jpeg: corrupt entropy-coded data (DC)");
                        if (s) pred[c] += huff_extend(br.bits(s), s);
                        blk[0] = (short)pred[c];
                        for (int k = 1; k < 64;) {
                            const int rs = br.symbol(ac);
                            if (rs < 0) JERR("cris_jpeg: corrupt entropy-coded data (AC)");
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                k += 16;
                                continue;
                            }
                            k += r;
                            blk[kNatural[k]] = (short)huff_extend(br.bits(s), s);
                            ++k;
                        }
                    }
                }
            }
            --left;
        }
    }
    return 0;
}


// ---- multi-scan files: progressive (T.81 Annex G, libjpeg jdphuff.c) and sequential files with one scan per component ----
struct Scan {
    int ns, comp[3], td[3], ta[3], Ss, Se, Ah, Al;
};

// one block of a scan; returns non-zero on corrupt data
inline int scan_block(BitReader& br, const Scan& S, int k_in_scan, const Tables& T, bool progressive, short* blk, int& pred, int& eobrun) {
    const HuffTable& dc = T.dc[S.td[k_in_scan]];
    const HuffTable& ac = T.ac[S.ta[k_in_scan]];
    if (!progressive) {
        int s = br.symbol(dc);
        if (s < 0 || s > 15) return -1;
        if (s) pred += huff_extend(br.bits(s), s);
        blk[0] = (short)pred;
        for (int k = 1; k < 64;) {
            const int rs = br.symbol(ac);
            if (rs < 0) return -1;
            const int r = rs >> 4;
            s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            blk[kNatural[k]] = (short)huff_extend(br.bits(s), s);
            ++k;
        }
        return 0;
    }
    if (S.Ss == 0) {
        if (S.Ah == 0) {                               // DC first pass
            const int s = br.symbol(dc);
            if (s < 0 || s > 15) return -1;
            if (s) pred += huff_extend(br.bits(s), s);
            blk[0] = (short)(pred * (1 << S.Al));
        } else if (br.bits(1)) {                       // DC refinement
            blk[0] = (short)(blk[0] | (1 << S.Al));
        }
        return 0;
    }
    if (S.Ah == 0) {                                   // AC first pass
        if (eobrun > 0) {
            --eobrun;
            return 0;
        }
        for (int k = S.Ss; k <= S.Se; ++k) {
            const int rs = br.symbol(ac);
            if (rs < 0) return -1;
            const int r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                blk[kNatural[k]] = (short)(huff_extend(br.bits(s), s) * (1 << S.Al));
            } else if (r == 15) {
                k += 15;
            } else {
                eobrun = (1 << r) + (r ? br.bits(r) : 0) - 1;
                break;
            }
        }
        return 0;
    }
    // AC refinement (jdphuff.c decode_mcu_AC_refine)
    const int p1 = 1 << S.Al, m1 = -(1 << S.Al);
    int k = S.Ss;
    if (eobrun == 0) {
        for (; k <= S.Se; ++k) {
            const int rs = br.symbol(ac);
            if (rs < 0) return -1;
            int r = rs >> 4, s = rs & 15;
            if (s) {
                s = br.bits(1) ? p1 : m1;
            } else if (r != 15) {
                eobrun = (1 << r) + (r ? br.bits(r) : 0);
                break;
            }
            do {
                short* c = blk + kNatural[k];
                if (*c != 0) {
                    if (br.bits(1) && (*c & p1) == 0) *c = (short)(*c + (*c >= 0 ? p1 : m1));
                } else if (--r < 0) {
                    break;
                }
                ++k;
            } while (k <= S.Se);
            if (s) blk[kNatural[k]] = (short)s;
        }
    }
    if (eobrun > 0) {
        for (; k <= S.Se; ++k) {
            short* c = blk + kNatural[k];
            if (*c != 0 && br.bits(1) && (*c & p1) == 0) *c = (short)(*c + (*c >= 0 ? p1 : m1));
        }
        --eobrun;
    }
    return 0;
}

int decode_general(const unsigned char* d, size_t n, const cris_jpeg_info& I, short* coef) {
    memset(coef, 0, (size_t)I.coef_count * sizeof(short));
    const bool progressive = I.multiscan == 2;
    Tables T;
    int ri = 0, comp_id[3] = {0, 0, 0};
    size_t pos = 2;
    int scans = 0;
    while (pos < n) {
        while (pos < n && d[pos] != 0xFF) ++pos;
        while (pos < n && d[pos] == 0xFF) ++pos;
        if (pos >= n) break;
        const int m = d[pos++];
        if (m == 0xD8 || m == 0x01 || m == 0x00 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        if (pos + 2 > n) break;
        const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
        if (len < 2 || pos + len > n) JERR("cris_jpeg: truncated segment");
        const unsigned char* s = d + pos + 2;
        const size_t sl = len - 2;
        pos += len;
        if (m == 0xC4) {
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) JERR("cris_jpeg: bad DHT");
                const int tc = s[i] >> 4, th = s[i] & 15;
                int nsym = 0;
                for (int k = 0; k < 16; ++k) nsym += s[i + 1 + k];
                if (tc > 1 || th > 3 || nsym > 256 || i + 17 + nsym > sl) JERR("cris_jpeg: bad DHT");
                if (build_huff(tc ? T.ac[th] : T.dc[th], s + i + 1, s + i + 17, nsym)) JERR("cris_jpeg: invalid Huffman table");
                i += 17 + nsym;
            }
        } else if (m == 0xDD) {
            if (sl < 2) JERR("cris_jpeg: bad DRI");
            ri = (s[0] << 8) | s[1];
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            for (int c = 0; c < I.ncomp && sl >= (size_t)(6 + 3 * I.ncomp); ++c) comp_id[c] = s[6 + 3 * c];
        } else if (m == 0xDA) {
            Scan S;
            if (sl < 1 || s[0] < 1 || s[0] > I.ncomp || sl < (size_t)(1 + 2 * s[0] + 3)) JERR("cris_jpeg: bad SOS");
            S.ns = s[0];
            for (int k = 0; k < S.ns; ++k) {
                int c = -1;
                for (int q = 0; q < I.ncomp; ++q)
                    if (comp_id[q] == s[1 + 2 * k]) c = q;
                if (c < 0) JERR("cris_jpeg: unknown component in SOS");
                S.comp[k] = c;
                S.td[k] = s[2 + 2 * k] >> 4;
                S.ta[k] = s[2 + 2 * k] & 15;
                if (S.td[k] > 3 || S.ta[k] > 3) JERR("cris_jpeg: bad Huffman table index");
            }
            S.Ss = s[1 + 2 * S.ns]; S.Se = s[2 + 2 * S.ns]; S.Ah = s[3 + 2 * S.ns] >> 4; S.Al = s[3 + 2 * S.ns] & 15;
            if (!progressive) { S.Ss = 0; S.Se = 63; S.Ah = 0; S.Al = 0; }
            if (S.Ss > S.Se || S.Se > 63 || S.Al > 13 || (S.Ss == 0 && S.Se != 0 && progressive) || (S.Ss > 0 && S.ns != 1))
                JERR("cris_jpeg: invalid progressive scan parameters (Ss %d Se %d Ah %d Al %d, %d components)", S.Ss, S.Se, S.Ah, S.Al, S.ns);
            for (int k = 0; k < S.ns; ++k) {
                const bool need_dc = !progressive || (S.Ss == 0 && S.Ah == 0), need_ac = !progressive || S.Ss > 0;
                if ((need_dc && !T.dc[S.td[k]].present) || (need_ac && !T.ac[S.ta[k]].present)) JERR("cris_jpeg: missing Huffman table");
            }
            BitReader br(d, n, pos);
            int pred[3] = {0, 0, 0}, eobrun = 0, left = ri;
            auto restart_if_due = [&]() -> int {
                if (ri && left == 0) {
                    if (!br.restart()) return -1;
                    pred[0] = pred[1] = pred[2] = 0;
                    eobrun = 0;
                    left = ri;
                }
                return 0;
            };
            if (S.ns == 1) {                                      // non-interleaved: the component's real blocks
                const int c = S.comp[0];
                const int bw = (I.down_w[c] + 7) / 8, bh = (I.down_h[c] + 7) / 8;
                for (int by = 0; by < bh; ++by)
                    for (int bx = 0; bx < bw; ++bx) {
                        if (restart_if_due()) JERR("cris_jpeg: missing restart marker");
                        short* blk = coef + I.coef_offset[c] + ((long)by * I.blocks_w[c] + bx) * 64;
                        if (scan_block(br, S, 0, T, progressive, blk, pred[0], eobrun)) JERR("cris_jpeg: corrupt entropy-coded data");
                        --left;
                    }
            } else {
                for (int my = 0; my < I.mcus_y; ++my)
                    for (int mx = 0; mx < I.mcus_x; ++mx) {
                        if (restart_if_due()) JERR("cris_jpeg: missing restart marker");
                        for (int k = 0; k < S.ns; ++k) {
                            const int c = S.comp[k];
                            for (int v = 0; v < I.comp_v[c]; ++v)
                                for (int u = 0; u < I.comp_h[c]; ++u) {
                                    short* blk = coef + I.coef_offset[c] + ((long)(my * I.comp_v[c] + v) * I.blocks_w[c] + (mx * I.comp_h[c] + u)) * 64;
                                    if (scan_block(br, S, k, T, progressive, blk, pred[k], eobrun)) JERR("cris_jpeg: corrupt entropy-coded data");
                                }
                        }
                        --left;
                    }
            }
            pos = br.pos;                                          // the reader never passes the marker that ends the scan
            ++scans;
        }
    }
    if (scans == 0) JERR("cris_jpeg: no scan");
    return 0;
}
}  // namespace

extern "C" int cris_jpeg_read_header(const unsigned char* data, size_t nbytes, cris_jpeg_info* info) {
    CRIS_CHECK_ARG(data && info, "null argument");
    return parse_header(data, nbytes, info, nullptr);
}

extern "C" int cris_jpeg_decode_coefficients(const unsigned char* data, size_t nbytes, const cris_jpeg_info* info, short* coef) {
    CRIS_CHECK_ARG(data && info && coef, "null argument");
    Tables T;
    cris_jpeg_info again;
    if (int rc = parse_header(data, nbytes, &again, &T)) return rc;
    if (again.coef_count != info->coef_count || again.scan_offset != info->scan_offset || again.width != info->width ||
        again.height != info->height)
        JERR("cris_jpeg_decode_coefficients: info does not belong to this file");
    if (again.multiscan) return decode_general(data, nbytes, again, coef);
    return decode_scan(data, nbytes, again, T, coef);
}

extern "C" int cris_jpeg_decode_coefficients_batch(int n, const unsigned char* const* data, const size_t* nbytes, const cris_jpeg_info* infos,
                                                   short* const* coefs, int n_threads) {
    CRIS_CHECK_ARG(n > 0 && data && nbytes && infos && coefs, "bad arguments");
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n;
    std::atomic<int> next(0);
    std::vector<std::string> errs(n);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            if (cris_jpeg_decode_coefficients(data[i], nbytes[i], &infos[i], coefs[i]) != 0) errs[i] = cris_last_error();   // this thread's message
        }
    };
    if (n_threads == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    int bad = -1;
    for (int i = 0; i < n; ++i)
        if (!errs[i].empty()) { bad = i; break; }
    if (bad >= 0) {
        cris_set_error("image %d: %s", bad, errs[bad].c_str());
        return -1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------
// grid: (ceil(max_blocks / 32), n images); 256 threads = 32 blocks of 8 lanes
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const cris_jpeg_image* __restrict__ tab) {
    __shared__ int ws[32][64 + 1];
    const cris_jpeg_image& im = tab[blockIdx.y];
    const cris_jpeg_info& I = im.info;
    const int lb = threadIdx.x >> 3, c8 = threadIdx.x & 7;
    const int blk = blockIdx.x * 32 + lb;                         // block index over all components of this image
    const bool live = blk < I.total_blocks;
    int comp = 0, rel = blk;
    if (live) {
        const int n0 = I.blocks_w[0] * I.blocks_h[0];
        if (rel >= n0) {
            rel -= n0;
            comp = 1;
            const int n1 = I.blocks_w[1] * I.blocks_h[1];
            if (rel >= n1) { rel -= n1; comp = 2; }
        }
        // pass 1: column c8
        const short* cf = im.coef + I.coef_offset[comp] + (long)rel * 64;
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)cf[r * 8 + c8] * (int)I.quant[comp][r * 8 + c8];
        cris_jpeg::idct8<11>(in, out);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r * 8 + c8] = out[r];
    }
    __syncthreads();
    if (live) {
        // pass 2: row c8
        int in[8], out[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) in[c] = ws[lb][c8 * 8 + c];
        cris_jpeg::idct8<18>(in, out);
        const int bw = I.blocks_w[comp];
        const int by = rel / bw, bx = rel - by * bw;
        unsigned char px[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) px[c] = cris_jpeg::idct_range_limit(out[c]);
        uint2 w;
        w.x = px[0] | (px[1] << 8) | (px[2] << 16) | ((unsigned)px[3] << 24);
        w.y = px[4] | (px[5] << 8) | (px[6] << 16) | ((unsigned)px[7] << 24);
        unsigned char* plane = im.planes + I.plane_offset[comp];
        *reinterpret_cast<uint2*>(plane + ((long)(by * 8 + c8) * (bw * 8) + bx * 8)) = w;
    }
}

// grid: (ceil(max_pixels / 256), n images)
__global__ __launch_bounds__(256) void jpeg_color_kernel(const cris_jpeg_image* __restrict__ tab) {
    const cris_jpeg_image& im = tab[blockIdx.y];
    const cris_jpeg_info& I = im.info;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)I.width * I.height) return;
    const int y = (int)(idx / I.width), x = (int)(idx - (long)y * I.width);
    const int lum = im.planes[I.plane_offset[0] + (long)y * (I.blocks_w[0] * 8) + x];
    unsigned char* o = im.rgb + idx * 3;
    if (I.ncomp == 1) {
        o[0] = o[1] = o[2] = (unsigned char)lum;
        return;
    }
    const int cb = cris_jpeg::chroma_at(im.planes + I.plane_offset[1], I.blocks_w[1] * 8, I.down_w[1], I.down_h[1], I.hmax, I.vmax, x, y);
    const int cr = cris_jpeg::chroma_at(im.planes + I.plane_offset[2], I.blocks_w[2] * 8, I.down_w[2], I.down_h[2], I.hmax, I.vmax, x, y);
    unsigned char rgb[3];
    cris_jpeg::ycc_to_rgb(lum, cb, cr, rgb);
    o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2];
}

extern "C" int cris_jpeg_reconstruct(const cris_jpeg_image* dev_table, int n, int max_blocks, long max_pixels, void* stream) {
    CRIS_CHECK_ARG(dev_table && n > 0 && n <= 65535 && max_blocks > 0 && max_pixels > 0, "bad arguments");
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(cris_cdiv(max_blocks, 32), n), dim3(256), 0, (hipStream_t)stream, dev_table);
    CRIS_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(cris_cdiv(max_pixels, 256), n), dim3(256), 0, (hipStream_t)stream, dev_table);
    CRIS_LAUNCH_CHECK();
    return 0;
}
