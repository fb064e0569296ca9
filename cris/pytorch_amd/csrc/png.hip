// 8-bit grayscale PNG decoding on the host (include/cris_hip.h "PNG masks"): the segmentation mask of an LMDB record
// (reference utils/dataset.py:148-149 `cv2.imdecode(np.frombuffer(ref['mask'], np.uint8), cv2.IMREAD_GRAYSCALE)`; the files are
// written by tools/data_process.py:115-117 `cv2.imwrite(.., mask * 255)`: 8-bit gray, non-interlaced).  Lossless: chunk
// walk, zlib stream (RFC 1950) around DEFLATE (RFC 1951: stored / fixed / dynamic Huffman blocks), the five PNG row filters.
// All of it is serial byte work per file - it stays on a host thread; the decoded mask (one byte per pixel, a few hundred kB)
// then goes to the GPU with the image for the letter-box warp (cris_preprocess_batch).  No HIP in this file.
#include "common.h"
#include "../../../include/cris_hip.h"
#include <string.h>
#include <vector>

namespace {

#define PERR(...)                    \
    do {                             \
        cris_set_error(__VA_ARGS__); \
        return -1;                   \
    } while (0)

struct Inflate {
    const unsigned char* in;
    size_t n, pos = 0;
    unsigned bitbuf = 0;
    int bitcnt = 0;
    std::vector<unsigned char>& out;
    size_t limit;
    Inflate(const unsigned char* d, size_t n_, std::vector<unsigned char>& o, size_t lim) : in(d), n(n_), out(o), limit(lim) {}
    // LSB-first bit reader; -1 past the end
    int bits(int need) {
        long val = bitbuf;
        while (bitcnt < need) {
            if (pos >= n) return -1;
            val |= (long)in[pos++] << bitcnt;
            bitcnt += 8;
        }
        bitbuf = (unsigned)(val >> need);
        bitcnt -= need;
        return (int)(val & ((1L << need) - 1));
    }
};

struct Huff {
    short count[16];
    short symbol[288];
};

// canonical code from code lengths; returns 0 for a complete code, >0 incomplete, <0 over-subscribed
int construct(Huff& h, const short* length, int n) {
    memset(h.count, 0, sizeof(h.count));
    for (int s = 0; s < n; ++s) h.count[length[s]]++;
    if (h.count[0] == n) return 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= h.count[len];
        if (left < 0) return left;
    }
    short offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + h.count[len];
    for (int s = 0; s < n; ++s)
        if (length[s] != 0) h.symbol[offs[length[s]]++] = (short)s;
    return left;
}

int decode(Inflate& s, const Huff& h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        const int b = s.bits(1);
        if (b < 0) return -1;
        code |= b;
        const int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

const short kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const short kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const short kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const short kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

int codes(Inflate& s, const Huff& lencode, const Huff& distcode) {
    for (;;) {
        int sym = decode(s, lencode);
        if (sym < 0) return -1;
        if (sym < 256) {
            if (s.out.size() >= s.limit) return -2;
            s.out.push_back((unsigned char)sym);
        } else if (sym == 256) {
            return 0;
        } else {
            sym -= 257;
            if (sym >= 29) return -1;
            int eb = s.bits(kLenExtra[sym]);
            if (eb < 0) return -1;
            const int len = kLenBase[sym] + eb;
            const int ds = decode(s, distcode);
            if (ds < 0 || ds >= 30) return -1;
            eb = s.bits(kDistExtra[ds]);
            if (eb < 0) return -1;
            const size_t dist = (size_t)kDistBase[ds] + eb;
            if (dist > s.out.size()) return -1;
            if (s.out.size() + len > s.limit) return -2;
            size_t from = s.out.size() - dist;
            for (int i = 0; i < len; ++i) s.out.push_back(s.out[from + i]);      // may overlap: byte by byte
        }
    }
}

int inflate_all(Inflate& s) {
    static Huff fixed_len, fixed_dist;
    static const bool fixed_ready = []() {
        short l[288];
        int i = 0;
        for (; i < 144; ++i) l[i] = 8;
        for (; i < 256; ++i) l[i] = 9;
        for (; i < 280; ++i) l[i] = 7;
        for (; i < 288; ++i) l[i] = 8;
        construct(fixed_len, l, 288);
        for (i = 0; i < 30; ++i) l[i] = 5;
        construct(fixed_dist, l, 30);
        return true;
    }();
    (void)fixed_ready;
    int last;
    do {
        last = s.bits(1);
        const int type = s.bits(2);
        if (last < 0 || type < 0) return -1;
        if (type == 0) {
            s.bitbuf = 0; s.bitcnt = 0;
            if (s.pos + 4 > s.n) return -1;
            const unsigned len = s.in[s.pos] | (s.in[s.pos + 1] << 8), nlen = s.in[s.pos + 2] | (s.in[s.pos + 3] << 8);
            s.pos += 4;
            if ((len ^ 0xFFFF) != nlen || s.pos + len > s.n) return -1;
            if (s.out.size() + len > s.limit) return -2;
            s.out.insert(s.out.end(), s.in + s.pos, s.in + s.pos + len);
            s.pos += len;
        } else if (type == 1) {
            if (int rc = codes(s, fixed_len, fixed_dist)) return rc;
        } else if (type == 2) {
            static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            const int nlen = s.bits(5), ndist = s.bits(5), ncode = s.bits(4);
            if (nlen < 0 || ndist < 0 || ncode < 0 || nlen + 257 > 286 || ndist + 1 > 30) return -1;
            short lengths[320];
            memset(lengths, 0, sizeof(lengths));
            for (int i = 0; i < ncode + 4; ++i) {
                const int v = s.bits(3);
                if (v < 0) return -1;
                lengths[order[i]] = (short)v;
            }
            Huff lencode, distcode;
            if (construct(lencode, lengths, 19) != 0) return -1;
            const int total = nlen + 257 + ndist + 1;
            int idx = 0;
            short ll[320];
            while (idx < total) {
                int sym = decode(s, lencode);
                if (sym < 0) return -1;
                if (sym < 16) {
                    ll[idx++] = (short)sym;
                } else {
                    int rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) return -1;
                        val = ll[idx - 1];
                        rep = s.bits(2);
                        if (rep < 0) return -1;
                        rep += 3;
                    } else if (sym == 17) {
                        rep = s.bits(3);
                        if (rep < 0) return -1;
                        rep += 3;
                    } else {
                        rep = s.bits(7);
                        if (rep < 0) return -1;
                        rep += 11;
                    }
                    if (idx + rep > total) return -1;
                    while (rep--) ll[idx++] = (short)val;
                }
            }
            if (ll[256] == 0) return -1;
            int err = construct(lencode, ll, nlen + 257);
            if (err < 0 || (err > 0 && nlen + 257 - lencode.count[0] != 1)) return -1;
            err = construct(distcode, ll + nlen + 257, ndist + 1);
            if (err < 0 || (err > 0 && ndist + 1 - distcode.count[0] != 1)) return -1;
            if (int rc = codes(s, lencode, distcode)) return rc;
        } else {
            return -1;
        }
    } while (!last);
    return 0;
}

inline unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

int parse_ihdr(const unsigned char* d, size_t n, int* w, int* h) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 + 25 || memcmp(d, sig, 8) != 0) PERR("cris_png: not a PNG file");
    if (be32(d + 8) != 13 || memcmp(d + 12, "IHDR", 4) != 0) PERR("cris_png: IHDR is not the first chunk");
    const unsigned W = be32(d + 16), H = be32(d + 20);
    const int depth = d[24], ctype = d[25], comp = d[26], filt = d[27], lace = d[28];
    if (W == 0 || H == 0 || (unsigned long)W * H > (1UL << 28)) PERR("cris_png: bad image size %u x %u", W, H);
    if (depth != 8 || ctype != 0) PERR("cris_png: only 8-bit grayscale masks are supported (bit depth %d, colour type %d)", depth, ctype);
    if (comp != 0 || filt != 0 || lace != 0) PERR("cris_png: interlaced or non-standard PNG is not supported");
    *w = (int)W;
    *h = (int)H;
    return 0;
}

}  // namespace

extern "C" int cris_png_gray8_size(const unsigned char* data, size_t nbytes, int* width, int* height) {
    CRIS_CHECK_ARG(data && width && height, "null argument");
    return parse_ihdr(data, nbytes, width, height);
}

extern "C" int cris_png_decode_gray8(const unsigned char* data, size_t nbytes, unsigned char* out, int width, int height) {
    CRIS_CHECK_ARG(data && out, "null argument");
    int w, h;
    if (int rc = parse_ihdr(data, nbytes, &w, &h)) return rc;
    if (w != width || h != height) PERR("cris_png_decode_gray8: the output buffer is %d x %d, the file %d x %d", width, height, w, h);
    // concatenate the IDAT chunks
    std::vector<unsigned char> z;
    size_t pos = 8;
    bool end = false;
    while (pos + 12 <= nbytes && !end) {
        const unsigned len = be32(data + pos);
        if (pos + 12 + (size_t)len > nbytes) PERR("cris_png: truncated chunk");
        const unsigned char* type = data + pos + 4;
        if (memcmp(type, "IDAT", 4) == 0) z.insert(z.end(), data + pos + 8, data + pos + 8 + len);
        else if (memcmp(type, "IEND", 4) == 0) end = true;
        pos += 12 + (size_t)len;
    }
    if (z.size() < 6) PERR("cris_png: no image data");
    if ((z[0] & 0x0F) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) PERR("cris_png: bad zlib header");
    const size_t want = (size_t)h * ((size_t)w + 1);
    std::vector<unsigned char> raw;
    raw.reserve(want);
    Inflate st(z.data() + 2, z.size() - 2, raw, want);
    const int rc = inflate_all(st);
    if (rc != 0 || raw.size() != want) PERR("cris_png: corrupt compressed data (%s)", rc == -2 ? "more data than the image holds" : "invalid DEFLATE stream or short data");
    // Adler-32 of the uncompressed bytes (RFC 1950)
    if (st.pos + 4 <= st.n) {
        unsigned a = 1, b = 0;
        for (size_t i = 0; i < raw.size(); ++i) {
            a = (a + raw[i]) % 65521u;
            b = (b + a) % 65521u;
        }
        if (((b << 16) | a) != be32(st.in + st.pos)) PERR("cris_png: Adler-32 mismatch");
    }
    // the five row filters (PNG 1.2 section 6; one byte per pixel)
    for (int y = 0; y < h; ++y) {
        const unsigned char* src = raw.data() + (size_t)y * (w + 1);
        unsigned char* row = out + (size_t)y * w;
        const unsigned char* up = y ? row - w : nullptr;
        const int f = src[0];
        ++src;
        switch (f) {
            case 0: memcpy(row, src, w); break;
            case 1:
                for (int x = 0; x < w; ++x) row[x] = (unsigned char)(src[x] + (x ? row[x - 1] : 0));
                break;
            case 2:
                for (int x = 0; x < w; ++x) row[x] = (unsigned char)(src[x] + (up ? up[x] : 0));
                break;
            case 3:
                for (int x = 0; x < w; ++x) row[x] = (unsigned char)(src[x] + (((x ? row[x - 1] : 0) + (up ? up[x] : 0)) >> 1));
                break;
            case 4:
                for (int x = 0; x < w; ++x) {
                    const int a = x ? row[x - 1] : 0, b = up ? up[x] : 0, c = (x && up) ? up[x - 1] : 0;
                    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                    row[x] = (unsigned char)(src[x] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
                }
                break;
            default: PERR("cris_png: unknown row filter %d", f);
        }
    }
    return 0;
}
