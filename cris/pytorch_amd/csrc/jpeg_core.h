// Per-block / per-pixel arithmetic of the JPEG reconstruction kernels (csrc/jpeg.hip), kept free of any HIP construct so that
// the SAME source can be compiled by g++ and checked on the CPU against the oracle (tests/test_jpeg_host.py builds
// tests/jpeg_core_probe.cpp around this header).  Everything here is integer arithmetic and must be bit-exact with
// libjpeg's decoder at its default settings - what cv2.imdecode runs for the reference's loader (utils/dataset.py:127-129):
//   * jidctint.c jpeg_idct_islow (dct_method JDCT_ISLOW): CONST_BITS 13, PASS1_BITS 2, JLONG (64-bit) intermediates;
//   * jdsample.c h2v1_fancy_upsample / h2v2_fancy_upsample (do_fancy_upsampling, only when downsampled_width > 2,
//     otherwise plain replication), rows beyond a component's real rows = its edge rows (jdmainct.c context rows);
//   * jdcolor.c ycc_rgb_convert: 16-bit fixed-point tables, FIX(1.40200) = 91881, FIX(1.77200) = 116130,
//     FIX(0.71414) = 46802, FIX(0.34414) = 22554, ONE_HALF added once on the Cb side of G.
#pragma once
#ifndef CRIS_HD
#define CRIS_HD __host__ __device__ __forceinline__
#endif

namespace cris_jpeg {

typedef long long jlong;                         // libjpeg's JLONG is `long` (64-bit on LP64)

#define CRIS_JFIX_0_298631336 2446
#define CRIS_JFIX_0_390180644 3196
#define CRIS_JFIX_0_541196100 4433
#define CRIS_JFIX_0_765366865 6270
#define CRIS_JFIX_0_899976223 7373
#define CRIS_JFIX_1_175875602 9633
#define CRIS_JFIX_1_501321110 12299
#define CRIS_JFIX_1_847759065 15137
#define CRIS_JFIX_1_961570560 16069
#define CRIS_JFIX_2_053119869 16819
#define CRIS_JFIX_2_562915447 20995
#define CRIS_JFIX_3_072711026 25172

// one 8-point pass of the islow inverse DCT; SHIFT = CONST_BITS - PASS1_BITS (11) for the column pass, CONST_BITS +
// PASS1_BITS + 3 (18) for the row pass.  in[] / out[] in natural order 0..7.
template <int SHIFT>
CRIS_HD void idct8(const int* in, int* out) {
    const jlong rnd = (jlong)1 << (SHIFT - 1);
    jlong z2 = in[2], z3 = in[6];
    jlong z1 = (z2 + z3) * CRIS_JFIX_0_541196100;
    jlong tmp2 = z1 + z3 * (-CRIS_JFIX_1_847759065);
    jlong tmp3 = z1 + z2 * CRIS_JFIX_0_765366865;
    jlong tmp0 = ((jlong)in[0] + in[4]) * 8192;       // << CONST_BITS
    jlong tmp1 = ((jlong)in[0] - in[4]) * 8192;
    const jlong tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    jlong z4 = tmp1 + tmp3;
    const jlong z5 = (z3 + z4) * CRIS_JFIX_1_175875602;
    tmp0 *= CRIS_JFIX_0_298631336; tmp1 *= CRIS_JFIX_2_053119869; tmp2 *= CRIS_JFIX_3_072711026; tmp3 *= CRIS_JFIX_1_501321110;
    z1 *= -CRIS_JFIX_0_899976223; z2 *= -CRIS_JFIX_2_562915447;
    z3 = z3 * (-CRIS_JFIX_1_961570560) + z5;
    z4 = z4 * (-CRIS_JFIX_0_390180644) + z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = (int)((tmp10 + tmp3 + rnd) >> SHIFT);
    out[7] = (int)((tmp10 - tmp3 + rnd) >> SHIFT);
    out[1] = (int)((tmp11 + tmp2 + rnd) >> SHIFT);
    out[6] = (int)((tmp11 - tmp2 + rnd) >> SHIFT);
    out[2] = (int)((tmp12 + tmp1 + rnd) >> SHIFT);
    out[5] = (int)((tmp12 - tmp1 + rnd) >> SHIFT);
    out[3] = (int)((tmp13 + tmp0 + rnd) >> SHIFT);
    out[4] = (int)((tmp13 - tmp0 + rnd) >> SHIFT);
}

// IDCT_range_limit[x & RANGE_MASK] of jdmaster.c prepare_range_limit_table: clamp(x + 128, 0, 255) for -512 <= x < 512, and
// the table's own wrap-around for values a corrupt stream can produce beyond that
CRIS_HD unsigned char idct_range_limit(int x) {
    const int i = x & 1023;
    return (unsigned char)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}

CRIS_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// chroma sample of output pixel (x, y) from a component plane `p` (row stride `pw`) whose real extent is dw x dh samples.
// hs / vs: horizontal / vertical replication factor of this component against the image (1 or 2).
CRIS_HD int chroma_at(const unsigned char* p, int pw, int dw, int dh, int hs, int vs, int x, int y) {
    if (hs == 1) return p[y * pw + x];                                    // 4:4:4: full size (vs == 1 as well)
    const int i = x >> 1, odd = x & 1;
    if (dw <= 2) return p[(vs == 2 ? (y >> 1) : y) * pw + i];             // jinit_upsampler: no fancy filter for <= 2 columns
    const int il = i > 0 ? i - 1 : 0, ir = i < dw - 1 ? i + 1 : dw - 1;
    if (vs == 1) {                                                        // h2v1 (4:2:2)
        const unsigned char* r = p + y * pw;
        if (!odd) return i == 0 ? r[0] : (3 * r[i] + r[il] + 1) >> 2;
        return i == dw - 1 ? r[i] : (3 * r[i] + r[ir] + 2) >> 2;
    }
    // h2v2 (4:2:0): nearest row j, next nearest above (even output rows) / below (odd), clamped to the real rows
    const int j = y >> 1;
    const int jo = (y & 1) ? (j < dh - 1 ? j + 1 : dh - 1) : (j > 0 ? j - 1 : 0);
    const unsigned char* r0 = p + j * pw;
    const unsigned char* r1 = p + jo * pw;
    const int cs = 3 * r0[i] + r1[i];
    if (!odd) return (3 * cs + (3 * r0[il] + r1[il]) + 8) >> 4;
    return (3 * cs + (3 * r0[ir] + r1[ir]) + 7) >> 4;
}

CRIS_HD void ycc_to_rgb(int y, int cb, int cr, unsigned char* rgb) {
    cb -= 128; cr -= 128;
    const int half = 1 << 15;
    const int r = y + ((91881 * cr + half) >> 16);
    const int g = y + ((-22554 * cb + half - 46802 * cr) >> 16);
    const int b = y + ((116130 * cb + half) >> 16);
    rgb[0] = (unsigned char)clampi(r, 0, 255);
    rgb[1] = (unsigned char)clampi(g, 0, 255);
    rgb[2] = (unsigned char)clampi(b, 0, 255);
}

}  // namespace cris_jpeg
