// Input preprocessing on the GPU (reference utils/dataset.py:146-168 `RefDataset.__getitem__`, :207-221 `convert`): per
// sample  img = cv2.warpAffine(rgb_u8, mat, (S, S), INTER_CUBIC, borderValue = CLIP mean * 255),
//         mask = cv2.warpAffine(mask_u8, mat, (S, S), INTER_LINEAR, borderValue = 0) / 255,
//         img = (img.float() / 255 - mean) / std  as [3, S, S].
// The reference does this in 32 DataLoader worker processes per node; at > 500 samples/s per GPU that is ~4 k letterbox warps
// per second and node, so the decoded uint8 images are uploaded as they are (a 640x480 RGB image is 0.9 MB, a third of
// its float tensor) and one launch per BATCH produces the network input directly in HBM.  HBM-bound integer kernel: one
// thread per destination pixel, 16 + 4 byte taps, the three channels and the mask of a pixel share the coordinates.
//
// Arithmetic = OpenCV's 8-bit path, restated in oracle/input_pipe.py (cv2 itself is not available offline - unpinned):
// fixed-point destination->source coordinates (AB_BITS 10, 1/32 pixel), 16-bit weight tables x 2^15 whose rows sum to 2^15
// (cris_remap_tables_u8 builds them like cv::initInterTab2D), border taps read the border colour, (sum + 2^14) >> 15
// saturated to 8 bits.  All integer except the coordinate set-up (double, explicitly unfused), so results are bit exact
// against the oracle.  The normalisation is a 256-entry lookup per channel built on the host with the reference's float ops.
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "common.h"
#include "../../../include/cris_hip.h"

#pragma clang fp contract(off)
__device__ __forceinline__ double ip_dmul(double a, double b) { return a * b; }
__device__ __forceinline__ double ip_dadd(double a, double b) { return a + b; }

#define IP_AB_BITS 10
#define IP_INTER_BITS 5
#define IP_COEF_BITS 15

// block = 256 destination pixels of sample blockIdx.y
__global__ __launch_bounds__(256) void preprocess_batch_kernel(const cris_sample_desc* __restrict__ samples, int S_h, int S_w,
                                                               const short* __restrict__ tab_linear, const short* __restrict__ tab_cubic,
                                                               const float* __restrict__ lut_img, const float* __restrict__ lut_mask,
                                                               cris_u8x4 border, float* __restrict__ img_out, float* __restrict__ mask_out) {
    const int b = blockIdx.y;
    const cris_sample_desc s = samples[b];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S_h * S_w) return;
    const int x = idx % S_w, y = idx / S_w;
    const double AB = (double)(1 << IP_AB_BITS);
    const int round_delta = (1 << IP_AB_BITS) / (1 << IP_INTER_BITS) / 2;
    const long adelta = (long)rint(ip_dmul(ip_dmul(s.inv[0], (double)x), AB)), bdelta = (long)rint(ip_dmul(ip_dmul(s.inv[3], (double)x), AB));
    const long X0 = (long)rint(ip_dmul(ip_dadd(ip_dmul(s.inv[1], (double)y), s.inv[2]), AB)) + round_delta;
    const long Y0 = (long)rint(ip_dmul(ip_dadd(ip_dmul(s.inv[4], (double)y), s.inv[5]), AB)) + round_delta;
    const long Xq = (X0 + adelta) >> (IP_AB_BITS - IP_INTER_BITS), Yq = (Y0 + bdelta) >> (IP_AB_BITS - IP_INTER_BITS);
    const int bx = (int)(Xq >> IP_INTER_BITS), by = (int)(Yq >> IP_INTER_BITS);
    const int ent = (int)(Yq & 31) * 32 + (int)(Xq & 31);
    const int H = s.H, W = s.W;
    if (s.img != nullptr) {                                  // INTER_CUBIC, 3 interleaved channels
        const short* w = tab_cubic + ent * 16;
        int acc[3] = {0, 0, 0};
        const int cv[3] = {border.v[0], border.v[1], border.v[2]};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int yy = by - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int xx = bx - 1 + kx;
                const int wt = w[ky * 4 + kx];
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const unsigned char* p = s.img + ((size_t)yy * W + xx) * 3;
                    acc[0] += (int)p[0] * wt; acc[1] += (int)p[1] * wt; acc[2] += (int)p[2] * wt;
                } else {
                    acc[0] += cv[0] * wt; acc[1] += cv[1] * wt; acc[2] += cv[2] * wt;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v = min(max((acc[c] + (1 << (IP_COEF_BITS - 1))) >> IP_COEF_BITS, 0), 255);
            img_out[(((size_t)b * 3 + c) * S_h + y) * S_w + x] = lut_img[c * 256 + v];
        }
    }
    if (s.mask != nullptr && mask_out != nullptr) {          // INTER_LINEAR, 1 channel, border 0
        const short* w = tab_linear + ent * 4;
        int acc = 0;
#pragma unroll
        for (int ky = 0; ky < 2; ++ky)
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const int yy = by + ky, xx = bx + kx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc += (int)s.mask[(size_t)yy * W + xx] * (int)w[ky * 2 + kx];
            }
        const int v = min(max((acc + (1 << (IP_COEF_BITS - 1))) >> IP_COEF_BITS, 0), 255);
        mask_out[((size_t)b * S_h + y) * S_w + x] = lut_mask[v];
    }
}

extern "C" int cris_preprocess_batch(const cris_sample_desc* samples_dev, int n, int S_h, int S_w, const short* tab_linear,
                                     const short* tab_cubic, const float* lut_img, const float* lut_mask, const unsigned char* border_rgb,
                                     float* img_out, float* mask_out, void* stream) {
    CRIS_CHECK_ARG(samples_dev && n > 0 && S_h > 0 && S_w > 0 && tab_linear && tab_cubic && lut_img && lut_mask && border_rgb && img_out,
                   "bad args");
    cris_u8x4 bd;
    bd.v[0] = border_rgb[0]; bd.v[1] = border_rgb[1]; bd.v[2] = border_rgb[2]; bd.v[3] = 0;
    hipLaunchKernelGGL(preprocess_batch_kernel, dim3(cris_cdiv(S_h * S_w, 256), n), dim3(256), 0, (hipStream_t)stream, samples_dev, S_h, S_w,
                       tab_linear, tab_cubic, lut_img, lut_mask, bd, img_out, mask_out);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ---- host helpers -------------------------------------------------------------------------------------------------

// cv::invertAffineTransform in double: the destination->source map cv2.warpAffine applies when it is given `mat`
extern "C" int cris_invert_affine(const double* mat, double* inv) {
    CRIS_CHECK_ARG(mat && inv, "bad args");
    double D = mat[0] * mat[4] - mat[1] * mat[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = mat[4] * D, A22 = mat[0] * D, A12 = -mat[1] * D, A21 = -mat[3] * D;
    inv[0] = A11; inv[1] = A12; inv[2] = -A11 * mat[2] - A12 * mat[5];
    inv[3] = A21; inv[4] = A22; inv[5] = -A21 * mat[2] - A22 * mat[5];
    return 0;
}

static short ip_sat_short(float v) {                         // saturate_cast<short>(float): cvRound (ties to even), saturated
    const long r = lrintf(v);
    return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

// cv::initInterTab2D(method, fixpt = true) for INTER_LINEAR ([1024][4]) and INTER_CUBIC ([1024][16]): entry fy*32 + fx, tap
// ky*ksize + kx.  The sum correction looks at the 2x2 window that starts at tap (ksize/2, ksize/2), like the original - for
// the bilinear table that window lies in the entries that follow (not yet filled: zero).
static void ip_table(int ksize, const float (*c1d)[4], short* out) {
    const int n = 32 * 32, kk = ksize * ksize;
    short* buf = (short*)calloc((size_t)n * kk + 4 * ksize + 4, sizeof(short));
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            short* it = buf + (size_t)(i * 32 + j) * kk;
            int isum = 0;
            for (int k1 = 0; k1 < ksize; ++k1)
                for (int k2 = 0; k2 < ksize; ++k2) {
                    const volatile float v = c1d[i][k1] * c1d[j][k2];
                    const volatile float vs = v * 32768.0f;
                    it[k1 * ksize + k2] = ip_sat_short(vs);
                    isum += it[k1 * ksize + k2];
                }
            if (isum != 32768) {
                const int diff = isum - 32768, k0 = ksize / 2;
                int Mk1 = k0, Mk2 = k0, mk1 = k0, mk2 = k0;
                for (int k1 = k0; k1 < k0 + 2; ++k1)
                    for (int k2 = k0; k2 < k0 + 2; ++k2) {
                        if (it[k1 * ksize + k2] < it[mk1 * ksize + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (it[k1 * ksize + k2] > it[Mk1 * ksize + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) it[Mk1 * ksize + Mk2] = (short)(it[Mk1 * ksize + Mk2] - diff);
                else it[mk1 * ksize + mk2] = (short)(it[mk1 * ksize + mk2] - diff);
            }
        }
    memcpy(out, buf, (size_t)n * kk * sizeof(short));
    free(buf);
}

extern "C" int cris_remap_tables_u8(short* tab_linear, short* tab_cubic) {
    CRIS_CHECK_ARG(tab_linear && tab_cubic, "bad args");
    static float lin[32][4], cub[32][4];
    const volatile float A = -0.75f;                         // (volatile: every float operation separately rounded)
    for (int i = 0; i < 32; ++i) {
        const volatile float t = (float)i * (1.0f / 32.0f);
        lin[i][0] = 1.0f - t; lin[i][1] = t; lin[i][2] = lin[i][3] = 0.f;
        volatile float u = t + 1.0f, c0, c1, c2, q;
        q = A * u; q = q - 5.0f * A; q = q * u; q = q + 8.0f * A; q = q * u; c0 = q - 4.0f * A;
        q = (A + 2.0f) * t; q = q - (A + 3.0f); q = q * t; q = q * t; c1 = q + 1.0f;
        u = 1.0f - t;
        q = (A + 2.0f) * u; q = q - (A + 3.0f); q = q * u; q = q * u; c2 = q + 1.0f;
        q = 1.0f - c0; q = q - c1; q = q - c2;
        cub[i][0] = c0; cub[i][1] = c1; cub[i][2] = c2; cub[i][3] = q;
    }
    ip_table(2, lin, tab_linear);
    ip_table(4, cub, tab_cubic);
    return 0;
}
