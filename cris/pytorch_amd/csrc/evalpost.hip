// Evaluation post-processing on the GPU (reference engine/engine.py:100-123 `validate`, :171-188 `inference`): sigmoid +
// bicubic (align_corners) upsample of the logits to the network input size, inverse affine warp to the original image size
// with OpenCV's INTER_CUBIC arithmetic, threshold and intersection / union counts.  The reference does steps 3-4 per sample
// on the CPU (cv2 + numpy after a device-to-host copy); here the prediction never leaves HBM.  HBM-bound elementwise kernels;
// the arithmetic order follows torch's upsample_bicubic2d and cv::warpAffine so that results can be compared value by value
// (oracle/eval_post.py; the cv2 step is restated from the published algorithm - cv2 is not available here).
#include "common.h"
#include "../../../include/cris_hip.h"

// No fused multiply-adds in this file: the kernels restate torch / OpenCV float arithmetic operation by operation, and hipcc
// contracts a*b+c by default (HIP's __fmul_rn / __fadd_rn are plain operators defined in a header OUTSIDE this pragma: their
// results still get fused; the ep_* helpers below are inside it).
#pragma clang fp contract(off)
__device__ __forceinline__ float ep_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float ep_add(float a, float b) { return a + b; }
__device__ __forceinline__ double ep_dmul(double a, double b) { return a * b; }
__device__ __forceinline__ double ep_dadd(double a, double b) { return a + b; }

__device__ __forceinline__ float ep_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch upsample_bicubic2d (A = -0.75): weights of taps -1, 0, +1, +2 for the fractional offset t
__device__ __forceinline__ void ep_cubic_torch(float t, float* c) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

// out[b][Y][X] = bicubic(sigmoid(logits[b]))(Y, X), align_corners = True, border indices clamped
__global__ __launch_bounds__(256) void sigmoid_bicubic_kernel(const float* __restrict__ logits, int Bn, int h, int w, int H, int W,
                                                              float sy_scale, float sx_scale, float* __restrict__ out) {
    const long total = (long)Bn * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int X = (int)(idx % W);
        const int Y = (int)((idx / W) % H);
        const int b = (int)(idx / ((long)W * H));
        const float fy = (float)Y * sy_scale, fx = (float)X * sx_scale;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float cy[4], cx[4];
        ep_cubic_torch(fy - (float)iy, cy);
        ep_cubic_torch(fx - (float)ix, cx);
        const float* src = logits + (size_t)b * h * w;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), h - 1);
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(ix - 1 + j, 0), w - 1);
                const float t = ep_mul(ep_sigmoid(src[yy * w + xx]), cx[j]);
                r = j == 0 ? t : ep_add(r, t);
            }
            const float t = ep_mul(r, cy[i]);
            acc = i == 0 ? t : ep_add(acc, t);
        }
        out[idx] = acc;
    }
}

extern "C" int cris_sigmoid_bicubic_up(const float* logits, int Bn, int h, int w, int H, int W, float* out, void* stream) {
    CRIS_CHECK_ARG(logits && out && Bn > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bad args");
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipLaunchKernelGGL(sigmoid_bicubic_kernel, dim3(cris_grid_1d((long)Bn * H * W, 256)), dim3(256), 0, (hipStream_t)stream, logits, Bn,
                       h, w, H, W, sy, sx, out);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ---- cv::warpAffine, INTER_CUBIC, BORDER_CONSTANT (imgwarp.cpp): fixed-point source coordinates (AB_BITS = 10) quantised to
// 1/32 pixel (INTER_BITS = 5), 4 x 4 weights = products of two entries of a 32 x 4 float table (A = -0.75) ----
#define EP_AB_BITS 10
#define EP_INTER_BITS 5
struct ep_warp_args {
    double m[6];                  // destination -> source map (the inverse of the matrix the caller passes, in double)
    float tab[32][4];             // cv::interpolateCubic coefficients of the 32 sub-pixel positions, built on the host
};

__global__ __launch_bounds__(256) void warp_affine_cubic_kernel(const float* __restrict__ src, int H, int W, const ep_warp_args a,
                                                                int w_out, int h_out, float border, float* __restrict__ dst) {
    const long total = (long)w_out * h_out;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w_out), y = (int)(idx / w_out);
        const double AB = (double)(1 << EP_AB_BITS);
        const int round_delta = (1 << EP_AB_BITS) / (1 << EP_INTER_BITS) / 2;
        // (explicit _rn operations: no fused multiply-add, so that the fixed-point coordinates equal a host evaluation bit for bit)
        const long adelta = (long)rint(ep_dmul(ep_dmul(a.m[0], (double)x), AB)), bdelta = (long)rint(ep_dmul(ep_dmul(a.m[3], (double)x), AB));
        const long X0 = (long)rint(ep_dmul(ep_dadd(ep_dmul(a.m[1], (double)y), a.m[2]), AB)) + round_delta;
        const long Y0 = (long)rint(ep_dmul(ep_dadd(ep_dmul(a.m[4], (double)y), a.m[5]), AB)) + round_delta;
        const long Xq = (X0 + adelta) >> (EP_AB_BITS - EP_INTER_BITS), Yq = (Y0 + bdelta) >> (EP_AB_BITS - EP_INTER_BITS);
        const int sx = (int)(Xq >> EP_INTER_BITS) - 1, sy = (int)(Yq >> EP_INTER_BITS) - 1;
        const int fx = (int)(Xq & 31), fy = (int)(Yq & 31);
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int yy = sy + ky;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int xx = sx + kx;
                const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
                const float v = inside ? src[(size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)] : border;
                const float wgt = ep_mul(a.tab[fy][ky], a.tab[fx][kx]);
                acc = ep_add(acc, ep_mul(v, wgt));
            }
        }
        dst[idx] = acc;
    }
}

extern "C" int cris_warp_affine_cubic(const float* src, int H, int W, const double* mat, int w_out, int h_out, float border, float* dst,
                                      void* stream) {
    CRIS_CHECK_ARG(src && mat && dst && H > 0 && W > 0 && w_out > 0 && h_out > 0, "bad args");
    ep_warp_args a;
    // cv::invertAffineTransform (the caller passes the matrix cv2.warpAffine is given, without WARP_INVERSE_MAP)
    double D = mat[0] * mat[4] - mat[1] * mat[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = mat[4] * D, A22 = mat[0] * D, A12 = -mat[1] * D, A21 = -mat[3] * D;
    a.m[0] = A11; a.m[1] = A12; a.m[2] = -A11 * mat[2] - A12 * mat[5];
    a.m[3] = A21; a.m[4] = A22; a.m[5] = -A21 * mat[2] - A22 * mat[5];
    const volatile float A = -0.75f;                            // (volatile: keep the float operations separate, no contraction)
    for (int i = 0; i < 32; ++i) {
        const volatile float t = (float)i / 32.0f;
        volatile float u = t + 1.0f, c0, c1, c2, q;
        q = A * u; q = q - 5.0f * A; q = q * u; q = q + 8.0f * A; q = q * u; c0 = q - 4.0f * A;
        q = (A + 2.0f) * t; q = q - (A + 3.0f); q = q * t; q = q * t; c1 = q + 1.0f;
        u = 1.0f - t;
        q = (A + 2.0f) * u; q = q - (A + 3.0f); q = q * u; q = q * u; c2 = q + 1.0f;
        q = 1.0f - c0; q = q - c1; q = q - c2;
        a.tab[i][0] = c0; a.tab[i][1] = c1; a.tab[i][2] = c2; a.tab[i][3] = q;
    }
    hipLaunchKernelGGL(warp_affine_cubic_kernel, dim3(cris_grid_1d((long)w_out * h_out, 256)), dim3(256), 0, (hipStream_t)stream, src, H, W,
                       a, w_out, h_out, border, dst);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// counts[0] += #(pred > thr and mask != 0), counts[1] += #(pred > thr or mask != 0): integer atomics (exact, order-free)
__global__ __launch_bounds__(256) void threshold_iou_kernel(const float* __restrict__ pred, const float* __restrict__ mask, long n, float thr,
                                                            int* __restrict__ counts) {
    int inter = 0, uni = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const bool p = pred[i] > thr, m = mask[i] != 0.f;
        inter += (p && m) ? 1 : 0;
        uni += (p || m) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inter += __shfl_xor(inter, o, 64);
        uni += __shfl_xor(uni, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts, inter);
        atomicAdd(counts + 1, uni);
    }
}

extern "C" int cris_threshold_iou(const float* pred, const float* mask, long n, float thr, int* counts, void* stream) {
    CRIS_CHECK_ARG(pred && mask && counts && n > 0, "bad args");
    hipLaunchKernelGGL(threshold_iou_kernel, dim3(cris_grid_1d(n, 256, 512)), dim3(256), 0, (hipStream_t)stream, pred, mask, n, thr, counts);
    CRIS_LAUNCH_CHECK();
    return 0;
}
