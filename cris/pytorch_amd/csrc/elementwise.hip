// Data-movement / elementwise / small reduction kernels of the CRIS path (all HBM- or latency-bound).
// bf16 tensors are accessed as 16-byte vectors (8 channels per lane) wherever the layout allows.
#include "common.h"
#include "../../../include/cris_hip.h"

__device__ __forceinline__ void ld8f(const float* p, float* f) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void ld8bf(const bf16_t* p, float* f) { unpack8(*reinterpret_cast<const uint4*>(p), f); }
__device__ __forceinline__ void st8bf(bf16_t* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack8(f); }

#define GRID_STRIDE(idx, total) \
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (total); idx += (long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------
// stem im2col: fp32 NCHW -> bf16 [B*OH*OW][32], k = ci*9 + kh*3 + kw (3x3, stride 2, pad 1)
// ------------------------------------------------------------------------------------------------
__global__ void stem_im2col_kernel(const float* __restrict__ img, int Bn, int H, int W, int OH, int OW, bf16_t* __restrict__ out) {
    const long total = (long)Bn * OH * OW * 4;
    const bool small = total < (1L << 24);
    GRID_STRIDE(idx, total) {
        const cris_idx4 q = cris_split4(idx, 4, OW, OH, small);
        const int kc = q.cv, ow = q.x, oh = q.y, b = q.b;
        const long m = idx >> 2;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc * 8 + j;
            float x = 0.f;
            if (k < 27) {
                const int ci = k / 9, t = k - ci * 9;
                const int kh = t / 3, kw = t - kh * 3;
                const int ih = oh * 2 + kh - 1, iw = ow * 2 + kw - 1;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) x = img[(((size_t)b * 3 + ci) * H + ih) * W + iw];
            }
            v[j] = x;
        }
        st8bf(out + m * 32 + kc * 8, v);
    }
}

extern "C" int cris_stem_im2col(const float* img, int Bn, int H, int W, cris_bf16* out, void* stream) {
    CRIS_CHECK_ARG(img && out && Bn > 0 && H > 0 && W > 0, "bad args");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)Bn * OH * OW * 4;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3(cris_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, img, Bn, H, W, OH, OW, out);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 2x2 average pool
// ------------------------------------------------------------------------------------------------
__global__ void avgpool2_fwd_kernel(const bf16_t* __restrict__ x, int ldx, int xcoff, int Bn, int H, int W, int C,
                                    bf16_t* __restrict__ y, int ldy, int ycoff) {
    const int CV = C >> 3, OH = H >> 1, OW = W >> 1;
    const long total = (long)Bn * OH * OW * CV;
    const bool small = total < (1L << 24);
    GRID_STRIDE(idx, total) {
        const cris_idx4 q = cris_split4(idx, CV, OW, OH, small);
        const int cv = q.cv, ow = q.x, oh = q.y, b = q.b;
        const long mo = ((long)b * OH + oh) * OW + ow;
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                ld8bf(x + (((size_t)b * H + oh * 2 + dy) * W + ow * 2 + dx) * ldx + xcoff + cv * 8, t);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += t[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= 0.25f;
        st8bf(y + (size_t)mo * ldy + ycoff + cv * 8, o);
    }
}

extern "C" int cris_avgpool2_fwd(const cris_bf16* x, int ldx, int xcoff, int Bn, int H, int W, int C, cris_bf16* y,
                                 int ldy, int ycoff, void* stream) {
    CRIS_CHECK_ARG(x && y && !(H & 1) && !(W & 1) && !(C & 7) && !(ldx & 7) && !(ldy & 7) && !(xcoff & 7) && !(ycoff & 7), "bad args");
    const long total = (long)Bn * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(cris_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, xcoff,
                       Bn, H, W, C, y, ldy, ycoff);
    CRIS_LAUNCH_CHECK();
    return 0;
}

__global__ void avgpool2_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, int dycoff, int Bn, int H, int W, int C,
                                    bf16_t* __restrict__ dx, int lddx, int dxcoff, int accum) {
    const int CV = C >> 3, OH = H >> 1, OW = W >> 1;
    const long total = (long)Bn * H * W * CV;
    const bool small = total < (1L << 24);
    GRID_STRIDE(idx, total) {
        const cris_idx4 q = cris_split4(idx, CV, W, H, small);
        const int cv = q.cv, w = q.x, h = q.y, b = q.b;
        const long m = ((long)b * H + h) * W + w;
        float g[8];
        ld8bf(dy + (((size_t)b * OH + (h >> 1)) * OW + (w >> 1)) * lddy + dycoff + cv * 8, g);
        bf16_t* d = dx + (size_t)m * lddx + dxcoff + cv * 8;
        float o[8];
        if (accum) ld8bf(d, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (accum ? o[j] : 0.f) + 0.25f * g[j];
        st8bf(d, o);
    }
}

extern "C" int cris_avgpool2_bwd(const cris_bf16* dy, int lddy, int dycoff, int Bn, int H, int W, int C, cris_bf16* dx,
                                 int lddx, int dxcoff, int accum, void* stream) {
    CRIS_CHECK_ARG(dy && dx && !(H & 1) && !(W & 1) && !(C & 7) && !(lddx & 7) && !(lddy & 7) && !(dxcoff & 7) && !(dycoff & 7), "bad args");
    const long total = (long)Bn * H * W * (C / 8);
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(cris_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dycoff,
                       Bn, H, W, C, dx, lddx, dxcoff, accum);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// x2 bilinear upsample, align_corners=False:  src = max(0, (dst+0.5)/2 - 0.5)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void up2_src(int dst, int n_in, int& i0, int& i1, float& l1) {
    float s = (dst + 0.5f) * 0.5f - 0.5f;
    s = fmaxf(s, 0.f);
    i0 = (int)s;
    i1 = i0 < n_in - 1 ? i0 + 1 : i0;
    l1 = s - (float)i0;
}

__global__ void upsample2_fwd_kernel(const bf16_t* __restrict__ x, int ldx, int xcoff, int Bn, int H, int W, int C,
                                     bf16_t* __restrict__ y, int ldy, int ycoff) {
    const int CV = C >> 3, OH = H * 2, OW = W * 2;
    const long total = (long)Bn * OH * OW * CV;
    const bool small = total < (1L << 24);
    GRID_STRIDE(idx, total) {
        const cris_idx4 q = cris_split4(idx, CV, OW, OH, small);
        const int cv = q.cv, ow = q.x, oh = q.y, b = q.b;
        const long mo = ((long)b * OH + oh) * OW + ow;
        int y0, y1, x0, x1;
        float ly, lx;
        up2_src(oh, H, y0, y1, ly);
        up2_src(ow, W, x0, x1, lx);
        const bf16_t* base = x + (size_t)b * H * W * ldx + xcoff + cv * 8;
        float a[8], bb[8], c[8], d[8], o[8];
        ld8bf(base + ((size_t)y0 * W + x0) * ldx, a);
        ld8bf(base + ((size_t)y0 * W + x1) * ldx, bb);
        ld8bf(base + ((size_t)y1 * W + x0) * ldx, c);
        ld8bf(base + ((size_t)y1 * W + x1) * ldx, d);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = w00 * a[j] + w01 * bb[j] + w10 * c[j] + w11 * d[j];
        st8bf(y + (size_t)mo * ldy + ycoff + cv * 8, o);
    }
}

extern "C" int cris_upsample2_fwd(const cris_bf16* x, int ldx, int xcoff, int Bn, int H, int W, int C, cris_bf16* y,
                                  int ldy, int ycoff, void* stream) {
    CRIS_CHECK_ARG(x && y && !(C & 7) && !(ldx & 7) && !(ldy & 7) && !(xcoff & 7) && !(ycoff & 7), "bad args");
    const long total = (long)Bn * H * W * 4 * (C / 8);
    hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(cris_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, xcoff,
                       Bn, H, W, C, y, ldy, ycoff);
    CRIS_LAUNCH_CHECK();
    return 0;
}

__global__ void upsample2_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, int dycoff, int Bn, int H, int W, int C,
                                     bf16_t* __restrict__ dx, int lddx, int dxcoff, int accum) {
    // gather form: input pixel (iy,ix) collects from output rows/cols 2i-2 .. 2i+2 whose source taps hit it
    const int CV = C >> 3, OH = H * 2, OW = W * 2;
    const long total = (long)Bn * H * W * CV;
    const bool small = total < (1L << 24);
    GRID_STRIDE(idx, total) {
        const cris_idx4 q = cris_split4(idx, CV, W, H, small);
        const int cv = q.cv, ix = q.x, iy = q.y, b = q.b;
        const long m = ((long)b * H + iy) * W + ix;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bf16_t* base = dy + (size_t)b * OH * OW * lddy + dycoff + cv * 8;
        for (int oy = max(0, 2 * iy - 2); oy <= min(OH - 1, 2 * iy + 2); ++oy) {
            int y0, y1; float ly;
            up2_src(oy, H, y0, y1, ly);
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y1 == iy) wy += ly;
            if (wy == 0.f) continue;
            for (int ox = max(0, 2 * ix - 2); ox <= min(OW - 1, 2 * ix + 2); ++ox) {
                int x0, x1; float lx;
                up2_src(ox, W, x0, x1, lx);
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - lx;
                if (x1 == ix) wx += lx;
                if (wx == 0.f) continue;
                float g[8];
                ld8bf(base + ((size_t)oy * OW + ox) * lddy, g);
                const float w = wy * wx;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
            }
        }
        bf16_t* d = dx + (size_t)m * lddx + dxcoff + cv * 8;
        if (accum) {
            float o[8];
            ld8bf(d, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += o[j];
        }
        st8bf(d, acc);
    }
}

extern "C" int cris_upsample2_bwd(const cris_bf16* dy, int lddy, int dycoff, int Bn, int H, int W, int C, cris_bf16* dx,
                                  int lddx, int dxcoff, int accum, void* stream) {
    CRIS_CHECK_ARG(dy && dx && !(C & 7) && !(lddx & 7) && !(lddy & 7) && !(dxcoff & 7) && !(dycoff & 7), "bad args");
    const long total = (long)Bn * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(cris_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dycoff,
                       Bn, H, W, C, dx, lddx, dxcoff, accum);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// CoordConv channels: x = linspace(-1,1,W)[w], y = linspace(-1,1,H)[h]  (torch.linspace symmetric formula)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace_m1_1(int i, int n) {
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

__global__ void fill_coords_kernel(bf16_t* x, int ldx, int coff, int nfill, int Bn, int H, int W) {
    const long total = (long)Bn * H * W;
    GRID_STRIDE(m, total) {
        const int w = (int)(m % W);
        const int h = (int)((m / W) % H);
        bf16_t* d = x + (size_t)m * ldx + coff;
        d[0] = f2bf(linspace_m1_1(w, W));
        d[1] = f2bf(linspace_m1_1(h, H));
        for (int j = 2; j < nfill; ++j) d[j] = 0;
    }
}

extern "C" int cris_fill_coords(cris_bf16* x, int ldx, int coff, int nfill, int Bn, int H, int W, void* stream) {
    CRIS_CHECK_ARG(x && nfill >= 2, "bad args");
    hipLaunchKernelGGL(fill_coords_kernel, dim3(cris_grid_1d((long)Bn * H * W, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       coff, nfill, Bn, H, W);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adds / casts
// ------------------------------------------------------------------------------------------------
__global__ void add_bf16_kernel(const bf16_t* a, int lda, int acoff, const bf16_t* b, int ldb, int bcoff, bf16_t* y, int ldy,
                                int ycoff, int M, int C) {
    const int CV = C >> 3;
    const long total = (long)M * CV;
    GRID_STRIDE(idx, total) {
        const int cv = (int)(idx % CV);
        const long m = idx / CV;
        float u[8], v[8];
        ld8bf(a + (size_t)m * lda + acoff + cv * 8, u);
        if (b) {
            ld8bf(b + (size_t)m * ldb + bcoff + cv * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] += v[j];
        }
        st8bf(y + (size_t)m * ldy + ycoff + cv * 8, u);
    }
}

extern "C" int cris_add_bf16(const cris_bf16* a, int lda, int acoff, const cris_bf16* b, int ldb, int bcoff, cris_bf16* y,
                             int ldy, int ycoff, int M, int C, void* stream) {
    CRIS_CHECK_ARG(a && y && !(C & 7) && !(lda & 7) && !(ldy & 7) && !(acoff & 7) && !(ycoff & 7) && (!b || (!(ldb & 7) && !(bcoff & 7))), "bad args");
    hipLaunchKernelGGL(add_bf16_kernel, dim3(cris_grid_1d((long)M * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, a, lda,
                       acoff, b, ldb, bcoff, y, ldy, ycoff, M, C);
    CRIS_LAUNCH_CHECK();
    return 0;
}

__global__ void add_rowtable_kernel(const bf16_t* a, int lda, const float* table, int trows, bf16_t* y, int ldy, int M, int C) {
    const int CV = C >> 3;
    const long total = (long)M * CV;
    GRID_STRIDE(idx, total) {
        const int cv = (int)(idx % CV);
        const long m = idx / CV;
        float u[8], v[8];
        ld8bf(a + (size_t)m * lda + cv * 8, u);
        ld8f(table + (size_t)(m % trows) * C + cv * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] += v[j];
        st8bf(y + (size_t)m * ldy + cv * 8, u);
    }
}

extern "C" int cris_add_rowtable(const cris_bf16* a, int lda, const float* table, int trows, cris_bf16* y, int ldy, int M,
                                 int C, void* stream) {
    CRIS_CHECK_ARG(a && table && y && trows > 0 && !(C & 7) && !(lda & 7) && !(ldy & 7), "bad args");
    hipLaunchKernelGGL(add_rowtable_kernel, dim3(cris_grid_1d((long)M * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, a,
                       lda, table, trows, y, ldy, M, C);
    CRIS_LAUNCH_CHECK();
    return 0;
}

__global__ void cast_f32_bf16_kernel(const float* x, bf16_t* y, long n) {
    GRID_STRIDE(i, n) y[i] = f2bf(x[i]);
}
extern "C" int cris_cast_f32_bf16(const float* x, cris_bf16* y, long n, void* stream) {
    CRIS_CHECK_ARG(x && y && n > 0, "bad args");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cris_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void cast_bf16_f32_kernel(const bf16_t* x, float* y, long n, int accum) {
    GRID_STRIDE(i, n) y[i] = (accum ? y[i] : 0.f) + bf2f(x[i]);
}
extern "C" int cris_cast_bf16_f32(const cris_bf16* x, float* y, long n, int accum, void* stream) {
    CRIS_CHECK_ARG(x && y && n > 0, "bad args");
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(cris_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, accum);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// QuickGELU (reference model/clip.py:234-236) on a bf16 pre-activation, and its gradient
__global__ void quickgelu_fwd_kernel(const bf16_t* x, bf16_t* y, long nvec) {
    GRID_STRIDE(i, nvec) {
        float v[8];
        ld8bf(x + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
        st8bf(y + i * 8, v);
    }
}
extern "C" int cris_quickgelu_fwd(const cris_bf16* x, cris_bf16* y, long n, void* stream) {
    CRIS_CHECK_ARG(x && y && n > 0 && !(n & 7), "n must be a multiple of 8");
    hipLaunchKernelGGL(quickgelu_fwd_kernel, dim3(cris_grid_1d(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n / 8);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void quickgelu_bwd_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* dx, long nvec) {
    GRID_STRIDE(i, nvec) {
        float v[8], g[8];
        ld8bf(x + i * 8, v);
        ld8bf(dy + i * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float s = 1.f / (1.f + __expf(-1.702f * v[j]));
            g[j] *= s * (1.f + 1.702f * v[j] * (1.f - s));
        }
        st8bf(dx + i * 8, g);
    }
}
extern "C" int cris_quickgelu_bwd(const cris_bf16* x, const cris_bf16* dy, cris_bf16* dx, long n, void* stream) {
    CRIS_CHECK_ARG(x && dy && dx && n > 0 && !(n & 7), "n must be a multiple of 8");
    hipLaunchKernelGGL(quickgelu_bwd_kernel, dim3(cris_grid_1d(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n / 8);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// y(bf16) = dropout_mask(idx) ? x * 1/(1-p) : 0  (gradient of an output dropout applied to an fp32 stream)
__global__ void cast_drop_kernel(const float* x, bf16_t* y, long n, float scale, uint32_t thresh, uint32_t seed, uint32_t stream_id,
                                 const uint32_t* seed_dev) {
    const uint32_t key = cris_drop_key(seed + (seed_dev ? seed_dev[0] : 0u), stream_id);
    GRID_STRIDE(i, n) {
        float v = x[i];
        if (thresh) v = cris_keep(key, (uint32_t)i, thresh) ? v * scale : 0.f;
        y[i] = f2bf(v);
    }
}
extern "C" int cris_cast_f32_bf16_drop(const float* x, cris_bf16* y, long n, float drop_p, uint32_t drop_thresh, uint32_t seed,
                                       uint32_t stream_id, const uint32_t* seed_dev, void* stream) {
    CRIS_CHECK_ARG(x && y && n > 0 && n < (1L << 32), "bad args");
    const float scale = drop_thresh ? 1.f / (1.f - drop_p) : 1.f;
    hipLaunchKernelGGL(cast_drop_kernel, dim3(cris_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, scale, drop_thresh,
                       seed, stream_id, seed_dev);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void step_advance_kernel(int* step, uint32_t* seed, int* exchange_gen) {
    const int s = step[0];                       // steps completed so far
    seed[0] = (uint32_t)s * 7919u + 17u;        // dropout seed of the step that starts now (0-based rule)
    step[0] = s + 1;                             // 1-based count read by cris_adam_step
    // generation of this step's peer-mailbox exchanges (p2p_ll.h): a counter of its own that NOTHING ever rewinds - the optimizer
    // step can be (load_optimizer_state_dict of an earlier checkpoint on a live trainer), and a rewound generation would accept
    // the words still lying in the mailboxes from the first time it was used
    if (exchange_gen) exchange_gen[0] += 1;
}
extern "C" int cris_step_advance(int32_t* step, uint32_t* seed, int32_t* exchange_gen, void* stream) {
    CRIS_CHECK_ARG(step && seed, "bad args");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, seed, exchange_gen);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void axpy_f32_kernel(float* dst, const float* src, float alpha, long n) {
    GRID_STRIDE(i, n) dst[i] += alpha * src[i];
}
extern "C" int cris_axpy_f32(float* dst, const float* src, float alpha, long n, void* stream) {
    CRIS_CHECK_ARG(dst && src && n > 0, "bad args");
    hipLaunchKernelGGL(axpy_f32_kernel, dim3(cris_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, alpha, n);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// ------------------------------------------------------------------------------------------------
// embedding
// ------------------------------------------------------------------------------------------------
__global__ void embed_fwd_kernel(const int64_t* tokens, const float* table, const float* pos, int Bn, int L, int D, float* out) {
    const int DV = D >> 2;
    const long total = (long)Bn * L * DV;
    GRID_STRIDE(idx, total) {
        const int dv = (int)(idx % DV);
        const long r = idx / DV;
        const int l = (int)(r % L);
        const int64_t tok = tokens[r];
        const float4 e = *reinterpret_cast<const float4*>(table + (size_t)tok * D + dv * 4);
        const float4 q = *reinterpret_cast<const float4*>(pos + (size_t)l * D + dv * 4);
        *reinterpret_cast<float4*>(out + (size_t)r * D + dv * 4) = make_float4(e.x + q.x, e.y + q.y, e.z + q.z, e.w + q.w);
    }
}
extern "C" int cris_embed_fwd(const int64_t* tokens, const float* table, const float* pos, int Bn, int L, int D, float* out,
                              void* stream) {
    CRIS_CHECK_ARG(tokens && table && pos && out && !(D & 3), "bad args");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(cris_grid_1d((long)Bn * L * (D / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       tokens, table, pos, Bn, L, D, out);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// Deterministic (no atomics): the thread of row r, column d adds up every row that holds the same token (same position) in
// row order and only the FIRST such row stores the sum - B*L is 136 / 176 rows, the scan is free.
__global__ void embed_bwd_kernel(const int64_t* tokens, const float* dx, int Bn, int L, int D, float* dtable, float* dpos,
                                 unsigned char* row_live) {
    const long total = (long)Bn * L * D;
    const int R = Bn * L;
    GRID_STRIDE(idx, total) {
        const int d = (int)(idx % D);
        const int r = (int)(idx / D);
        const int64_t tok = tokens[r];
        bool first = true;
        for (int q = 0; q < r; ++q) first = first && tokens[q] != tok;
        if (first) {
            float a = 0.f;
            for (int q = r; q < R; ++q)
                if (tokens[q] == tok) a += dx[(size_t)q * D + d];
            dtable[(size_t)tok * D + d] = a;
            if (row_live && d == 0) row_live[tok] = 1;
        }
        if (r < L) {                                   // position r: rows r, r + L, r + 2L, ...
            float a = 0.f;
            for (int b = 0; b < Bn; ++b) a += dx[((size_t)b * L + r) * D + d];
            dpos[(size_t)r * D + d] = a;
        }
    }
}
extern "C" int cris_embed_bwd(const int64_t* tokens, const float* dx, int Bn, int L, int D, float* dtable, float* dpos,
                              unsigned char* row_live, void* stream) {
    CRIS_CHECK_ARG(tokens && dx && dtable && dpos, "bad args");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(cris_grid_1d((long)Bn * L * D, 256)), dim3(256), 0, (hipStream_t)stream, tokens, dx,
                       Bn, L, D, dtable, dpos, row_live);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// EOT select: argmax over token ids (first max wins), gather that row
__global__ void eot_gather_kernel(const int64_t* tokens, const bf16_t* x, int L, int D, bf16_t* out, int* eot_index) {
    __shared__ int s_idx;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int best = 0;
        int64_t bv = tokens[(size_t)b * L];
        for (int l = 1; l < L; ++l) {
            const int64_t v = tokens[(size_t)b * L + l];
            if (v > bv) { bv = v; best = l; }
        }
        s_idx = best;
        eot_index[b] = best;
    }
    __syncthreads();
    const bf16_t* src = x + ((size_t)b * L + s_idx) * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[(size_t)b * D + d] = src[d];
}
extern "C" int cris_eot_gather(const int64_t* tokens, const cris_bf16* x, int Bn, int L, int D, cris_bf16* out, int* eot_index,
                               void* stream) {
    CRIS_CHECK_ARG(tokens && x && out && eot_index, "bad args");
    hipLaunchKernelGGL(eot_gather_kernel, dim3(Bn), dim3(256), 0, (hipStream_t)stream, tokens, x, L, D, out, eot_index);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void eot_scatter_add_kernel(const int* eot_index, const bf16_t* g, int L, int D, bf16_t* dx) {
    const int b = blockIdx.x;
    bf16_t* dst = dx + ((size_t)b * L + eot_index[b]) * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dst[d] = f2bf(bf2f(dst[d]) + bf2f(g[(size_t)b * D + d]));
}
extern "C" int cris_eot_scatter_add(const int* eot_index, const cris_bf16* dstate_rows, int Bn, int L, int D, cris_bf16* dx,
                                    void* stream) {
    CRIS_CHECK_ARG(eot_index && dstate_rows && dx, "bad args");
    hipLaunchKernelGGL(eot_scatter_add_kernel, dim3(Bn), dim3(256), 0, (hipStream_t)stream, eot_index, dstate_rows, L, D, dx);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// attnpool positional embedding resize (constant linear map R [T][G*G]) and helpers
// ------------------------------------------------------------------------------------------------
__global__ void posresize_fwd_kernel(const float* R, const float* pos, int T, int GG, int C, float* posr) {
    const long total = (long)T * C;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C);
        const int t = (int)(idx / C);
        float s = 0.f;
#pragma unroll 7
        for (int j = 0; j < GG; ++j) s += R[t * GG + j] * pos[(size_t)(1 + j) * C + c];      // (loads independent of the sum: unrolled; 22 us -> see r05 trace)
        posr[idx] = s;
    }
}
extern "C" int cris_posresize_fwd(const float* R, const float* pos, int T, int G, int C, float* posr, void* stream) {
    CRIS_CHECK_ARG(R && pos && posr, "bad args");
    hipLaunchKernelGGL(posresize_fwd_kernel, dim3(cris_grid_1d((long)T * C, 256)), dim3(256), 0, (hipStream_t)stream, R, pos, T,
                       G * G, C, posr);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void posresize_bwd_kernel(const float* R, const float* dposr, int T, int GG, int C, float* dpos) {
    const long total = (long)GG * C;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C);
        const int j = (int)(idx / C);
        float s = 0.f;
#pragma unroll 13
        for (int t = 0; t < T; ++t) s += R[t * GG + j] * dposr[(size_t)t * C + c];      // (loads independent of the sum: unrolled)
        dpos[(size_t)(1 + j) * C + c] += s;
    }
}
extern "C" int cris_posresize_bwd(const float* R, const float* dposr, int T, int G, int C, float* dpos, void* stream) {
    CRIS_CHECK_ARG(R && dposr && dpos, "bad args");
    hipLaunchKernelGGL(posresize_bwd_kernel, dim3(cris_grid_1d((long)G * G * C, 256)), dim3(256), 0, (hipStream_t)stream, R,
                       dposr, T, G * G, C, dpos);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void batch_rowsum_kernel(const bf16_t* dx, int ldx, int Bn, int T, int C, float* out) {
    const long total = (long)T * C;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C);
        const int t = (int)(idx / C);
        float s = 0.f;
        for (int b = 0; b < Bn; ++b) s += bf2f(dx[((size_t)b * T + t) * ldx + c]);
        out[idx] = s;
    }
}
extern "C" int cris_batch_rowsum(const cris_bf16* dx, int ldx, int Bn, int T, int C, float* out, void* stream) {
    CRIS_CHECK_ARG(dx && out, "bad args");
    hipLaunchKernelGGL(batch_rowsum_kernel, dim3(cris_grid_1d((long)T * C, 256)), dim3(256), 0, (hipStream_t)stream, dx, ldx, Bn,
                       T, C, out);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// text-to-pixel dynamic 3x3 conv (groups = batch): pred[b,oh,ow] = sum_{c,kh,kw} x[b,oh+kh-1,ow+kw-1,c] w[b][c*9+kh*3+kw] + bias[b]
// one lane = 8 channels; LP = C/8 lanes cooperate on a pixel (LP power of two <= 64)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dynconv_fwd_kernel(const bf16_t* __restrict__ x, int H, int W, int C, const float* __restrict__ wb,
                                                          int ldwb, float* __restrict__ pred, int pix_per_block) {
    const int b = blockIdx.y;
    const int LP = C >> 3;
    const int cl = threadIdx.x % LP, sub = threadIdx.x / LP, nsub = 256 / LP;
    float w[9][8];
    const float* wp = wb + (size_t)b * ldwb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[t][j] = wp[(cl * 8 + j) * 9 + t];
    const float bias = wp[C * 9];
    const int HW = H * W;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    for (int pb = p0; pb < p1; pb += nsub) {
        const int pix = pb + sub;
        float acc = 0.f;
        if (pix < p1) {
            const int oh = pix / W, ow = pix - oh * W;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    // branch-free taps: clamped address + 0/1 weight, so the 9 loads of a pixel are issued together
                    const int ih = oh + kh - 1, iw = ow + kw - 1;
                    const float ok = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) ? 1.f : 0.f;
                    const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
                    float v[8];
                    ld8bf(x + (((size_t)b * H + ihc) * W + iwc) * C + cl * 8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += ok * v[j] * w[kh * 3 + kw][j];
                }
        }
        for (int o = LP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (cl == 0 && pix < p1) pred[(size_t)b * HW + pix] = acc + bias;
    }
}

extern "C" int cris_dynconv_fwd(const cris_bf16* x, int Bn, int H, int W, int C, const float* wb, int ldwb, float* pred,
                                void* stream) {
    const int LP = C / 8;
    CRIS_CHECK_ARG(x && wb && pred && !(C & 7) && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, "C/8 must be a power of two <= 64");
    // pixels per block: every block first gathers its 72 weights per lane (strided reads), so few pixels per block is mostly prologue
    static const int ppb_env = cris_env_int("CRIS_DYNCONV_PPB", 128);     // (call r05e: 64 -> 128: -0.03 ms per step; 256: level)
    const int ppb = ppb_env;
    dim3 grid(cris_cdiv(H * W, ppb), Bn);
    hipLaunchKernelGGL(dynconv_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, H, W, C, wb, ldwb, pred, ppb);
    CRIS_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void dynconv_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ dpred, int H, int W, int C,
                                                          const float* __restrict__ wb, int ldwb, bf16_t* __restrict__ dx,
                                                          float* __restrict__ dwb_part, int pix_per_block) {
    extern __shared__ float sdw[];              // [C*9 + 1]
    const int b = blockIdx.y;
    const int LP = C >> 3;
    const int cl = threadIdx.x % LP, sub = threadIdx.x / LP, nsub = 256 / LP;
    for (int i = threadIdx.x; i < C * 9 + 1; i += 256) sdw[i] = 0.f;
    __syncthreads();
    float w[9][8], dw[9][8];
    const float* wp = wb + (size_t)b * ldwb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w[t][j] = wp[(cl * 8 + j) * 9 + t];
            dw[t][j] = 0.f;
        }
    float dbias = 0.f;
    const int HW = H * W;
    const float* dp = dpred + (size_t)b * HW;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    for (int pix = p0 + sub; pix < p1; pix += nsub) {
        const int ph = pix / W, pw = pix - ph * W;
        // (1) dx at this pixel: sum over taps of dpred[ph-kh+1, pw-kw+1] * w[tap]
        // (2) dw[tap] += dpred[ph-kh+1, pw-kw+1] * x[ph, pw]      (the weight gradient taken at the INPUT pixel, see below)
        float gx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float g0 = dp[pix];
        if (cl == 0) dbias += g0;
        // Both sums use the SAME nine shifted values g = dpred[ph-kh+1, pw-kw+1] (0 outside the image): the weight gradient is
        // taken at the INPUT pixel - dw[tap] += x[ph,pw] * g - so x is read once per pixel instead of once per tap (round 5: the
        // nine clamped reads per pixel were 9 x 44 MB through the L1 / L2, 52 us at 104 x 104; the terms summed are the same).
        float xq[8];
        ld8bf(x + ((size_t)b * HW + pix) * C + cl * 8, xq);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                // branch-free taps (clamped address, 0/1 weight): all loads of a pixel in flight together
                const int oh = ph - kh + 1, ow = pw - kw + 1;
                const float okg = ((unsigned)oh < (unsigned)H && (unsigned)ow < (unsigned)W) ? 1.f : 0.f;
                const float g = okg * dp[min(max(oh, 0), H - 1) * W + min(max(ow, 0), W - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    gx[j] += g * w[kh * 3 + kw][j];
                    dw[kh * 3 + kw][j] += g * xq[j];
                }
            }
        st8bf(dx + ((size_t)b * HW + pix) * C + cl * 8, gx);
    }
    // block sums in a fixed order (no atomics): lanes of one wave that hold the same channels combine by shuffles, the four
    // waves then add into LDS one after the other; the block stores its row of the partials table
    for (int o = LP; o < 64; o <<= 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) dw[t][j] += __shfl_xor(dw[t][j], o, 64);
        dbias += __shfl_xor(dbias, o, 64);
    }
    const int lane = threadIdx.x & 63;
    for (int w = 0; w < 4; ++w) {
        if ((int)(threadIdx.x >> 6) == w && lane < LP) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) sdw[(cl * 8 + j) * 9 + t] += dw[t][j];
            if (cl == 0) sdw[C * 9] += dbias;
        }
        __syncthreads();
    }
    float* part = dwb_part + ((size_t)blockIdx.x * gridDim.y + b) * ldwb;
    for (int i = threadIdx.x; i < ldwb; i += 256) part[i] = i < C * 9 + 1 ? sdw[i] : 0.f;      // padding columns: zeros
}

#ifndef DYNCONV_BWD_PPB
#define DYNCONV_BWD_PPB 128                       // pixels per block: 85 x B blocks at 104 x 104 (was 512: 22 x B blocks, 89 us)
#endif
static int dynconv_bwd_blocks(int HW) { return cris_cdiv(HW, DYNCONV_BWD_PPB); }
extern "C" long cris_dynconv_bwd_ws_floats(int Bn, int H, int W, int ldwb) { return (long)dynconv_bwd_blocks(H * W) * Bn * ldwb; }
void cris_launch_sum_partials(const float* part, int nparts, int ncol, float* out, hipStream_t stream);      // norm.hip

extern "C" int cris_dynconv_bwd(const cris_bf16* x, const float* dpred, int Bn, int H, int W, int C, const float* wb, int ldwb,
                                cris_bf16* dx, float* dwb, float* ws, void* stream) {
    const int LP = C / 8;
    CRIS_CHECK_ARG(x && dpred && wb && dx && dwb && ws && !(C & 7) && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0 && ldwb >= C * 9 + 1,
                   "bad args");
    const int ppb = DYNCONV_BWD_PPB;
    dim3 grid(dynconv_bwd_blocks(H * W), Bn);
    hipLaunchKernelGGL(dynconv_bwd_kernel, grid, dim3(256), (size_t)(C * 9 + 1) * sizeof(float), (hipStream_t)stream, x, dpred, H, W,
                       C, wb, ldwb, dx, ws, ppb);
    CRIS_LAUNCH_CHECK();
    // dwb[b][i] += the blocks' partial rows in block order (deterministic); padding columns of the table are never read
    cris_launch_sum_partials(ws, grid.x, Bn * ldwb, dwb, (hipStream_t)stream);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// mask resize (nearest), BCE with logits (mean), train metric
// ------------------------------------------------------------------------------------------------
__global__ void mask_resize_kernel(const float* mask, int Bn, int IH, int IW, int OH, int OW, float* out) {
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    const long total = (long)Bn * OH * OW;
    GRID_STRIDE(idx, total) {
        const int ow = (int)(idx % OW);
        const int oh = (int)((idx / OW) % OH);
        const int b = (int)(idx / ((long)OW * OH));
        const int ih = min((int)floorf(oh * sh), IH - 1);
        const int iw = min((int)floorf(ow * sw), IW - 1);
        out[idx] = mask[((size_t)b * IH + ih) * IW + iw];
    }
}
extern "C" int cris_mask_resize_nearest(const float* mask, int Bn, int IH, int IW, int OH, int OW, float* out, void* stream) {
    CRIS_CHECK_ARG(mask && out, "bad args");
    hipLaunchKernelGGL(mask_resize_kernel, dim3(cris_grid_1d((long)Bn * OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, mask, Bn,
                       IH, IW, OH, OW, out);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// BCE partial sums: one partial per block (fixed order inside the block), then one wave adds the partials in block order -
// deterministic without atomics, and the element pass still runs on many CUs
#define BCE_BLOCKS 128
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* x, const float* t, long n, float* part) {
    __shared__ float sw[4];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)BCE_BLOCKS * 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        const float4 y = *reinterpret_cast<const float4*>(t + i * 4);
        s += fmaxf(v.x, 0.f) - v.x * y.x + log1pf(__expf(-fabsf(v.x)));
        s += fmaxf(v.y, 0.f) - v.y * y.y + log1pf(__expf(-fabsf(v.y)));
        s += fmaxf(v.z, 0.f) - v.z * y.z + log1pf(__expf(-fabsf(v.z)));
        s += fmaxf(v.w, 0.f) - v.w * y.w + log1pf(__expf(-fabsf(v.w)));
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float v = x[i];
            s += fmaxf(v, 0.f) - v * t[i] + log1pf(__expf(-fabsf(v)));
        }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(64) void bce_finish_kernel(const float* part, long n, float* loss) {
    float a = part[threadIdx.x] + part[threadIdx.x + 64];          // BCE_BLOCKS == 128
    a = wave_sum(a);                                               // fixed butterfly order
    if (threadIdx.x == 0) loss[0] = a / (float)n;
}
extern "C" int cris_bce_ws_floats(void) { return BCE_BLOCKS; }
extern "C" int cris_bce_fwd(const float* logits, const float* target, long n, float* loss, float* ws, void* stream) {
    CRIS_CHECK_ARG(logits && target && loss && ws && n > 0, "bad args");
    CRIS_CHECK_ARG((uintptr_t)logits % 16 == 0 && (uintptr_t)target % 16 == 0, "operands must be 16-byte aligned");
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(BCE_BLOCKS), dim3(256), 0, (hipStream_t)stream, logits, target, n, ws);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, n, loss);
    CRIS_LAUNCH_CHECK();
    return 0;
}
__global__ void bce_bwd_kernel(const float* x, const float* t, long n, const float* gscale, float* dx) {
    const float g = (gscale ? gscale[0] : 1.f) / (float)n;
    GRID_STRIDE(i, n) {
        const float v = x[i];
        dx[i] = (1.f / (1.f + __expf(-v)) - t[i]) * g;
    }
}
extern "C" int cris_bce_bwd(const float* logits, const float* target, long n, const float* gscale, float* dlogits, void* stream) {
    CRIS_CHECK_ARG(logits && target && dlogits && n > 0, "bad args");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(cris_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, target, n, gscale, dlogits);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ONE block of 1024 threads walks the samples in order (counts are exact integers, the mean over samples is summed in sample
// order): deterministic, and 8 x 10816 elements are launch-latency sized anyway
__global__ __launch_bounds__(1024) void train_metric_kernel(const float* x, const float* t, int Bn, int HW, float thr, float pr_iou, float* out) {
    __shared__ float si[16], su[16];
    float iou_sum = 0.f, pr_sum = 0.f;
    for (int b = 0; b < Bn; ++b) {
        float inter = 0.f, uni = 0.f;
        for (int i = threadIdx.x; i < HW; i += 1024) {
            const bool o = 1.f / (1.f + __expf(-x[(size_t)b * HW + i])) >= thr;
            const bool g = t[(size_t)b * HW + i] != 0.f;
            inter += (o && g) ? 1.f : 0.f;
            uni += (o || g) ? 1.f : 0.f;
        }
        inter = wave_sum(inter);
        uni = wave_sum(uni);
        __syncthreads();                                  // previous sample's table consumed
        if ((threadIdx.x & 63) == 0) { si[threadIdx.x >> 6] = inter; su[threadIdx.x >> 6] = uni; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, u = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { a += si[w]; u += su[w]; }
            const float iou = a / (u + 1e-6f);
            iou_sum += iou;
            pr_sum += iou > pr_iou ? 1.f : 0.f;
        }
    }
    if (threadIdx.x == 0) {
        out[0] = 100.f * iou_sum / (float)Bn;
        out[1] = 100.f * pr_sum / (float)Bn;
    }
}
extern "C" int cris_train_metric(const float* logits, const float* target, int Bn, int HW, float thr, float pr_iou, float* out,
                                 void* stream) {
    CRIS_CHECK_ARG(logits && target && out, "bad args");
    hipLaunchKernelGGL(train_metric_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, target, Bn, HW, thr, pr_iou, out);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fused multi-tensor Adam (torch.optim.Adam, non-amsgrad)
// ------------------------------------------------------------------------------------------------
#ifndef ADAM_ELEMS
#define ADAM_ELEMS 8192
#endif
#define AP_T 64                                   // packed tensors: a block owns AP_TN output rows x 64 input channels x all taps
#define AP_TN(PT) ((PT) == 9 ? 32 : 64)           // 9 taps: 32 rows (37 KB of LDS, four blocks per CU; 64 rows = 74 KB, two blocks, ran
                                                  // the table at 3.4 TB/s against 5.7 for the 1-tap one)
#define AP_LROW(PT) (AP_T * (PT) + 2)             // bf16 elements per LDS tile row (+1 dword: conflict-free column reads)

struct adam_coef {
    float beta1, beta2, eps, wd, bc1, rsb2, gscale;
};
// torch.optim.Adam (non-amsgrad): step_size = lr / bias_correction1, denom = sqrt(v) / sqrt(bias_correction2) + eps
__device__ __forceinline__ void adam_update(const adam_coef& k, float lr, float g, float& p, float& m, float& v) {
    g *= k.gscale;
    if (k.wd != 0.f) g += k.wd * p;
    m = k.beta1 * m + (1.f - k.beta1) * g;
    v = k.beta2 * v + (1.f - k.beta2) * g * g;
    p = p - (lr / k.bc1) * m / (sqrtf(v) * k.rsb2 + k.eps);
}

// One tile of a GEMM weight: Adam on the fp32 master values AND the bf16 operand copies the next step's kernels read -
// F [n][tap][Cpad] (forward) and D [c][taps-1-tap][Npad] (input gradient) - written from the freshly updated values through an
// LDS tile, so the fp32 weights are not read a second time by a separate packing pass (1.8 GB / step at CRIS-R50).
template <int PT>
__device__ __forceinline__ void adam_pack_tile(const cris_adam_desc& d, int lb, const adam_coef& k, bf16_t* tile) {
    constexpr int LROW = AP_LROW(PT);
    const int tiles_c = (d.cin + AP_T - 1) / AP_T;
    const int tn = lb / tiles_c, tc = lb - tn * tiles_c;
    constexpr int TN = AP_TN(PT);
    const int n0 = tn * TN, c0 = tc * AP_T;
    const int rows = min(TN, d.N - n0), cw = min(AP_T, d.cin - c0);
    if (!d.transposed) {
        // parameter layout [n][c][tap]: for one n the tile's (c, tap) range is contiguous.  (Eight elements per thread with all
        // 32 loads requested ahead of the first update - more bytes in flight for the two 74-KB-LDS blocks a CU holds - ran the
        // 9-tap table 25% slower, 0.41 against 0.33 ms: call r03u.)
        for (int i = threadIdx.x; i < rows * AP_T * PT; i += 256) {
            const int n_l = i / (AP_T * PT), e = i - n_l * (AP_T * PT);
            const int c_l = e / PT, tap = e - c_l * PT;
            if (c_l >= cw) continue;
            const long pi = ((long)(n0 + n_l) * d.cin + c0) * PT + e;
            const long gi = d.taps > 0 ? ((long)(n0 + n_l) * PT + tap) * d.cpad + c0 + c_l : pi;
            float p = d.p[pi], m = d.m[pi], v = d.v[pi];
            adam_update(k, d.lr, d.g[gi], p, m, v);
            d.p[pi] = p; d.m[pi] = m; d.v[pi] = v;
            tile[n_l * LROW + e] = f2bf(p);
        }
    } else {
        // parameter stored [c][n] (used as x @ P), one tap: contiguous along n
        for (int i = threadIdx.x; i < cw * AP_T; i += 256) {
            const int c_l = i / AP_T, n_l = i - c_l * AP_T;
            if (n_l >= rows) continue;
            const long pi = (long)(c0 + c_l) * d.N + n0 + n_l;
            float p = d.p[pi], m = d.m[pi], v = d.v[pi];
            adam_update(k, d.lr, d.g[pi], p, m, v);
            d.p[pi] = p; d.m[pi] = m; d.v[pi] = v;
            tile[n_l * LROW + c_l * PT] = f2bf(p);
        }
    }
    __syncthreads();
    if (d.dstF) {                                  // F[n*ldF + tap*Cpad + c]: consecutive threads = consecutive channels
        const int cpadF = d.cpad;
        const long ldF = d.ldF > 0 ? d.ldF : (long)PT * cpadF;        // (row stride: cris_pack_desc.ldF)
        for (int i = threadIdx.x; i < rows * PT * AP_T; i += 256) {
            const int c_l = i & (AP_T - 1), r = i >> 6;
            const int n_l = r / PT, tap = r - n_l * PT;
            if (c_l < cw) d.dstF[(long)(n0 + n_l) * ldF + (long)tap * cpadF + c0 + c_l] = tile[n_l * LROW + c_l * PT + tap];
        }
    }
    if (d.dstD) {                                  // D[c*ldD + (PT-1-tap)*Npad + n]: consecutive threads = consecutive rows n
        const long ldD = d.ldD > 0 ? d.ldD : (long)PT * d.npad;
        for (int i = threadIdx.x; i < cw * PT * TN; i += 256) {
            const int n_l = i & (TN - 1), r = i / TN;
            const int c_l = r / PT, tapf = r - c_l * PT;
            if (n_l < rows) d.dstD[(long)(c0 + c_l) * ldD + (long)tapf * d.npad + n0 + n_l] = tile[n_l * LROW + c_l * PT + (PT - 1 - tapf)];
        }
    }
}

template <int PT>
__global__ __launch_bounds__(256) void adam_kernel(const cris_adam_desc* __restrict__ tab, int n_desc, float beta1, float beta2, float eps,
                                                   float wd, float bc1, float bc2, float gscale, const int* __restrict__ step_dev,
                                                   const float* __restrict__ loss_scale_dev, const float* __restrict__ skip_dev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char adam_smem[];
    // torch.amp.GradScaler semantics on the device (cris_adam_step_amp): the whole update is skipped when the scaler found a
    // non-finite gradient, and the gradients still carry the loss scale - no host synchronisation, no separate unscale pass
    if (skip_dev && skip_dev[0] != 0.f) return;
    if (loss_scale_dev) gscale = gscale / loss_scale_dev[0];
    if (step_dev) {                       // step count lives on the device (HIP-graph replay): bias corrections from it,
        __shared__ float s_bc[2];         // in double precision like torch.optim.Adam's host arithmetic (1 - beta**t)
        if (threadIdx.x == 0) {
            const double t = (double)step_dev[0];
            s_bc[0] = (float)(1.0 - pow((double)beta1, t));
            s_bc[1] = (float)(1.0 - pow((double)beta2, t));
        }
        __syncthreads();
        bc1 = s_bc[0];
        bc2 = s_bc[1];
    }
    int lo = 0, hi = n_desc - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_start <= bid) lo = mid; else hi = mid - 1;
    }
    const cris_adam_desc d = tab[lo];
    adam_coef k;
    k.beta1 = beta1; k.beta2 = beta2; k.eps = eps; k.wd = wd; k.bc1 = bc1; k.rsb2 = rsqrtf(bc2); k.gscale = gscale;
    if (d.dstF || d.dstD) {               // block-uniform
        adam_pack_tile<PT>(d, bid - d.block_start, k, reinterpret_cast<bf16_t*>(adam_smem));
        return;
    }
    const long base = (long)(bid - d.block_start) * ADAM_ELEMS;
    const unsigned char* live = (wd == 0.f && d.taps == 0) ? d.row_live : nullptr;      // see cris_adam_desc.row_live
    for (int e = threadIdx.x; e < ADAM_ELEMS; e += 256) {
        const long i = base + e;
        if (i >= d.n) break;
        if (live && !live[i / d.row_len]) continue;       // a row without any gradient so far: the update is the identity
        long gi = i;
        if (d.taps > 0) {                 // gradient kept in the GEMM layout [n][tap][cpad] (cris_conv_wgrad)
            const long per_n = (long)d.cin * d.taps;
            const long n = i / per_n;
            const int r = (int)(i - n * per_n);
            const int c = r / d.taps, tap = r - c * d.taps;
            gi = (n * d.taps + tap) * d.cpad + c;
        }
        float p = d.p[i], m = d.m[i], v = d.v[i];
        adam_update(k, d.lr, d.g[gi], p, m, v);
        d.p[i] = p; d.m[i] = m; d.v[i] = v;
    }
}
// blocks a descriptor needs (host): tiles for tensors whose bf16 packs are refreshed by the update, else 8192-element chunks
extern "C" int cris_adam_blocks(const cris_adam_desc* d) {
    if (d->dstF || d->dstD) return cris_cdiv(d->N, d->taps == 9 ? AP_TN(9) : AP_TN(1)) * cris_cdiv(d->cin, AP_T);
    return cris_cdiv(d->n, ADAM_ELEMS);
}
extern "C" int cris_adam_block_elems(void) { return ADAM_ELEMS; }

// GEMM-layout gradient -> parameter-layout gradient for a table of tensors (same descriptor as Adam: p = destination in the
// parameter layout, g = source [n][tap][cpad]); used when a torch optimizer / DDP wants ordinary .grad tensors
__global__ __launch_bounds__(256) void unpack_grads_kernel(const cris_adam_desc* __restrict__ tab, int n_desc) {
    int lo = 0, hi = n_desc - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_start <= bid) lo = mid; else hi = mid - 1;
    }
    const cris_adam_desc d = tab[lo];
    const long base = (long)(bid - d.block_start) * ADAM_ELEMS;
    const long per_n = (long)d.cin * max(d.taps, 1);
    for (int e = threadIdx.x; e < ADAM_ELEMS; e += 256) {
        const long i = base + e;
        if (i >= d.n) break;
        long gi = i;
        if (d.taps > 0) {
            const long n = i / per_n;
            const int r = (int)(i - n * per_n);
            const int c = r / d.taps, tap = r - c * d.taps;
            gi = (n * d.taps + tap) * d.cpad + c;
        }
        d.p[i] = d.g[gi];
    }
}
extern "C" int cris_unpack_grads(const cris_adam_desc* dev_table, int n_desc, int total_blocks, void* stream) {
    CRIS_CHECK_ARG(dev_table && n_desc > 0 && total_blocks > 0, "empty table");
    hipLaunchKernelGGL(unpack_grads_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, dev_table, n_desc);
    CRIS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cris_adam_step_amp(const cris_adam_desc* dev_table, int n_desc, int total_blocks, float beta1, float beta2, float eps,
                                  float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, const int32_t* step_dev,
                                  const float* loss_scale_dev, const float* skip_dev, int pack_taps, void* stream) {
    CRIS_CHECK_ARG(dev_table && n_desc > 0 && total_blocks > 0, "empty table");
    CRIS_CHECK_ARG(pack_taps == 1 || pack_taps == 9, "a table holds tensors packed with 1 tap (and unpacked ones) or with 9 taps");
    typedef void (*adam_fn)(const cris_adam_desc*, int, float, float, float, float, float, float, float, const int*, const float*, const float*);
    const adam_fn k9 = adam_kernel<9>, k1 = adam_kernel<1>;
    constexpr int LDS9 = AP_TN(9) * AP_LROW(9) * 2, LDS1 = AP_TN(1) * AP_LROW(1) * 2;
    static const int ready = (int)hipFuncSetAttribute((const void*)k9, hipFuncAttributeMaxDynamicSharedMemorySize, LDS9);
    if (ready != 0) {
        cris_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (%d)", __func__, ready);
        return ready;
    }
    hipLaunchKernelGGL(pack_taps == 9 ? k9 : k1, dim3(total_blocks), dim3(256), pack_taps == 9 ? LDS9 : LDS1, (hipStream_t)stream, dev_table,
                       n_desc, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale, step_dev, loss_scale_dev, skip_dev);
    CRIS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cris_adam_step(const cris_adam_desc* dev_table, int n_desc, int total_blocks, float beta1, float beta2, float eps,
                              float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, const int32_t* step_dev,
                              int pack_taps, void* stream) {
    return cris_adam_step_amp(dev_table, n_desc, total_blocks, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale, step_dev,
                              nullptr, nullptr, pack_taps, stream);
}

// counter[0] += 1 unless skip[0] != 0: the optimizer's step count under a GradScaler that may skip the step (found_inf)
__global__ void counter_advance_unless_kernel(int32_t* counter, const float* skip) {
    if (!(skip && skip[0] != 0.f)) counter[0] += 1;
}
extern "C" int cris_counter_advance_unless(int32_t* counter, const float* skip, void* stream) {
    CRIS_CHECK_ARG(counter, "null counter");
    hipLaunchKernelGGL(counter_advance_unless_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, skip);
    CRIS_LAUNCH_CHECK();
    return 0;
}
