// Implicit-GEMM convolution / linear (forward and input-gradient) and weight-gradient kernels for gfx950.
//
// Forward: out[M,N] = A_im2col[M,K] * Wt[N,K]^T, bf16 operands, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//   tile 128 x BN x 64 (BN = 128: 2x2 waves of 64x64; BN = 64: 4x1 waves of 32x64), 256 threads,
//   register-staged global->LDS double buffer (the im2col gather + zero fill needs per-lane addresses,
//   so LDS-DMA is not used), LDS rows of 128 B with the 16-byte chunk index XOR-swizzled by (row>>1)&7 so
//   that ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct 16-B slots (conflict free,
//   see MI355X_MICROARCH.md section LDS), XCD-aware tile order (consecutive tiles of one XCD share the A panel).
// Wgrad: dW[N,K] = dY[M,N]^T * X_im2col[M,K]; the reduction dim (pixels) is the strided one for both
//   operands, so each thread loads 8 rows x 16 B, transposes the 8x8 bf16 block in registers and writes
//   m-contiguous 16-B chunks to LDS ([n][128 m] / [k][128 m], chunk XOR (row&15)); split over m with fp32
//   atomics straight into the parameter-layout gradient.
#include "common.h"
#include "../../../include/cris_hip.h"

#define BM 128
#define BK 64

// ------------------------------------------------------------------------------------------------
// forward / dgrad
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_off(int row, int chunk) {          // bytes; rows are 128 B (64 bf16)
    return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4);
}

template <int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const cris_conv_gemm_params p) {
    constexpr int WTM = BM / WAVES_M;          // wave tile rows
    constexpr int WTN = BN / WAVES_N;
    constexpr int FM = WTM / 16;
    constexpr int FN = WTN / 16;
    constexpr int NA = BM * 8 / 256;           // 16-B vectors per thread per K step (A)
    constexpr int NB = BN * 8 / 256;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    // XCD-aware tile order: blocks b, b+8, b+16.. run on the same XCD (observed round-robin); give each
    // XCD a contiguous run of tiles, n fastest, so its L2 keeps one A panel and sweeps the weights.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n;
    const int tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread A row decomposition (constant over the K loop) ----
    const int kc = t & 7;                      // 16-B chunk (8 channels) inside the 64-wide K step
    const int lrow = t >> 3;                   // 0..31
    const int OHW = p.OH * p.OW;
    int a_pix[NA], a_ih[NA], a_iw[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int b = m / OHW;
            const int r = m - b * OHW;
            const int oh = r / p.OW;
            const int ow = r - oh * p.OW;
            a_pix[i] = b * p.H * p.W;
            a_ih[i] = oh * p.stride - p.pad;
            a_iw[i] = ow * p.stride - p.pad;
        } else {
            a_pix[i] = 0; a_ih[i] = -(1 << 28); a_iw[i] = 0;       // never in range -> zeros
        }
    }
    // running (tap, c) of this thread's chunk
    int kcur = kc * 8;
    int c_cur = kcur % p.C;
    int tap = kcur / p.C;
    int kh = tap / p.KW;
    int kw = tap - kh * p.KW;

    uint4 ra[NA], rb[NB];
    auto load_tile = [&]() {
        const bool kvalid = kcur < p.K;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int ih = a_ih[i] + kh, iw = a_iw[i] + kw;
            if (kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
                const size_t off = (size_t)(a_pix[i] + ih * p.W + iw) * p.lda + p.a_coff + c_cur;
                v = *reinterpret_cast<const uint4*>(p.A + off);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int n = n0 + lrow + 32 * i;
            if (kvalid && n < p.N) v = *reinterpret_cast<const uint4*>(p.Wt + (size_t)n * p.ldb + kcur);
            rb[i] = v;
        }
        // advance to the next K step
        kcur += BK;
        c_cur += BK;
        while (c_cur >= p.C) {
            c_cur -= p.C;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sa = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(sa + lds_off(lrow + 32 * i, kc)) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<uint4*>(sb + lds_off(lrow + 32 * i, kc)) = rb[i];
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_tile();
    store_tile(0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile();
        const unsigned char* sa = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * WTM + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(row, ks * 4 + fg));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WTN + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(row, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + r ----
    const bool has_drop = p.drop_thresh > 0u;
    const uint32_t dkey = cris_drop_key(p.drop_seed, p.drop_stream);
    const uint32_t dthr = p.drop_thresh;
    const float dscale = has_drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const int Hh = p.outT ? p.T_E / 64 : 1;
    const int part = tile_m * WAVES_M + wm;                  // statistics partial index (one per wave row-block)
    const int part_row0 = m0 + wm * WTM;
    const int part_cnt = max(0, min(WTM, p.M - part_row0));
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * WTN + j * 16 + fr;
        const bool cvalid = col < p.N;
        const float bias = (p.bias && cvalid) ? p.bias[col] : 0.f;
        float vals[FM][4];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int rowb = m0 + wm * WTM + i * 16 + fg * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = rowb + r;
                float x = acc[i][j][r] + bias;
                if (p.act == 1) x = fmaxf(x, 0.f);
                else if (p.act == 2) x = x / (1.0f + __expf(-1.702f * x));
                if (has_drop) x = cris_keep(dkey, (uint32_t)m * (uint32_t)p.N + (uint32_t)col, dthr) ? x * dscale : 0.f;
                const bool valid = cvalid && m < p.M;
                if (valid && p.resid) {
                    const size_t ro = (size_t)m * p.ldr + p.r_coff + col;
                    x += p.resid_f32 ? reinterpret_cast<const float*>(p.resid)[ro]
                                     : bf2f(reinterpret_cast<const bf16_t*>(p.resid)[ro]);
                }
                if (!valid) x = 0.f;
                v[r] = x;
                vals[i][r] = x;
                if (valid && p.out) {
                    const size_t oo = (size_t)m * p.ldc + p.c_coff + col;
                    if (p.out_f32) reinterpret_cast<float*>(p.out)[oo] = x;
                    else reinterpret_cast<bf16_t*>(p.out)[oo] = f2bf(x);
                }
            }
            if (p.outT && cvalid && rowb < p.M) {
                const int sec = col / p.T_E;
                const int e = col - sec * p.T_E;
                const int h = e >> 6, d = e & 63;
                bf16_t* base = p.outT + (size_t)sec * p.T_sec_stride;
                if ((p.T_L & 3) == 0 && rowb + 3 < p.M) {
                    const int b = rowb / p.T_L, l = rowb - b * p.T_L;
                    uint2 w;
                    w.x = pack2bf(v[0], v[1]);
                    w.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(base + ((size_t)(b * Hh + h) * 64 + d) * p.T_Lpad + l) = w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = rowb + r;
                        if (m < p.M) {
                            const int b = m / p.T_L, l = m - b * p.T_L;
                            base[((size_t)(b * Hh + h) * 64 + d) * p.T_Lpad + l] = f2bf(v[r]);
                        }
                    }
                }
            }
        }
        if (p.colsum) {
            // BatchNorm statistics, robust + deterministic: per wave row-block (sum, M2 about the block mean);
            // cris_bn_finalize merges the blocks with Chan's formula.  No atomics, no E[x^2]-E[x]^2 cancellation.
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) s1 += vals[i][r];                 // invalid rows hold 0
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            const float mu = part_cnt > 0 ? s1 / (float)part_cnt : 0.f;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WTM + i * 16 + fg * 4 + r;
                    const float d = vals[i][r] - mu;
                    q += (m < p.M) ? d * d : 0.f;
                }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (fg == 0 && cvalid) {
                p.colsum[(size_t)part * p.N + col] = s1;
                p.colsq[(size_t)part * p.N + col] = q;
            }
        }
    }
}

extern "C" int cris_conv_gemm_stat_rows(int N) { return N <= 64 ? BM / 4 : BM / 2; }

extern "C" int cris_conv_gemm(const cris_conv_gemm_params* pp, void* stream) {
    const cris_conv_gemm_params& p = *pp;
    CRIS_CHECK_ARG(p.A && p.Wt, "null operand");
    CRIS_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "empty problem");
    CRIS_CHECK_ARG((p.C & 7) == 0 && (p.lda & 7) == 0 && (p.a_coff & 7) == 0, "A channels/ld/offset must be multiples of 8");
    CRIS_CHECK_ARG((p.ldb & 7) == 0 && (p.K & 7) == 0 && p.ldb >= p.K, "W ld / K must be multiples of 8");
    CRIS_CHECK_ARG(p.K == p.KH * p.KW * p.C, "K != KH*KW*C");
    CRIS_CHECK_ARG(p.M == p.Bn * p.OH * p.OW, "M != Bn*OH*OW");
    CRIS_CHECK_ARG(p.out || p.outT || p.colsum, "no output");
    CRIS_CHECK_ARG(!p.outT || ((p.T_E & 63) == 0 && p.T_L > 0 && (p.T_Lpad & 3) == 0 && p.T_Lpad >= p.T_L && p.M % p.T_L == 0),
                   "bad transposed-store geometry");
    CRIS_CHECK_ARG((uintptr_t)p.A % 16 == 0 && (uintptr_t)p.Wt % 16 == 0, "operands must be 16-byte aligned");
    CRIS_CHECK_ARG((long)p.M * p.N < (1L << 32) || p.drop_thresh == 0u, "dropout index overflow");
    const int tiles_m = cris_cdiv(p.M, BM);
    hipStream_t s = (hipStream_t)stream;
    if (p.N <= 64) {
        const int tiles_n = cris_cdiv(p.N, 64);
        hipLaunchKernelGGL((conv_gemm_kernel<64, 4, 1>), dim3(tiles_m * tiles_n), dim3(256), 0, s, p);
    } else {
        const int tiles_n = cris_cdiv(p.N, 128);
        hipLaunchKernelGGL((conv_gemm_kernel<128, 2, 2>), dim3(tiles_m * tiles_n), dim3(256), 0, s, p);
    }
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
#define WG_T 128          // output tile (n) x (k) and reduction step (m)

__device__ __forceinline__ int wg_off(int row, int chunk) {      // rows are 256 B (128 bf16 of m)
    return row * 256 + (((chunk ^ row) & 15) << 4);
}

// transpose an 8x8 block of bf16 held as 8 row vectors (uint4 = 8 bf16) into 8 column vectors
__device__ __forceinline__ void transpose8x8(const uint4* r, uint4* o) {
    const uint32_t* rw = reinterpret_cast<const uint32_t*>(r);       // rw[row*4 + word]
    uint32_t* ow = reinterpret_cast<uint32_t*>(o);                    // ow[col*4 + word]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int w = j >> 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t a = rw[(2 * q) * 4 + w], b = rw[(2 * q + 1) * 4 + w];
            ow[j * 4 + q] = (j & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        }
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const cris_wgrad_params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WG_T * 256];
    unsigned char* sy = smem;                  // dY^T tile [128 n][128 m]
    unsigned char* sx = smem + WG_T * 256;     // X^T  tile [128 k][128 m]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int k0 = blockIdx.x * WG_T;
    const int n0 = blockIdx.y * WG_T;
    int rows_per = (p.M + p.splits - 1) / p.splits;
    rows_per = (rows_per + WG_T - 1) / WG_T * WG_T;
    const int m_begin = blockIdx.z * rows_per;
    const int m_end = min(p.M, m_begin + rows_per);
    if (m_begin >= m_end) return;

    const int mg = (t & 7) + 8 * (t >> 7);     // 8-row group 0..15
    const int vec = (t >> 3) & 15;             // 16-B vector 0..15 along n (dY) / k (X)
    const int OHW = p.OH * p.OW;

    const int yn = n0 + vec * 8;
    const bool yvalid = yn < p.N_ld;
    const int xk = k0 + vec * 8;
    const bool xvalid = xk < p.K;
    const int xtap = xvalid ? xk / p.C : 0;
    const int xc = xvalid ? xk - xtap * p.C : 0;
    const int xkh = xtap / p.KW, xkw = xtap - xkh * p.KW;

    uint4 ry[8], rx[8];
    auto load_step = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = mb + mg * 8 + i;
            uint4 vy = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
            if (m < m_end) {
                if (yvalid) vy = *reinterpret_cast<const uint4*>(p.dY + (size_t)m * p.ldy + p.y_coff + yn);
                if (xvalid) {
                    const int b = m / OHW;
                    const int r = m - b * OHW;
                    const int oh = r / p.OW;
                    const int ow = r - oh * p.OW;
                    const int ih = oh * p.stride - p.pad + xkh, iw = ow * p.stride - p.pad + xkw;
                    if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                        vx = *reinterpret_cast<const uint4*>(p.X + (size_t)((b * p.H + ih) * p.W + iw) * p.ldx + p.x_coff + xc);
                }
            }
            ry[i] = vy;
            rx[i] = vx;
        }
    };
    auto store_step = [&]() {
        uint4 o[8];
        transpose8x8(ry, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4*>(sy + wg_off(vec * 8 + j, mg)) = o[j];
        transpose8x8(rx, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4*>(sx + wg_off(vec * 8 + j, mg)) = o[j];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    load_step(m_begin);
    for (int mb = m_begin; mb < m_end; mb += WG_T) {
        store_step();
        __syncthreads();
        if (mb + WG_T < m_end) load_step(mb + WG_T);       // in flight during the MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sy + wg_off(wr * 64 + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sx + wg_off(wc * 64 + j * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    const int taps = p.KH * p.KW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + wc * 64 + j * 16 + fr;
        if (k >= p.K) continue;
        const int tp = k / p.C;
        const int c = k - tp * p.C;
        if (c >= p.C_real) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * 64 + i * 16 + fg * 4 + r;
                if (n < p.N) atomicAdd(p.dW + ((size_t)n * p.C_real + c) * taps + tp, acc[i][j][r]);
            }
        }
    }
}

extern "C" int cris_conv_wgrad(const cris_wgrad_params* pp, void* stream) {
    const cris_wgrad_params& p = *pp;
    CRIS_CHECK_ARG(p.dY && p.X && p.dW, "null operand");
    CRIS_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.splits > 0, "empty problem");
    CRIS_CHECK_ARG((p.C & 7) == 0 && (p.ldx & 7) == 0 && (p.x_coff & 7) == 0, "X channels/ld/offset must be multiples of 8");
    CRIS_CHECK_ARG((p.ldy & 7) == 0 && (p.y_coff & 7) == 0 && (p.N_ld & 7) == 0 && p.N_ld >= p.N, "dY ld/offset/N_ld");
    CRIS_CHECK_ARG(p.K == p.KH * p.KW * p.C && p.M == p.Bn * p.OH * p.OW, "geometry");
    CRIS_CHECK_ARG(p.C_real > 0 && p.C_real <= p.C, "C_real");
    dim3 grid(cris_cdiv(p.K, WG_T), cris_cdiv(p.N, WG_T), p.splits);
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight packing (batched) and column sums
// ------------------------------------------------------------------------------------------------
#define PACK_ELEMS 4096     // destination elements handled per block

__global__ __launch_bounds__(256) void pack_weights_kernel(const cris_pack_desc* __restrict__ tab, int n_desc) {
    // binary search for the tensor this block belongs to
    int lo = 0, hi = n_desc - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_start <= bid) lo = mid; else hi = mid - 1;
    }
    const cris_pack_desc d = tab[lo];
    const int lb = bid - d.block_start;
    const long nF = d.dstF ? (long)d.N * d.taps * d.Cpad : 0;
    const long blocksF = (nF + PACK_ELEMS - 1) / PACK_ELEMS;
    if (lb < blocksF) {
        // F layout: dst[(n*taps + tap)*Cpad + c]; gather from src (reads of neighbouring taps hit L1/L2)
        const long base = (long)lb * PACK_ELEMS;
        for (int e = threadIdx.x; e < PACK_ELEMS; e += 256) {
            const long i = base + e;
            if (i >= nF) break;
            const int c = (int)(i % d.Cpad);
            const long r = i / d.Cpad;
            const int tap = (int)(r % d.taps);
            const int n = (int)(r / d.taps);
            float v = 0.f;
            if (c < d.Cin) v = d.src_transposed ? d.src[(size_t)c * d.N + n] : d.src[((size_t)n * d.Cin + c) * d.taps + tap];
            d.dstF[i] = f2bf(v);
        }
    } else {
        // D layout: dst[(c*taps + (taps-1-tap))*Npad + n]: 64(n) x 64(c) tile transpose through LDS per tap
        __shared__ float tile[64][65];
        const int tn = (d.Npad + 63) / 64, tc = (d.Cin + 63) / 64;
        int r = lb - (int)blocksF;
        const int nt = r % tn; r /= tn;
        const int ct = r % tc;
        const int tapf = r / tc;
        const int tap = d.taps - 1 - tapf;
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
        for (int rr = ty; rr < 64; rr += 4) {
            float v = 0.f;
            if (d.src_transposed) {                                    // src[c][n]: read along n
                const int c = ct * 64 + rr, n = nt * 64 + tx;
                if (c < d.Cin && n < d.N) v = d.src[(size_t)c * d.N + n];
                tile[tx][rr] = v;                                      // tile[n][c]
            } else {                                                   // src[n][c][tap]: read along c
                const int n = nt * 64 + rr, c = ct * 64 + tx;
                if (n < d.N && c < d.Cin) v = d.src[((size_t)n * d.Cin + c) * d.taps + tap];
                tile[rr][tx] = v;
            }
        }
        __syncthreads();
        for (int rr = ty; rr < 64; rr += 4) {
            const int c = ct * 64 + rr, n = nt * 64 + tx;
            if (c < d.Cin && n < d.Npad) d.dstD[((size_t)c * d.taps + tapf) * d.Npad + n] = f2bf(tile[tx][rr]);
        }
    }
}

extern "C" int cris_pack_blocks(const cris_pack_desc* d) {
    long nF = d->dstF ? (long)d->N * d->taps * d->Cpad : 0;
    long bF = (nF + PACK_ELEMS - 1) / PACK_ELEMS;
    long bD = d->dstD ? (long)d->taps * ((d->Cin + 63) / 64) * ((d->Npad + 63) / 64) : 0;
    return (int)(bF + bD);
}

extern "C" int cris_pack_block_elems(void) { return PACK_ELEMS; }

extern "C" int cris_pack_weights(const cris_pack_desc* dev_table, int n_desc, int total_blocks, void* stream) {
    CRIS_CHECK_ARG(dev_table && n_desc > 0 && total_blocks > 0, "empty table");
    hipLaunchKernelGGL(pack_weights_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, dev_table, n_desc);
    CRIS_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int ldx, int coff, int M, int N,
                                                     float* __restrict__ out, int rows_per_block) {
    // block (bx: column group of 256, by: row chunk); thread = one column, coalesced across the block
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int mb = blockIdx.y * rows_per_block;
    const int me = min(M, mb + rows_per_block);
    float s = 0.f;
    for (int m = mb; m < me; ++m) s += bf2f(x[(size_t)m * ldx + coff + n]);
    atomicAdd(out + n, s);
}

extern "C" int cris_colsum_bf16(const cris_bf16* x, int ldx, int coff, int M, int N, float* out, void* stream) {
    CRIS_CHECK_ARG(x && out && M > 0 && N > 0, "bad args");
    const int gx = cris_cdiv(N, 256);
    int gy = 2048 / gx;
    if (gy < 1) gy = 1;
    int rpb = cris_cdiv(M, gy);
    if (rpb < 8) rpb = 8;
    gy = cris_cdiv(M, rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, x, ldx, coff, M, N, out, rpb);
    CRIS_LAUNCH_CHECK();
    return 0;
}
