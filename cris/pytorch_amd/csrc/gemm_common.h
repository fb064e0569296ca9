// Pieces shared by the implicit-GEMM kernels (gemm.hip: 4-wave tiles, skinny kernel; gemm8.hip: 8-wave ping-pong tiles):
// the XOR-swizzled LDS row layout and the epilogue of one wave tile.
#pragma once
#include "common.h"
#include "../../../include/cris_hip.h"
#include <type_traits>

#define BK 64
#ifndef CRIS_FAST_EPILOGUE
#define CRIS_FAST_EPILOGUE 1      // 0: always the general epilogue (A/B builds)
#endif
#ifndef CRIS_FAST_EPILOGUE_GEN
#define CRIS_FAST_EPILOGUE_GEN 1  // 0: EPI 0 always takes the general form (A/B builds)
#endif

// XCD-aware block order: blocks b, b+8, b+16 .. run on the same XCD (round-robin dispatch); XCD x gets the CONTIGUOUS run of
// logical blocks [x*total/8, (x+1)*total/8) so that its private L2 keeps the operand panel the run shares
__device__ __forceinline__ int cris_xcd_logical_block(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ int lds_off(int row, int chunk) {          // bytes; rows are 128 B (64 bf16)
    return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4);
}

// hardware round-to-nearest-even conversion (v_cvt_pk_bf16_f32); equals f2bf() on every finite value
__device__ __forceinline__ bf16_t f2bf_hw(float x) { return __builtin_bit_cast(bf16_t, (__bf16)x); }

// Lean epilogue of an INTERIOR wave tile of 32x32 fragments (all FM*32 rows < M, all FN*32 columns < N; EPI 1 / 2): what every
// block of a convolution GEMM runs.  The general form below spends ~25 VALU instructions per output element - a v_mul_lo_u32 for
// the offset, bounds selects, bf16 rounding by hand, a branch around every store - so a 256x256 tile cost ~16 us of epilogue
// and the K <= 256 layers were bound by it (probe, call r03g: 23.6 us of a 34.5 us launch remain with the main loop AND the
// stores removed).  Here the byte offset of an element is (lane part, four VGPRs set up once) + (fragment part, an SGPR
// expression handed to the buffer instruction as its scalar offset): no address arithmetic per element, no bounds tests,
// v_cvt_pk_bf16_f32 for the rounding; the stored values are written back into the accumulator array for the second statistics
// pass.  Values, rounding and BatchNorm partials are those of the general form.
template <int EPI, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_fast32(const cris_conv_gemm_params& p, f32x16 (&acc)[FM][FN], int row0, int col0, int part,
                                                     int lane) {
    const int fr = lane & 31, fg = lane >> 5;
    const bool has_res = p.resid != nullptr;
    constexpr bool bnr = EPI == 3;          // lean + BatchNorm-backward partials (cris_hip.h: bnr_y): an instantiation of its own,
                                            // so that the plain lean kernels keep their register count (128x128: 244 -> two waves per SIMD)
    const int act = EPI == 2 ? p.act : 0;
    const int sld = p.stat_ld ? p.stat_ld : p.N;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.M * p.ldc * 2), CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.resid), 0,
                                                                        has_res ? (int)((size_t)p.M * p.ldr * 2) : 0, CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bnr_y), 0,
                                                                        bnr ? (int)((size_t)p.M * p.bnr_ldy * 2) : 0, CRIS_BUF_FLAGS);
    unsigned vo[4], vr[4], vy[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        vo[r] = ((unsigned)(row0 + fg * 4 + r) * (unsigned)p.ldc + (unsigned)(p.c_coff + col0 + fr)) * 2u;
        vr[r] = has_res ? ((unsigned)(row0 + fg * 4 + r) * (unsigned)p.ldr + (unsigned)(p.r_coff + col0 + fr)) * 2u : CRIS_OOB;
        vy[r] = bnr ? ((unsigned)(row0 + fg * 4 + r) * (unsigned)p.bnr_ldy + (unsigned)(p.bnr_coff + col0 + fr)) * 2u : CRIS_OOB;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const float bias = (EPI == 2 && p.bias) ? p.bias[col0 + j * 32 + fr] : 0.f;
        float s1 = 0.f;
        float b_mean = 0.f, b_inv = 0.f, b_sc = 0.f, b_sh = 0.f, b0 = 0.f, b1 = 0.f;
        if (bnr) {
            const int c = col0 + j * 32 + fr;
            b_mean = p.bnr_mean[c]; b_inv = p.bnr_invstd[c]; b_sc = p.bnr_scale[c]; b_sh = p.bnr_shift[c];
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float yv[16];
            if (bnr) {                              // wave-uniform: the pre-BatchNorm values of this fragment, 16 loads in flight
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int sy = ((i * 32 + (e >> 2) * 8) * p.bnr_ldy + j * 32) * 2;
                    yv[e] = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rsY, vy[e & 3], sy, 0));
                }
            }
            // the residual of a whole 32x32 fragment is requested before its first use: one memory round trip per fragment
            // instead of one per 4-row group (a branch around the loads would put a wait behind each group)
            float rres[16];
            if (has_res) {                          // wave-uniform: ONE branch per fragment, all 16 loads in flight together
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int sr = ((i * 32 + (e >> 2) * 8) * p.ldr + j * 32) * 2;
                    rres[e] = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rsR, vr[e & 3], sr, 0));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) rres[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int so = ((i * 32 + (e >> 2) * 8) * p.ldc + j * 32) * 2;          // wave-uniform: scalar offset of this 4-row group
                float x = acc[i][j][e];
                if constexpr (EPI == 2) {
                    x += bias;
                    if (act == 1) x = fmaxf(x, 0.f);
                }
                x += rres[e];                       // (+ 0 without a residual: as the general form does)
                if constexpr (EPI == 2) {
                    if (act == 3) x = fmaxf(x, 0.f);
                }
                acc[i][j][e] = x;
                s1 += x;
                const bf16_t xb = f2bf_hw(x);
                __builtin_amdgcn_raw_buffer_store_b16((short)xb, rsO, vo[e & 3], so, CRIS_STORE_AUX);
                if (bnr) {                          // as bn_bwd_reduce_fast_kernel (MASK 2) on the STORED gradient
                    const float g = (yv[e] * b_sc + b_sh) > 0.f ? bf2f(xb) : 0.f;
                    b0 += g;
                    b1 += g * ((yv[e] - b_mean) * b_inv);
                }
            }
        }
        if (bnr) {
            b0 += __shfl_xor(b0, 32, 64);
            b1 += __shfl_xor(b1, 32, 64);
            if (fg == 0) {
                p.colsum[(size_t)part * sld + col0 + j * 32 + fr] = b0;
                p.colsq[(size_t)part * sld + col0 + j * 32 + fr] = b1;
            }
        } else if (p.colsum) {
            s1 += __shfl_xor(s1, 32, 64);
            const float mu = s1 / (float)(FM * 32);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float d = acc[i][j][e] - mu;
                    q += d * d;
                }
            q += __shfl_xor(q, 32, 64);
            if (fg == 0) {
                p.colsum[(size_t)part * sld + col0 + j * 32 + fr] = s1;
                p.colsq[(size_t)part * sld + col0 + j * 32 + fr] = q;
            }
        }
    }
}

// The same treatment for the GENERAL epilogue (EPI 0) of an interior wave tile of 32x32 fragments: per-column bias, ReLU /
// QuickGELU, bf16 or fp32 residual and output, the head-split transposed copy, BatchNorm partials - the projections of the
// decoder / attention pool / text encoder (bias + transposed copy) and the fp32 residual-stream writers.  Dropout stays on the
// general form (three launches per step).  Same values as the general form: same operation order per element, f2bf() == the
// hardware conversion on every finite value.  The (batch, token) position of a row for the transposed copy is carried along
// the rows of the wave tile (one division per call instead of one per 4-row group); T_L >= 8 so that a step of 8 rows wraps
// at most once.
template <bool RES_F32, bool OUT_F32, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_fast32_gen(const cris_conv_gemm_params& p, f32x16 (&acc)[FM][FN], int row0, int col0, int part,
                                                         int lane) {
    const int fr = lane & 31, fg = lane >> 5;
    const bool has_res = p.resid != nullptr, has_out = p.out != nullptr, has_T = p.outT != nullptr;
    constexpr unsigned RES = RES_F32 ? 4u : 2u, OES = OUT_F32 ? 4u : 2u;
    const int act = p.act;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, has_out ? (int)((size_t)p.M * p.ldc * OES) : 0, CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.resid), 0,
                                                                        has_res ? (int)((size_t)p.M * p.ldr * RES) : 0, CRIS_BUF_FLAGS);
    unsigned vo[4], vr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        vo[r] = has_out ? ((unsigned)(row0 + fg * 4 + r) * (unsigned)p.ldc + (unsigned)(p.c_coff + col0 + fr)) * OES : CRIS_OOB;
        vr[r] = has_res ? ((unsigned)(row0 + fg * 4 + r) * (unsigned)p.ldr + (unsigned)(p.r_coff + col0 + fr)) * RES : CRIS_OOB;
    }
    // transposed copy: (batch, token) of this lane's first row
    int tb0 = 0, tl0 = 0;
    const bool t_pack = has_T && (p.T_L & 3) == 0;
    if (has_T) {
        const int m = row0 + fg * 4;
        tb0 = m / p.T_L;
        tl0 = m - tb0 * p.T_L;
    }
    const unsigned t_bstride = (unsigned)p.T_E * (unsigned)p.T_Lpad;           // elements between batches in one section
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = col0 + j * 32 + fr;
        const float bias = p.bias ? p.bias[col] : 0.f;
        bf16_t* tcol = nullptr;
        if (has_T) {
            const int sec = col / p.T_E;
            const int e_ = col - sec * p.T_E;
            tcol = p.outT + (size_t)sec * p.T_sec_stride + (size_t)e_ * p.T_Lpad;
        }
        int tb = tb0, tl = tl0;
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float rres[16];
            if (has_res) {                          // wave-uniform: one branch per fragment, all 16 loads in flight together
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int sr = ((i * 32 + (e >> 2) * 8) * p.ldr + j * 32) * (int)RES;
                    if constexpr (RES_F32) rres[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsR, vr[e & 3], sr, 0));
                    else rres[e] = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rsR, vr[e & 3], sr, 0));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) rres[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int so = ((i * 32 + (e >> 2) * 8) * p.ldc + j * 32) * (int)OES;
                float x = acc[i][j][e] + bias;
                if (act == 1) x = fmaxf(x, 0.f);
                else if (act == 2) x = x / (1.0f + __expf(-1.702f * x));
                x += rres[e];
                if (act == 3) x = fmaxf(x, 0.f);
                acc[i][j][e] = x;
                s1 += x;
                if constexpr (OUT_F32) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rsO, vo[e & 3], so, CRIS_STORE_AUX);
                else __builtin_amdgcn_raw_buffer_store_b16((short)f2bf_hw(x), rsO, vo[e & 3], so, CRIS_STORE_AUX);
            }
            if (has_T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // rows (i*32 + g*8 + fg*4) .. +3 of this column: (tb, tl) is the position of the first of them
                    if (t_pack) {                   // T_L % 4 == 0: the four rows are four consecutive tokens of one sample
                        uint2 w;
                        w.x = (uint32_t)f2bf_hw(acc[i][j][g * 4 + 0]) | ((uint32_t)f2bf_hw(acc[i][j][g * 4 + 1]) << 16);
                        w.y = (uint32_t)f2bf_hw(acc[i][j][g * 4 + 2]) | ((uint32_t)f2bf_hw(acc[i][j][g * 4 + 3]) << 16);
                        *reinterpret_cast<uint2*>(tcol + (size_t)((unsigned)tb * t_bstride + (unsigned)tl)) = w;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int lr = tl + r, br = tb;
                            if (lr >= p.T_L) { lr -= p.T_L; ++br; }
                            tcol[(size_t)((unsigned)br * t_bstride + (unsigned)lr)] = f2bf_hw(acc[i][j][g * 4 + r]);
                        }
                    }
                    tl += 8;                        // next 4-row group of this lane: 8 rows further
                    if (tl >= p.T_L) { tl -= p.T_L; ++tb; }
                }
            }
        }
        if (p.colsum) {
            s1 += __shfl_xor(s1, 32, 64);
            const float mu = s1 / (float)(FM * 32);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float d = acc[i][j][e] - mu;
                    q += d * d;
                }
            q += __shfl_xor(q, 32, 64);
            if (fg == 0) {
                const int sld = p.stat_ld ? p.stat_ld : p.N;
                p.colsum[(size_t)part * sld + col] = s1;
                p.colsq[(size_t)part * sld + col] = q;
            }
        }
    }
}

// Epilogue of one wave tile (FM x FN fragments of 16x16, C/D layout col = lane&15, row = (lane>>4)*4 + r) whose first
// row / column are row0 / col0: bias, activation, dropout, residual, bf16|fp32 store, head-split transposed copy and the
// BatchNorm statistics partial `part` (sum, M2 about the part mean over the FM*16 rows of this wave tile).
// EPI 1 ("lean"): compile-time promise of the plain conv / dgrad case (bf16 output, optional bf16 residual, optional
// statistics; no bias, activation, dropout, transposed copy or fp32 I/O) - the epilogue every block of the ~190
// convolution GEMMs per training step runs; the general form (EPI 0) costs thousands of instructions per wave.
// EPI 2: the lean case plus a per-column bias and ReLU before (act 1) or after (act 3) the residual - the convolutions of
// the inference path, whose BatchNorms are folded into weights and bias (cris/pytorch_amd/infer.py).
// EPI 3: the lean case of an input-gradient GEMM whose output is the gradient of relu(bn(y)): instead of the forward statistics
// the colsum / colsq tables receive the BatchNorm-backward partial sums (cris_hip.h: bnr_y).
template <int EPI, int MT, int FM, int FN, typename ACC>
__device__ __forceinline__ void gemm_epilogue(const cris_conv_gemm_params& p, ACC (&acc)[FM][FN], int row0, int col0, int part,
                                              int lane) {
    // MT = 16: v_mfma_f32_16x16x32 C/D layout  col = lane&15, row = (lane>>4)*4 + r            (r = 0..3)
    // MT = 32: v_mfma_f32_32x32x16 C/D layout  col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)   (reg = 0..15)
    // both: per lane NG groups of 4 consecutive rows of one column
    if constexpr (EPI != 0 && MT == 32 && std::is_same<ACC, f32x16>::value) {
        // wave-uniform: the whole wave tile lies inside the problem -> the cheap form (same values)
        if (p.out && row0 + FM * 32 <= p.M && col0 + FN * 32 <= p.N && CRIS_FAST_EPILOGUE) {
            gemm_epilogue_fast32<EPI, FM, FN>(p, acc, row0, col0, part, lane);
            return;
        }
    }
    if constexpr (EPI == 0 && MT == 32 && std::is_same<ACC, f32x16>::value) {
        // interior wave tile, no dropout: the cheap general form (bias / activation / fp32 streams / transposed copy)
        if (CRIS_FAST_EPILOGUE_GEN && row0 + FM * 32 <= p.M && col0 + FN * 32 <= p.N && p.drop_thresh == 0u && (!p.outT || p.T_L >= 8)) {
            const bool rf = p.resid && p.resid_f32, of = p.out && p.out_f32;
            if (!rf && !of) { gemm_epilogue_fast32_gen<false, false, FM, FN>(p, acc, row0, col0, part, lane); return; }
            if (rf && of) { gemm_epilogue_fast32_gen<true, true, FM, FN>(p, acc, row0, col0, part, lane); return; }
            if (!p.resid && of) { gemm_epilogue_fast32_gen<false, true, FM, FN>(p, acc, row0, col0, part, lane); return; }
        }
    }
    constexpr bool LEAN = EPI != 0;
    constexpr bool BIAS_ACT = EPI == 0 || EPI == 2;            // bias / activation compiled in
    constexpr int NG = MT == 16 ? 1 : 4;
    const int fr = lane & (MT - 1), fg = lane / MT;
    const bool has_drop = !LEAN && p.drop_thresh > 0u;
    const uint32_t dkey = LEAN ? 0u : cris_drop_key(p.drop_seed + (p.drop_seed_dev ? p.drop_seed_dev[0] : 0u), p.drop_stream);
    const uint32_t dthr = p.drop_thresh;
    const float dscale = has_drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const int Hh = (!LEAN && p.outT) ? p.T_E / 64 : 1;
    const int part_cnt = max(0, min(FM * MT, p.M - row0));
    // residual reads and output writes go through raw buffer descriptors: an element outside the problem (row >= M,
    // column >= N) is an out-of-range offset - reads return 0, writes are dropped - so the epilogue has no per-element
    // branches and its residual loads are issued together instead of one wait per element
    const bool has_res = p.resid != nullptr, has_out = p.out != nullptr;
    const bool res_f32 = !LEAN && p.resid_f32, out_f32 = !LEAN && p.out_f32;
    const int act = BIAS_ACT ? p.act : 0;
    const unsigned res_es = res_f32 ? 4u : 2u, out_es = out_f32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.resid), 0, has_res ? (int)((size_t)p.M * p.ldr * res_es) : 0, CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, has_out ? (int)((size_t)p.M * p.ldc * out_es) : 0,
                                                                        CRIS_BUF_FLAGS);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = col0 + j * MT + fr;
        const bool cvalid = col < p.N;
        const float bias = (BIAS_ACT && p.bias) ? p.bias[cvalid ? col : 0] : 0.f;
        float vals[FM * NG][4];
#pragma unroll
        for (int ig = 0; ig < FM * NG; ++ig) {
            const int i = ig / NG, g = ig % NG;
            const int rowb = row0 + i * MT + (MT == 16 ? fg * 4 : g * 8 + fg * 4);
            float v[4], rres[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = rowb + r;
                    const unsigned off = (cvalid && m < p.M) ? ((unsigned)m * (unsigned)p.ldr + (unsigned)(p.r_coff + col)) * res_es : CRIS_OOB;
                    if (res_f32) rres[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsR, off, 0, 0));
                    else rres[r] = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rsR, off, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = rowb + r;
                float x = acc[i][j][g * 4 + r] + bias;
                if (act == 1) x = fmaxf(x, 0.f);
                else if (!LEAN && act == 2) x = x / (1.0f + __expf(-1.702f * x));
                if (has_drop) x = cris_keep(dkey, (uint32_t)m * (uint32_t)p.N + (uint32_t)col, dthr) ? x * dscale : 0.f;
                const bool valid = cvalid && m < p.M;
                x += rres[r];
                if (act == 3) x = fmaxf(x, 0.f);
                if (!valid) x = 0.f;
                v[r] = x;
                vals[ig][r] = x;
                if (has_out) {
                    const unsigned off = valid ? ((unsigned)m * (unsigned)p.ldc + (unsigned)(p.c_coff + col)) * out_es : CRIS_OOB;
                    if (out_f32) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rsO, off, 0, CRIS_STORE_AUX);
                    else __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(x), rsO, off, 0, CRIS_STORE_AUX);
                }
            }
            if (!LEAN && p.outT && cvalid && rowb < p.M) {
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
        if constexpr (EPI == 3) {
            // BatchNorm-backward partials of an edge tile (see gemm_epilogue_fast32): rows / columns outside the problem hold 0
            const int sld = p.stat_ld ? p.stat_ld : p.N;
            const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.bnr_y), 0,
                                                                                (int)((size_t)p.M * p.bnr_ldy * 2), CRIS_BUF_FLAGS);
            const int cc = cvalid ? col : 0;
            const float b_mean = p.bnr_mean[cc], b_inv = p.bnr_invstd[cc], b_sc = p.bnr_scale[cc], b_sh = p.bnr_shift[cc];
            float b0 = 0.f, b1 = 0.f;
#pragma unroll
            for (int ig = 0; ig < FM * NG; ++ig) {
                const int i = ig / NG, g = ig % NG;
                const int rowb = row0 + i * MT + (MT == 16 ? fg * 4 : g * 8 + fg * 4);
                float yv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = rowb + r;
                    const unsigned off = (cvalid && m < p.M) ? ((unsigned)m * (unsigned)p.bnr_ldy + (unsigned)(p.bnr_coff + col)) * 2u : CRIS_OOB;
                    yv[r] = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rsY, off, 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xs = bf2f(f2bf(vals[ig][r]));                   // the stored gradient (0 outside the problem)
                    const float gg = (yv[r] * b_sc + b_sh) > 0.f ? xs : 0.f;
                    b0 += gg;
                    b1 += gg * ((yv[r] - b_mean) * b_inv);
                }
            }
            if (MT == 16) { b0 += __shfl_xor(b0, 16, 64); b1 += __shfl_xor(b1, 16, 64); }
            b0 += __shfl_xor(b0, 32, 64);
            b1 += __shfl_xor(b1, 32, 64);
            if (fg == 0 && cvalid && part_cnt > 0) {
                p.colsum[(size_t)part * sld + col] = b0;
                p.colsq[(size_t)part * sld + col] = b1;
            }
        } else if (p.colsum) {
            // BatchNorm statistics, robust + deterministic: per wave row-block (sum, M2 about the block mean);
            // cris_bn_finalize merges the blocks with Chan's formula.  No atomics, no E[x^2]-E[x]^2 cancellation.
            float s1 = 0.f;
#pragma unroll
            for (int ig = 0; ig < FM * NG; ++ig)
#pragma unroll
                for (int r = 0; r < 4; ++r) s1 += vals[ig][r];                // invalid rows hold 0
            if (MT == 16) s1 += __shfl_xor(s1, 16, 64);                       // lanes sharing this column
            s1 += __shfl_xor(s1, 32, 64);
            const float mu = part_cnt > 0 ? s1 / (float)part_cnt : 0.f;
            float q = 0.f;
#pragma unroll
            for (int ig = 0; ig < FM * NG; ++ig)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ig / NG, g = ig % NG;
                    const int m = row0 + i * MT + (MT == 16 ? fg * 4 : g * 8 + fg * 4) + r;
                    const float d = vals[ig][r] - mu;
                    q += (m < p.M) ? d * d : 0.f;
                }
            if (MT == 16) q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (fg == 0 && cvalid && part_cnt > 0) {          // parts = ceil(M / rows-per-part): none beyond the last row
                const int sld = p.stat_ld ? p.stat_ld : p.N;
                p.colsum[(size_t)part * sld + col] = s1;
                p.colsq[(size_t)part * sld + col] = q;
            }
        }
    }
}

