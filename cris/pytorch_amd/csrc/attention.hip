// Fused multi-head attention (head dim 64) forward / backward for gfx950.
//
// Sequence lengths on the CRIS path are short (text 17/22, attnpool 169/225, decoder 676/900 visual
// tokens x 17/22 words), so K/V of one (batch, head) are L2 resident and LDS staging would be pure
// overhead (cdna_hip_programming.md, common mistake 7).  Every MFMA operand is therefore loaded from
// global memory *directly in fragment order*:
//   S^T = K Q^T      A = K  [key][d]   (16 B per lane, token-major)         B = Q  [query][d]
//   O^T = V^T P^T    A = V^T[d][key]   (two 8-B loads from the head-split   B = P^T straight out of the
//                                       transposed copy the producing GEMM    S^T accumulator registers
//                                       wrote)
// Computing S transposed makes each lane own ONE query column (col = lane&15): the running max / sum and
// the O rescale are per-lane scalars, and the S^T accumulator (4 consecutive keys per register group) is
// already a valid B operand if the MFMA's reduction slots are mapped to keys as
//   slot (g, j) -> key  k0 + (j>>2)*16 + g*4 + (j&3)
// (any permutation of the reduction index is legal as long as A uses the same one).
// One wave = 16 queries (fwd, bwd_dq) or 16 keys (bwd_dkv); 4 waves per block.
#include "common.h"
#include "../../../include/cris_hip.h"

#define NEG_INF (-__builtin_inff())

// Fragment loads are UNCONDITIONAL: rows beyond the sequence are read from the last valid row (clamped index) and their
// scores / probabilities are masked afterwards.  A per-lane `if (ok) load` makes hipcc branch around every load and wait
// for it inside the branch, which serialises the loads of a step (these kernels sat at 75 % wave-wait cycles).
__device__ __forceinline__ bf16x8 ld_frag16(const bf16_t* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}
// two 8-byte pieces (reduction slots j=0..3 and j=4..7)
__device__ __forceinline__ bf16x8 ld_frag8x2(const bf16_t* p0, const bf16_t* p1) {
    const uint2 a = *reinterpret_cast<const uint2*>(p0);
    const uint2 b = *reinterpret_cast<const uint2*>(p1);
    return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ bf16x8 pack_frag(const float* v) {
    return __builtin_bit_cast(bf16x8, make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])));
}
__device__ __forceinline__ float grp_max(float v) {      // across the 4 lanes that share lane&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float grp_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const cris_attn_params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.Hn, h = bh - b * p.Hn;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= p.Lq) return;
    const int q = q0 + fr;
    const bool qok = q < p.Lq;

    const bf16_t* Qp = p.Q + (size_t)(b * p.Lq + min(q, p.Lq - 1)) * p.ldq + h * 64 + fg * 8;   // clamped: masked by !qok
    const bf16x8 bq0 = ld_frag16(Qp), bq1 = ld_frag16(Qp + 32);

    const bf16_t* Kb = p.K + (size_t)b * p.Lk * p.ldk + h * 64 + fg * 8;
    const bf16_t* Vt = p.Vt + (size_t)bh * 64 * p.Lk_pad;
    const int64_t* toks = p.key_tokens ? p.key_tokens + (size_t)b * p.Lk : nullptr;

    const bool has_drop = p.drop_thresh > 0u;
    const uint32_t dkey = cris_drop_key(p.drop_seed + (p.drop_seed_dev ? p.drop_seed_dev[0] : 0u), p.drop_stream);
    const float inv_keep = has_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t didx0 = ((uint32_t)bh * (uint32_t)p.Lq + (uint32_t)q) * (uint32_t)p.Lk;

    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = NEG_INF, l_run = 0.f;

    const int kend = p.causal ? min(p.Lk, q0 + 16) : p.Lk;
    for (int k0 = 0; k0 < kend; k0 += 32) {
        float s[8];
        // V^T fragments of this step: issued first, in flight under the QK^T MFMAs and the softmax
        bf16x8 av[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16_t* vp = Vt + (size_t)(db * 16 + fr) * p.Lk_pad + k0 + fg * 4;
            av[db] = ld_frag8x2(vp, vp + 16);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key = k0 + kb * 16 + fr;
            const bf16_t* kp = Kb + (size_t)min(key, p.Lk - 1) * p.ldk;
            const bf16x8 a0 = ld_frag16(kp), a1 = ld_frag16(kp + 32);
            f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f};
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bq0, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq1, st, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = k0 + kb * 16 + fg * 4 + r;
                float v = st[r] * p.scale;
                bool masked = kk >= p.Lk || (p.causal && kk > q);
                if (toks) masked = masked || toks[min(kk, p.Lk - 1)] == 0;
                s[kb * 4 + r] = masked ? NEG_INF : v;
            }
        }
        float mx = s[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, s[j]);
        mx = grp_max(mx);
        const float m_new = fmaxf(m_run, mx);
        float pv[8];
        float alpha = 1.f, rs = 0.f;
        if (m_new == NEG_INF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = 0.f;
        } else {
            alpha = __expf(m_run - m_new);                 // m_run = -inf -> 0
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pv[j] = __expf(s[j] - m_new);              // masked -> 0
                rs += pv[j];
            }
        }
        rs = grp_sum(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][r] *= alpha;
        if (has_drop) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = k0 + (j >> 2) * 16 + fg * 4 + (j & 3);
                pv[j] = cris_keep(dkey, didx0 + (uint32_t)kk, p.drop_thresh) ? pv[j] * inv_keep : 0.f;
            }
        }
        const bf16x8 bp = pack_frag(pv);
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[db], bp, o[db], 0, 0, 0);
    }
    const float inv_l = l_run > 0.f ? 1.f / l_run : 0.f;
    if (qok) {
        bf16_t* op = p.O + (size_t)(b * p.Lq + q) * p.ldo + h * 64 + fg * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 w;
            w.x = pack2bf(o[db][0] * inv_l, o[db][1] * inv_l);
            w.y = pack2bf(o[db][2] * inv_l, o[db][3] * inv_l);
            *reinterpret_cast<uint2*>(op + db * 16) = w;
        }
        if (fg == 0 && p.lse) p.lse[(size_t)bh * p.Lq + q] = l_run > 0.f ? m_run + __logf(l_run) : __builtin_inff();
    }
}

static int attn_check(const cris_attn_params& p) {
    CRIS_CHECK_ARG(p.Q && p.K && p.B > 0 && p.Hn > 0 && p.Lq > 0 && p.Lk > 0, "null operand");
    CRIS_CHECK_ARG((p.ldq & 7) == 0 && (p.ldk & 7) == 0, "ld must be a multiple of 8");
    CRIS_CHECK_ARG((p.Lk_pad & 3) == 0 && p.Lk_pad >= ((p.Lk + 31) / 32) * 32, "Lk_pad must cover whole 32-key tiles");
    CRIS_CHECK_ARG((long)p.B * p.Hn * p.Lq * p.Lk < (1L << 32) || p.drop_thresh == 0u, "dropout index overflow");
    return 0;
}

// long-sequence variants (defined at the end of this file): unmasked attention whose loop runs over >= attn_lds_min rows
__global__ void attn_fwd_lds_kernel(const cris_attn_params p);
__global__ void attn_bwd_dq_lds_kernel(const cris_attn_params p);
__global__ void attn_bwd_dkv_lds_kernel(const cris_attn_params p);
#ifndef AL_WAVES
#define AL_WAVES 4                         // waves per block
#endif
#ifndef AL_G
#define AL_G 1                             // 16-row groups per wave: a block covers 16 * AL_G * AL_WAVES = 64 rows
#endif
#define AL_LDS_FWD (3 * 2 * 8192)
#define AL_LDS_DQ (2 * 3 * 8192)
#define AL_LDS_DKV (2 * (4 * 8192 + 512))
static bool attn_use_lds(const cris_attn_params& p, int loop_rows, bool key_mask_ok = false) {
    static const int lds_min = cris_env_int("CRIS_ATTN_LDS_MIN", 384);
    return !p.causal && (key_mask_ok || !p.key_tokens) && loop_rows >= lds_min && (p.Lk_pad & 7) == 0 && ((uintptr_t)p.Q & 15) == 0 &&
           ((uintptr_t)p.K & 15) == 0 && ((uintptr_t)p.V & 15) == 0 && (size_t)p.B * p.Lk * p.ldk * 2 < (1UL << 31) &&
           (size_t)p.B * p.Lq * p.ldq * 2 < (1UL << 31) && (size_t)p.B * p.Hn * 64 * (p.Lk_pad > p.Lq_pad ? p.Lk_pad : p.Lq_pad) * 2 < (1UL << 31);
}
static int attn_launch_lds(int which, const cris_attn_params& p, dim3 grid, void* stream) {
    static const int ready = (int)hipFuncSetAttribute((const void*)attn_bwd_dkv_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AL_LDS_DKV);
    if (ready != 0) {
        cris_set_error("cris_attn: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (%d)", ready);
        return ready;
    }
    if (which == 0) hipLaunchKernelGGL(attn_fwd_lds_kernel, grid, dim3(64 * AL_WAVES), AL_LDS_FWD, (hipStream_t)stream, p);
    else if (which == 1) hipLaunchKernelGGL(attn_bwd_dq_lds_kernel, grid, dim3(64 * AL_WAVES), AL_LDS_DQ, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_bwd_dkv_lds_kernel, grid, dim3(64 * AL_WAVES), AL_LDS_DKV, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cris_attn_fwd(const cris_attn_params* pp, void* stream) {
    const cris_attn_params& p = *pp;
    if (attn_check(p)) return -1;
    CRIS_CHECK_ARG(p.Vt && p.O && (p.ldo & 3) == 0, "forward operands");
    dim3 grid(cris_cdiv(p.Lq, 64), p.B * p.Hn);
    if (attn_use_lds(p, p.Lk)) return attn_launch_lds(0, p, grid, stream);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward, query side: delta = rowsum(dO*O), dQ = scale * dS K
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const cris_attn_params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.Hn, h = bh - b * p.Hn;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= p.Lq) return;
    const int q = q0 + fr;
    const bool qok = q < p.Lq;

    const int qcl = min(q, p.Lq - 1);                      // rows beyond Lq read the last row; they are masked (!qok)
    const bf16_t* Qp = p.Q + (size_t)(b * p.Lq + qcl) * p.ldq + h * 64 + fg * 8;
    const bf16x8 bq0 = ld_frag16(Qp), bq1 = ld_frag16(Qp + 32);
    const bf16_t* dOp = p.dO + (size_t)(b * p.Lq + qcl) * p.lddo + h * 64 + fg * 8;
    const bf16x8 bd0 = ld_frag16(dOp), bd1 = ld_frag16(dOp + 32);
    // delta = sum_d dO*O for this query (this lane covers d = fg*8..+7 and 32+fg*8..+7)
    float delta = 0.f;
    {
        const bf16_t* Op = p.O + (size_t)(b * p.Lq + q) * p.ldo + h * 64 + fg * 8;
        uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
        if (qok) {
            o0 = *reinterpret_cast<const uint4*>(Op);
            o1 = *reinterpret_cast<const uint4*>(Op + 32);
        }
        float a[8], c[8];
        unpack8(__builtin_bit_cast(uint4, bd0), a);
        unpack8(o0, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += a[j] * c[j];
        unpack8(__builtin_bit_cast(uint4, bd1), a);
        unpack8(o1, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += a[j] * c[j];
        delta = grp_sum(delta);
        if (qok && fg == 0) p.delta[(size_t)bh * p.Lq + q] = delta;
    }
    const float lse = qok ? p.lse[(size_t)bh * p.Lq + q] : __builtin_inff();

    const bf16_t* Kb = p.K + (size_t)b * p.Lk * p.ldk + h * 64 + fg * 8;
    const bf16_t* Vb = p.V + (size_t)b * p.Lk * p.ldv + h * 64 + fg * 8;
    const bf16_t* Kt = p.Kt + (size_t)bh * 64 * p.Lk_pad;
    const int64_t* toks = p.key_tokens ? p.key_tokens + (size_t)b * p.Lk : nullptr;
    const bool has_drop = p.drop_thresh > 0u;
    const uint32_t dkey = cris_drop_key(p.drop_seed + (p.drop_seed_dev ? p.drop_seed_dev[0] : 0u), p.drop_stream);
    const float inv_keep = has_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    const uint32_t didx0 = ((uint32_t)bh * (uint32_t)p.Lq + (uint32_t)q) * (uint32_t)p.Lk;

    f32x4 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kend = p.causal ? min(p.Lk, q0 + 16) : p.Lk;
    for (int k0 = 0; k0 < kend; k0 += 32) {
        float ds[8];
        bf16x8 akt[4];                                   // K^T fragments of this step, in flight under the score MFMAs
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16_t* kp = Kt + (size_t)(db * 16 + fr) * p.Lk_pad + k0 + fg * 4;
            akt[db] = ld_frag8x2(kp, kp + 16);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int key = min(k0 + kb * 16 + fr, p.Lk - 1);
            const bf16_t* kp = Kb + (size_t)key * p.ldk;
            const bf16_t* vp = Vb + (size_t)key * p.ldv;
            const bf16x8 a0 = ld_frag16(kp), a1 = ld_frag16(kp + 32);
            const bf16x8 v0 = ld_frag16(vp), v1 = ld_frag16(vp + 32);
            f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = st;
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bq0, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq1, st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, bd0, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, bd1, dp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = k0 + kb * 16 + fg * 4 + r;
                bool masked = kk >= p.Lk || (p.causal && kk > q) || !qok;
                if (toks) masked = masked || toks[min(kk, p.Lk - 1)] == 0;
                float pr = masked ? 0.f : __expf(st[r] * p.scale - lse);
                float dpv = dp[r];
                if (has_drop) dpv = cris_keep(dkey, didx0 + (uint32_t)kk, p.drop_thresh) ? dpv * inv_keep : 0.f;
                ds[kb * 4 + r] = pr * (dpv - delta);
            }
        }
        const bf16x8 bds = pack_frag(ds);
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(akt[db], bds, dq[db], 0, 0, 0);
    }
    if (qok) {
        bf16_t* op = p.dQ + (size_t)(b * p.Lq + q) * p.lddq + h * 64 + fg * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 w;
            w.x = pack2bf(dq[db][0] * p.scale, dq[db][1] * p.scale);
            w.y = pack2bf(dq[db][2] * p.scale, dq[db][3] * p.scale);
            *reinterpret_cast<uint2*>(op + db * 16) = w;
        }
    }
}

extern "C" int cris_attn_bwd_dq(const cris_attn_params* pp, void* stream) {
    const cris_attn_params& p = *pp;
    if (attn_check(p)) return -1;
    CRIS_CHECK_ARG(p.V && p.Kt && p.O && p.dO && p.lse && p.delta && p.dQ, "backward(dq) operands");
    CRIS_CHECK_ARG((p.ldv & 7) == 0 && (p.lddo & 7) == 0 && (p.ldo & 7) == 0 && (p.lddq & 3) == 0, "ld");
    dim3 grid(cris_cdiv(p.Lq, 64), p.B * p.Hn);
    if (attn_use_lds(p, p.Lk)) return attn_launch_lds(1, p, grid, stream);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward, key side: dV = P_drop^T dO, dK = scale * dS^T Q     (one wave = 16 keys, loops over queries)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const cris_attn_params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.Hn, h = bh - b * p.Hn;
    const int kk0 = (blockIdx.x * 4 + wave) * 16;
    if (kk0 >= p.Lk) return;
    const int key = kk0 + fr;
    const bool kok = key < p.Lk;
    bool kmask = !kok;
    if (kok && p.key_tokens) kmask = p.key_tokens[(size_t)b * p.Lk + key] == 0;

    // B operands (col = key): K[key][d], V[key][d]
    const int keyc = min(key, p.Lk - 1);                   // keys beyond Lk read the last key; nothing is stored for them
    const bf16_t* Kp = p.K + (size_t)(b * p.Lk + keyc) * p.ldk + h * 64 + fg * 8;
    const bf16_t* Vp = p.V + (size_t)(b * p.Lk + keyc) * p.ldv + h * 64 + fg * 8;
    const bf16x8 bk0 = ld_frag16(Kp), bk1 = ld_frag16(Kp + 32);
    const bf16x8 bv0 = ld_frag16(Vp), bv1 = ld_frag16(Vp + 32);

    const bf16_t* Qb = p.Q + (size_t)b * p.Lq * p.ldq + h * 64 + fg * 8;
    const bf16_t* dOb = p.dO + (size_t)b * p.Lq * p.lddo + h * 64 + fg * 8;
    const bf16_t* Qt = p.Qt + (size_t)bh * 64 * p.Lq_pad;
    const bf16_t* dOt = p.dOt + (size_t)bh * 64 * p.Lq_pad;
    const float* lsep = p.lse + (size_t)bh * p.Lq;
    const float* delp = p.delta + (size_t)bh * p.Lq;
    const bool has_drop = p.drop_thresh > 0u;
    const uint32_t dkey = cris_drop_key(p.drop_seed + (p.drop_seed_dev ? p.drop_seed_dev[0] : 0u), p.drop_stream);
    const float inv_keep = has_drop ? 1.f / (1.f - p.drop_p) : 1.f;

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dk[i] = dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int qstart = p.causal ? (kk0 / 32) * 32 : 0;      // queries < key never attend under the causal mask
    for (int q0 = qstart; q0 < p.Lq; q0 += 32) {
        float pd[8], ds[8];
        bf16x8 ado[4], aqt[4];                           // dO^T / Q^T fragments of this step, issued first
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const size_t ro = (size_t)(db * 16 + fr) * p.Lq_pad + q0 + fg * 4;
            ado[db] = ld_frag8x2(dOt + ro, dOt + ro + 16);
            aqt[db] = ld_frag8x2(Qt + ro, Qt + ro + 16);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qrow = min(q0 + qb * 16 + fr, p.Lq - 1);   // A-operand row of this lane (clamped: masked below)
            const bf16_t* qp = Qb + (size_t)qrow * p.ldq;
            const bf16_t* dop = dOb + (size_t)qrow * p.lddo;
            const bf16x8 aq0 = ld_frag16(qp), aq1 = ld_frag16(qp + 32);
            const bf16x8 ad0 = ld_frag16(dop), ad1 = ld_frag16(dop + 32);
            f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = sv;
            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq0, bk0, sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq1, bk1, sv, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad0, bv0, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad1, bv1, dp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = q0 + qb * 16 + fg * 4 + r;     // C/D row of this lane = query
                const bool ok = qq < p.Lq && !kmask && !(p.causal && key > qq);
                const int qc = min(qq, p.Lq - 1);
                const float pr = ok ? __expf(sv[r] * p.scale - lsep[qc]) : 0.f;
                const float dlt = ok ? delp[qc] : 0.f;
                float dpv = dp[r];
                float prd = pr;
                if (has_drop) {
                    const bool keep = cris_keep(dkey, ((uint32_t)bh * (uint32_t)p.Lq + (uint32_t)qq) * (uint32_t)p.Lk + (uint32_t)key, p.drop_thresh);
                    prd = keep ? pr * inv_keep : 0.f;
                    dpv = keep ? dpv * inv_keep : 0.f;
                }
                pd[qb * 4 + r] = prd;
                ds[qb * 4 + r] = pr * (dpv - dlt);
            }
        }
        const bf16x8 bp = pack_frag(pd), bds = pack_frag(ds);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado[db], bp, dv[db], 0, 0, 0);
            dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aqt[db], bds, dk[db], 0, 0, 0);
        }
    }
    if (kok) {
        bf16_t* kp = p.dK + (size_t)(b * p.Lk + key) * p.lddk + h * 64 + fg * 4;
        bf16_t* vp = p.dV + (size_t)(b * p.Lk + key) * p.lddv + h * 64 + fg * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 w;
            w.x = pack2bf(dk[db][0] * p.scale, dk[db][1] * p.scale);
            w.y = pack2bf(dk[db][2] * p.scale, dk[db][3] * p.scale);
            *reinterpret_cast<uint2*>(kp + db * 16) = w;
            w.x = pack2bf(dv[db][0], dv[db][1]);
            w.y = pack2bf(dv[db][2], dv[db][3]);
            *reinterpret_cast<uint2*>(vp + db * 16) = w;
        }
    }
}

extern "C" int cris_attn_bwd_dkv(const cris_attn_params* pp, void* stream) {
    const cris_attn_params& p = *pp;
    if (attn_check(p)) return -1;
    CRIS_CHECK_ARG(p.V && p.Qt && p.dOt && p.dO && p.lse && p.delta && p.dK && p.dV, "backward(dkv) operands");
    CRIS_CHECK_ARG((p.ldv & 7) == 0 && (p.lddo & 7) == 0 && (p.lddk & 3) == 0 && (p.lddv & 3) == 0, "ld");
    CRIS_CHECK_ARG((p.Lq_pad & 3) == 0 && p.Lq_pad >= ((p.Lq + 31) / 32) * 32, "Lq_pad must cover whole 32-query tiles");
    dim3 grid(cris_cdiv(p.Lk, 64), p.B * p.Hn);
    // (the key side also serves the decoder's cross attention: 17 / 22 keys with a padding mask, but a 676 / 900-query loop)
    if (attn_use_lds(p, p.Lq, true) && (p.Lq_pad & 7) == 0) return attn_launch_lds(2, p, grid, stream);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// Long-sequence variants (decoder self-attention: 676 / 900 visual tokens, 64 heads per batch of 8).
//
// At these lengths every 16-query wave of the kernels above re-reads ALL of K / V (Q / dO) from L2: 0.5 GB of L2 reads per
// launch and one L2 round trip per 32-key step.  Here a block of AL_WAVES waves shares 64-row operand tiles through LDS:
// [64][64] bf16 tiles (128-B rows, 16-B chunk index XOR (row & 7): conflict-free 16-B / 8-B fragment reads) staged by
// LDS-DMA (buffer_load_dwordx4 ... lds, swizzle applied on the source side, rows beyond the sequence = out-of-range offsets
// = zeros) through a ring with counted vmcnt + one raw barrier per 64-row step.  A wave works on AL_G 16-row groups: round 2
// ran 2 waves x TWO groups (each fragment read from LDS feeds two MFMA chains), round 3 runs 4 waves x ONE group - the grid is only
// (L/64) x batch*heads = 704 blocks, so two waves per block left 1.4 waves per SIMD and every MFMA -> softmax -> MFMA dependency
// exposed; 2.75 waves per SIMD: forward 51 -> 39 us, dq 54 -> 44, dkv 67 -> 44 (call r03ab; LDS reads are not the limit).  The arithmetic (S^T formulation, slot
// permutation of the PV product, online softmax, dropout hash, masks of rows beyond the sequence) is exactly that of the
// kernels above; no causal mask, key-padding mask on the key-side backward only (the cross attention's 676-query loop).  One difference in bookkeeping: the rows of the two
// 16-row MFMA blocks of a 32-row half step are interleaved (block kb takes rows (m>>2)*8 + kb*4 + (m&3)), so that the 8
// reduction slots a lane owns in the second product are 8 CONSECUTIVE rows fg*8 .. fg*8+7 and every LDS fragment read is one
// 16-byte ds_read_b128 (hipcc puts an s_waitcnt vmcnt(0) - a full drain of the DMA ring - in front of merged 8-byte LDS
// reads that follow an LDS-DMA, but not in front of ds_read_b128).
// ================================================================================================
#define AL_TILE 8192                       // bytes of one [64][64] bf16 tile
#define AL_NI (8 / AL_WAVES)               // 1-KB DMA instructions per wave per tile

// stage rows row0 .. row0+63 (128 B each, from byte `base` + row * `stride`) of a matrix with `nrows` rows
__device__ __forceinline__ void al_stage(const __amdgpu_buffer_rsrc_t& rs, unsigned base, unsigned stride, int row0, int nrows,
                                         unsigned char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < AL_NI; ++i) {
        const int inst = wave * AL_NI + i;
        const int r = inst * 8 + (lane >> 3);                    // tile row; r & 7 == lane >> 3
        const unsigned chunk = (unsigned)((lane & 7) ^ (lane >> 3));
        const unsigned off = (row0 + r < nrows) ? base + (unsigned)(row0 + r) * stride + chunk * 16u : CRIS_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + inst * 1024), 16, off, 0, 0, 0);
    }
}
// 64 floats (one per lane) of a vector with `n` entries starting at index i0
__device__ __forceinline__ void al_stage_f32(const __amdgpu_buffer_rsrc_t& rs, unsigned base, int i0, int n, unsigned char* lds, int lane) {
    const unsigned off = (i0 + lane < n) ? base + (unsigned)(i0 + lane) * 4u : CRIS_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds, 4, off, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 al_ld16(const unsigned char* tile, int row, int chunk) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tile + row * 128 + (((chunk ^ row) & 7) << 4)));
}
// XCD-aware block order: consecutive workgroups go to the 8 XCDs round-robin, each with its own 4 MB L2.  The grid is
// (sequence tiles, batch*heads); with the plain order the ~11 tile blocks of one (batch, head) - which all stream the SAME
// K / V (Q / dO) panel - land on 8 different XCDs and the panel is fetched from HBM up to 8 times (round 2 PMC: 96-152 MB
// fetched per launch against ~20 MB of operands).  Here every XCD gets a contiguous run of (batch, head) pairs, so a panel
// is fetched by one L2 only.  (x, bh) out: the tile index and the batch*heads index this block works on.
__device__ __forceinline__ void al_block(int& x, int& bh) {
    const int nx = gridDim.x, total = nx * gridDim.y;
    int lin = blockIdx.y * nx + blockIdx.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = lin & 7, idx = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bh = lin / nx;
    x = lin - bh * nx;
}
// ---- forward: block = AL_WAVES x 32 queries; stage = K tile | V^T tile -------------------------------------------------
__global__ __launch_bounds__(64 * AL_WAVES) void attn_fwd_lds_kernel(const cris_attn_params p) {
    constexpr int STAGES = 3, STAGE_BYTES = 2 * AL_TILE, NDMA = 2 * AL_NI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bx, bh;
    al_block(bx, bh);
    const int b = bh / p.Hn, h = bh - b * p.Hn;
    const int q0w = bx * (16 * AL_G * AL_WAVES) + wave * (16 * AL_G);

    int q[2];
    bool qok[2];
    bf16x8 bq[2][2];
    uint32_t didx0[2];
#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        q[g] = q0w + g * 16 + fr;
        qok[g] = q[g] < p.Lq;
        const bf16_t* Qp = p.Q + (size_t)(b * p.Lq + min(q[g], p.Lq - 1)) * p.ldq + h * 64 + fg * 8;   // clamped: masked by !qok
        bq[g][0] = ld_frag16(Qp);
        bq[g][1] = ld_frag16(Qp + 32);
        didx0[g] = ((uint32_t)bh * (uint32_t)p.Lq + (uint32_t)q[g]) * (uint32_t)p.Lk;
    }
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.K), 0, (int)((size_t)p.B * p.Lk * p.ldk * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsVt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.Vt), 0, (int)((size_t)p.B * p.Hn * 64 * p.Lk_pad * 2), CRIS_BUF_FLAGS);
    const unsigned baseK = ((unsigned)b * (unsigned)p.Lk * (unsigned)p.ldk + (unsigned)h * 64u) * 2u;
    const unsigned baseVt = (unsigned)bh * 64u * (unsigned)p.Lk_pad * 2u;
    const int nsteps = (p.Lk + 63) / 64;
    auto issue = [&](int s, int kt) {
        unsigned char* st = smem + s * STAGE_BYTES;
        al_stage(rsK, baseK, (unsigned)p.ldk * 2u, kt * 64, p.Lk, st, wave, lane);
        al_stage(rsVt, baseVt + (unsigned)kt * 128u, (unsigned)p.Lk_pad * 2u, 0, kt < nsteps ? 64 : 0, st + AL_TILE, wave, lane);
    };

    const bool has_drop = p.drop_thresh > 0u;
    const uint32_t dkey = cris_drop_key(p.drop_seed + (p.drop_seed_dev ? p.drop_seed_dev[0] : 0u), p.drop_stream);
    const float inv_keep = has_drop ? 1.f / (1.f - p.drop_p) : 1.f;
    // pre-mixed hash word of (query row, key fg*8) per group; key k0 + j adds (k0 + j) * CRIS_DROP_MUL (see cris_keep_h)
    uint32_t dh0[2];
#pragma unroll
    for (int g = 0; g < AL_G; ++g) dh0[g] = (didx0[g] + (uint32_t)(fg * 8)) * CRIS_DROP_MUL + dkey;

    f32x4 o[2][4];
    float m_run[2], l_run[2];
#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        m_run[g] = NEG_INF;
        l_run[g] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[g][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        issue(s, s);
    }
    int buf = 0;
    for (int kt = 0; kt < nsteps; ++kt) {
        CRIS_VMCNT((STAGES - 2) * NDMA);            // this wave's share of step kt has landed ...
        __builtin_amdgcn_s_barrier();               // ... and everyone's; everyone is also done reading step kt-1
        {
            int nb = buf + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            issue(nb, kt + STAGES - 1);
        }
        const unsigned char* tK = smem + buf * STAGE_BYTES;
        const unsigned char* tV = tK + AL_TILE;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k0 = kt * 64 + half * 32;
            const bool tail = k0 + 32 > p.Lk;
            const uint32_t k0m = (uint32_t)k0 * CRIS_DROP_MUL;
            bf16x8 av[4], ak[2][2];
#pragma unroll
            for (int db = 0; db < 4; ++db) av[db] = al_ld16(tV, db * 16 + fr, half * 4 + fg);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int row = half * 32 + (fr >> 2) * 8 + kb * 4 + (fr & 3);      // interleaved blocks (see above)
                ak[kb][0] = al_ld16(tK, row, fg);
                ak[kb][1] = al_ld16(tK, row, fg + 4);
            }
#pragma unroll
            for (int g = 0; g < AL_G; ++g) {
                float s[8];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f};
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[kb][0], bq[g][0], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[kb][1], bq[g][1], st, 0, 0, 0);
                    if (tail) {                                   // (uniform) only the last key tile has keys beyond Lk
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kk = k0 + fg * 8 + kb * 4 + r;
                            s[kb * 4 + r] = kk >= p.Lk ? NEG_INF : st[r] * p.scale;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[kb * 4 + r] = st[r] * p.scale;
                    }
                }
                float mx = s[0];
#pragma unroll
                for (int j = 1; j < 8; ++j) mx = fmaxf(mx, s[j]);
                mx = grp_max(mx);
                const float m_new = fmaxf(m_run[g], mx);
                float pv[8];
                float alpha = 1.f, rs = 0.f;
                if (m_new == NEG_INF) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = 0.f;
                } else {
                    alpha = __expf(m_run[g] - m_new);             // m_run = -inf -> 0
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        pv[j] = __expf(s[j] - m_new);             // masked -> 0
                        rs += pv[j];
                    }
                }
                rs = grp_sum(rs);
                l_run[g] = l_run[g] * alpha + rs;
                m_run[g] = m_new;
                if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {      // (uniform) the running maximum of some row moved: rescale
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[g][i][r] *= alpha;
                }
                if (has_drop) {
                    const uint32_t hb = dh0[g] + k0m;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        pv[j] = cris_keep_h(hb + (uint32_t)j * CRIS_DROP_MUL, p.drop_thresh) ? pv[j] * inv_keep : 0.f;
                }
                const bf16x8 bp = pack_frag(pv);
#pragma unroll
                for (int db = 0; db < 4; ++db) o[g][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[db], bp, o[g][db], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++buf == STAGES) buf = 0;
    }
    CRIS_VMCNT(0);                                  // drain the (out-of-range) tail DMAs before the block retires

#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        const float inv_l = l_run[g] > 0.f ? 1.f / l_run[g] : 0.f;
        if (qok[g]) {
            bf16_t* op = p.O + (size_t)(b * p.Lq + q[g]) * p.ldo + h * 64 + fg * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 w;
                w.x = pack2bf(o[g][db][0] * inv_l, o[g][db][1] * inv_l);
                w.y = pack2bf(o[g][db][2] * inv_l, o[g][db][3] * inv_l);
                *reinterpret_cast<uint2*>(op + db * 16) = w;
            }
            if (fg == 0 && p.lse) p.lse[(size_t)bh * p.Lq + q[g]] = l_run[g] > 0.f ? m_run[g] + __logf(l_run[g]) : __builtin_inff();
        }
    }
}

// ---- backward, query side: block = AL_WAVES x 32 queries; stage = K tile | V tile | K^T tile -------------------------------
__global__ __launch_bounds__(64 * AL_WAVES) void attn_bwd_dq_lds_kernel(const cris_attn_params p) {
    constexpr int STAGES = 2, STAGE_BYTES = 3 * AL_TILE, NDMA = 3 * AL_NI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bx, bh;
    al_block(bx, bh);
    const int b = bh / p.Hn, h = bh - b * p.Hn;
    const int q0w = bx * (16 * AL_G * AL_WAVES) + wave * (16 * AL_G);

    int q[2];
    bool qok[2];
    bf16x8 bq[2][2], bd[2][2];
    float delta[2], lse[2];
    uint32_t didx0[2];
#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        q[g] = q0w + g * 16 + fr;
        qok[g] = q[g] < p.Lq;
        const int qcl = min(q[g], p.Lq - 1);                   // rows beyond Lq read the last row; they are masked (!qok)
        const bf16_t* Qp = p.Q + (size_t)(b * p.Lq + qcl) * p.ldq + h * 64 + fg * 8;
        bq[g][0] = ld_frag16(Qp);
        bq[g][1] = ld_frag16(Qp + 32);
        const bf16_t* dOp = p.dO + (size_t)(b * p.Lq + qcl) * p.lddo + h * 64 + fg * 8;
        bd[g][0] = ld_frag16(dOp);
        bd[g][1] = ld_frag16(dOp + 32);
        const bf16_t* Op = p.O + (size_t)(b * p.Lq + qcl) * p.ldo + h * 64 + fg * 8;
        const uint4 o0 = *reinterpret_cast<const uint4*>(Op), o1 = *reinterpret_cast<const uint4*>(Op + 32);
        float a[8], c[8], dl = 0.f;
        unpack8(__builtin_bit_cast(uint4, bd[g][0]), a);
        unpack8(o0, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += a[j] * c[j];
        unpack8(__builtin_bit_cast(uint4, bd[g][1]), a);
        unpack8(o1, c);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += a[j] * c[j];
        dl = grp_sum(dl);
        delta[g] = dl;
        if (qok[g] && fg == 0) p.delta[(size_t)bh * p.Lq + q[g]] = dl;
        lse[g] = qok[g] ? p.lse[(size_t)bh * p.Lq + qcl] : __builtin_inff();      // rows beyond Lq: exp(s - inf) = 0, no test per element
        didx0[g] = ((uint32_t)bh * (uint32_t)p.Lq + (uint32_t)q[g]) * (uint32_t)p.Lk;
    }
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.K), 0, (int)((size_t)p.B * p.Lk * p.ldk * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.V), 0, (int)((size_t)p.B * p.Lk * p.ldv * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsKt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.Kt), 0, (int)((size_t)p.B * p.Hn * 64 * p.Lk_pad * 2), CRIS_BUF_FLAGS);
    const unsigned baseK = ((unsigned)b * (unsigned)p.Lk * (unsigned)p.ldk + (unsigned)h * 64u) * 2u;
    const unsigned baseV = ((unsigned)b * (unsigned)p.Lk * (unsigned)p.ldv + (unsigned)h * 64u) * 2u;
    const unsigned baseKt = (unsigned)bh * 64u * (unsigned)p.Lk_pad * 2u;
    const int nsteps = (p.Lk + 63) / 64;
    auto issue = [&](int s, int kt) {
        unsigned char* st = smem + s * STAGE_BYTES;
        al_stage(rsK, baseK, (unsigned)p.ldk * 2u, kt * 64, p.Lk, st, wave, lane);
        al_stage(rsV, baseV, (unsigned)p.ldv * 2u, kt * 64, p.Lk, st + AL_TILE, wave, lane);
        al_stage(rsKt, baseKt + (unsigned)kt * 128u, (unsigned)p.Lk_pad * 2u, 0, kt < nsteps ? 64 : 0, st + 2 * AL_TILE, wave, lane);
    };
    const bool has_drop = p.drop_thresh > 0u;
    const uint32_t dkey = cris_drop_key(p.drop_seed + (p.drop_seed_dev ? p.drop_seed_dev[0] : 0u), p.drop_stream);
    const float inv_keep = has_drop ? 1.f / (1.f - p.drop_p) : 1.f;

    uint32_t dh0[2];                              // pre-mixed hash words, as in the forward kernel
#pragma unroll
    for (int g = 0; g < AL_G; ++g) dh0[g] = (didx0[g] + (uint32_t)(fg * 8)) * CRIS_DROP_MUL + dkey;
    f32x4 dq[2][4];
#pragma unroll
    for (int g = 0; g < AL_G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[g][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        issue(s, s);
    }
    int buf = 0;
    for (int kt = 0; kt < nsteps; ++kt) {
        CRIS_VMCNT((STAGES - 2) * NDMA);
        __builtin_amdgcn_s_barrier();
        {
            int nb = buf + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            issue(nb, kt + STAGES - 1);
        }
        const unsigned char* tK = smem + buf * STAGE_BYTES;
        const unsigned char* tV = tK + AL_TILE;
        const unsigned char* tKt = tK + 2 * AL_TILE;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int k0 = kt * 64 + half * 32;
            const bool tail = k0 + 32 > p.Lk;
            const uint32_t k0m = (uint32_t)k0 * CRIS_DROP_MUL;
            bf16x8 akt[4], ak[2][2], avv[2][2];
#pragma unroll
            for (int db = 0; db < 4; ++db) akt[db] = al_ld16(tKt, db * 16 + fr, half * 4 + fg);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int row = half * 32 + (fr >> 2) * 8 + kb * 4 + (fr & 3);
                ak[kb][0] = al_ld16(tK, row, fg);
                ak[kb][1] = al_ld16(tK, row, fg + 4);
                avv[kb][0] = al_ld16(tV, row, fg);
                avv[kb][1] = al_ld16(tV, row, fg + 4);
            }
#pragma unroll
            for (int g = 0; g < AL_G; ++g) {
                float ds[8];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = st;
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[kb][0], bq[g][0], st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak[kb][1], bq[g][1], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avv[kb][0], bd[g][0], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avv[kb][1], bd[g][1], dp, 0, 0, 0);
                    const uint32_t hb = dh0[g] + k0m + (uint32_t)(kb * 4) * CRIS_DROP_MUL;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pr = __expf(st[r] * p.scale - lse[g]);
                        if (tail) pr = (k0 + fg * 8 + kb * 4 + r) >= p.Lk ? 0.f : pr;      // (uniform) last key tile only
                        float dpv = dp[r];
                        if (has_drop) dpv = cris_keep_h(hb + (uint32_t)r * CRIS_DROP_MUL, p.drop_thresh) ? dpv * inv_keep : 0.f;
                        ds[kb * 4 + r] = pr * (dpv - delta[g]);
                    }
                }
                const bf16x8 bds = pack_frag(ds);
#pragma unroll
                for (int db = 0; db < 4; ++db) dq[g][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(akt[db], bds, dq[g][db], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++buf == STAGES) buf = 0;
    }
    CRIS_VMCNT(0);
#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        if (qok[g]) {
            bf16_t* op = p.dQ + (size_t)(b * p.Lq + q[g]) * p.lddq + h * 64 + fg * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 w;
                w.x = pack2bf(dq[g][db][0] * p.scale, dq[g][db][1] * p.scale);
                w.y = pack2bf(dq[g][db][2] * p.scale, dq[g][db][3] * p.scale);
                *reinterpret_cast<uint2*>(op + db * 16) = w;
            }
        }
    }
}

// ---- backward, key side: block = AL_WAVES x 32 keys; stage = Q tile | dO tile | Q^T tile | dO^T tile | lse[64] | delta[64] ----
__global__ __launch_bounds__(64 * AL_WAVES) void attn_bwd_dkv_lds_kernel(const cris_attn_params p) {
    constexpr int STAGES = 2, STAGE_BYTES = 4 * AL_TILE + 512, NDMA = 4 * AL_NI + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bx, bh;
    al_block(bx, bh);
    const int b = bh / p.Hn, h = bh - b * p.Hn;
    const int k0w = bx * (16 * AL_G * AL_WAVES) + wave * (16 * AL_G);

    int key[2];
    bool kok[2], kpad[2];
    bf16x8 bk[2][2], bv[2][2];
#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        key[g] = k0w + g * 16 + fr;
        kok[g] = key[g] < p.Lk;
        const int keyc = min(key[g], p.Lk - 1);                // keys beyond Lk read the last key; nothing is stored for them
        kpad[g] = p.key_tokens != nullptr && p.key_tokens[(size_t)b * p.Lk + keyc] == 0;      // padded key: attends nothing
        const bf16_t* Kp = p.K + (size_t)(b * p.Lk + keyc) * p.ldk + h * 64 + fg * 8;
        const bf16_t* Vp = p.V + (size_t)(b * p.Lk + keyc) * p.ldv + h * 64 + fg * 8;
        bk[g][0] = ld_frag16(Kp);
        bk[g][1] = ld_frag16(Kp + 32);
        bv[g][0] = ld_frag16(Vp);
        bv[g][1] = ld_frag16(Vp + 32);
    }
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Q), 0, (int)((size_t)p.B * p.Lq * p.ldq * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsdO = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dO), 0, (int)((size_t)p.B * p.Lq * p.lddo * 2),
                                                                         CRIS_BUF_FLAGS);
    const int tbytes = (int)((size_t)p.B * p.Hn * 64 * p.Lq_pad * 2);
    const __amdgpu_buffer_rsrc_t rsQt = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Qt), 0, tbytes, CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsdOt = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dOt), 0, tbytes, CRIS_BUF_FLAGS);
    const int vbytes = (int)((size_t)p.B * p.Hn * p.Lq * 4);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.lse), 0, vbytes, CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.delta), 0, vbytes, CRIS_BUF_FLAGS);
    const unsigned baseQ = ((unsigned)b * (unsigned)p.Lq * (unsigned)p.ldq + (unsigned)h * 64u) * 2u;
    const unsigned basedO = ((unsigned)b * (unsigned)p.Lq * (unsigned)p.lddo + (unsigned)h * 64u) * 2u;
    const unsigned baseT = (unsigned)bh * 64u * (unsigned)p.Lq_pad * 2u;
    const unsigned baseL = (unsigned)bh * (unsigned)p.Lq * 4u;
    const int nsteps = (p.Lq + 63) / 64;
    auto issue = [&](int s, int qt) {
        unsigned char* st = smem + s * STAGE_BYTES;
        al_stage(rsQ, baseQ, (unsigned)p.ldq * 2u, qt * 64, p.Lq, st, wave, lane);
        al_stage(rsdO, basedO, (unsigned)p.lddo * 2u, qt * 64, p.Lq, st + AL_TILE, wave, lane);
        al_stage(rsQt, baseT + (unsigned)qt * 128u, (unsigned)p.Lq_pad * 2u, 0, qt < nsteps ? 64 : 0, st + 2 * AL_TILE, wave, lane);
        al_stage(rsdOt, baseT + (unsigned)qt * 128u, (unsigned)p.Lq_pad * 2u, 0, qt < nsteps ? 64 : 0, st + 3 * AL_TILE, wave, lane);
        al_stage_f32(rsL, baseL, qt * 64, p.Lq, st + 4 * AL_TILE, lane);          // (every wave stages the same 64 values)
        al_stage_f32(rsD, baseL, qt * 64, p.Lq, st + 4 * AL_TILE + 256, lane);
    };
    const bool has_drop = p.drop_thresh > 0u;
    const uint32_t dkey = cris_drop_key(p.drop_seed + (p.drop_seed_dev ? p.drop_seed_dev[0] : 0u), p.drop_stream);
    const float inv_keep = has_drop ? 1.f / (1.f - p.drop_p) : 1.f;

    // dropout index of (query qq, key) = (bh*Lq + qq)*Lk + key with qq = q0 + fg*8 + j: the pre-mixed word of (fg*8, key) per
    // group; q0 + j adds (q0 + j) * (Lk * CRIS_DROP_MUL), a scalar
    const uint32_t LkM = (uint32_t)p.Lk * CRIS_DROP_MUL;
    uint32_t dh0[2];
    bool kvalid[2];
#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        dh0[g] = (((uint32_t)bh * (uint32_t)p.Lq + (uint32_t)(fg * 8)) * (uint32_t)p.Lk + (uint32_t)key[g]) * CRIS_DROP_MUL + dkey;
        kvalid[g] = kok[g] && !kpad[g];
    }
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int g = 0; g < AL_G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) dk[g][i] = dv[g][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        issue(s, s);
    }
    int buf = 0;
    for (int qt = 0; qt < nsteps; ++qt) {
        CRIS_VMCNT((STAGES - 2) * NDMA);
        __builtin_amdgcn_s_barrier();
        {
            int nb = buf + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            issue(nb, qt + STAGES - 1);
        }
        const unsigned char* tQ = smem + buf * STAGE_BYTES;
        const unsigned char* tdO = tQ + AL_TILE;
        const unsigned char* tQt = tQ + 2 * AL_TILE;
        const unsigned char* tdOt = tQ + 3 * AL_TILE;
        const float* s_lse = reinterpret_cast<const float*>(tQ + 4 * AL_TILE);
        const float* s_del = s_lse + 64;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q0 = qt * 64 + half * 32;
            const bool tail = q0 + 32 > p.Lq;
            const uint32_t q0m = (uint32_t)q0 * LkM;
            bf16x8 ado[4], aqt[4], aq[2][2], ad[2][2];
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                ado[db] = al_ld16(tdOt, db * 16 + fr, half * 4 + fg);
                aqt[db] = al_ld16(tQt, db * 16 + fr, half * 4 + fg);
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int row = half * 32 + (fr >> 2) * 8 + qb * 4 + (fr & 3);
                aq[qb][0] = al_ld16(tQ, row, fg);
                aq[qb][1] = al_ld16(tQ, row, fg + 4);
                ad[qb][0] = al_ld16(tdO, row, fg);
                ad[qb][1] = al_ld16(tdO, row, fg + 4);
            }
            float lq[8], dl[8];                          // lse / delta of this lane's 8 C/D rows (queries)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lq_i = half * 32 + fg * 8 + j;
                lq[j] = s_lse[lq_i];
                dl[j] = s_del[lq_i];
            }
#pragma unroll
            for (int g = 0; g < AL_G; ++g) {
                float pd[8], ds[8];
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = sv;
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[qb][0], bk[g][0], sv, 0, 0, 0);
                    sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[qb][1], bk[g][1], sv, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad[qb][0], bv[g][0], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad[qb][1], bv[g][1], dp, 0, 0, 0);
                    const uint32_t hb = dh0[g] + q0m + (uint32_t)(qb * 4) * LkM;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // C/D row of this lane = query qq = q0 + fg*8 + qb*4 + r (interleaved blocks).  pr = 0 for keys that are
                        // beyond Lk or padded (per lane) and, in the last query tile only (uniform), for rows beyond Lq; with pr = 0
                        // the products below are 0 whatever delta is (it is finite), so delta needs no select of its own
                        float pr = kvalid[g] ? __expf(sv[r] * p.scale - lq[qb * 4 + r]) : 0.f;
                        if (tail) pr = (q0 + fg * 8 + qb * 4 + r) < p.Lq ? pr : 0.f;
                        float dpv = dp[r];
                        float prd = pr;
                        if (has_drop) {
                            const bool keep = cris_keep_h(hb + (uint32_t)r * LkM, p.drop_thresh);
                            prd = keep ? pr * inv_keep : 0.f;
                            dpv = keep ? dpv * inv_keep : 0.f;
                        }
                        pd[qb * 4 + r] = prd;
                        ds[qb * 4 + r] = pr * (dpv - dl[qb * 4 + r]);
                    }
                }
                const bf16x8 bp = pack_frag(pd), bds = pack_frag(ds);
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    dv[g][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado[db], bp, dv[g][db], 0, 0, 0);
                    dk[g][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aqt[db], bds, dk[g][db], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++buf == STAGES) buf = 0;
    }
    CRIS_VMCNT(0);
#pragma unroll
    for (int g = 0; g < AL_G; ++g) {
        if (kok[g]) {
            bf16_t* kp = p.dK + (size_t)(b * p.Lk + key[g]) * p.lddk + h * 64 + fg * 4;
            bf16_t* vp = p.dV + (size_t)(b * p.Lk + key[g]) * p.lddv + h * 64 + fg * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 w;
                w.x = pack2bf(dk[g][db][0] * p.scale, dk[g][db][1] * p.scale);
                w.y = pack2bf(dk[g][db][2] * p.scale, dk[g][db][3] * p.scale);
                *reinterpret_cast<uint2*>(kp + db * 16) = w;
                w.x = pack2bf(dv[g][db][0], dv[g][db][1]);
                w.y = pack2bf(dv[g][db][2], dv[g][db][3]);
                *reinterpret_cast<uint2*>(vp + db * 16) = w;
            }
        }
    }
}
