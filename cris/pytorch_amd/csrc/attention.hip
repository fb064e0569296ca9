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

extern "C" int cris_attn_fwd(const cris_attn_params* pp, void* stream) {
    const cris_attn_params& p = *pp;
    if (attn_check(p)) return -1;
    CRIS_CHECK_ARG(p.Vt && p.O && (p.ldo & 3) == 0, "forward operands");
    dim3 grid(cris_cdiv(p.Lq, 64), p.B * p.Hn);
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
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}
