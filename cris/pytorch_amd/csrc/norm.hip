// BatchNorm (training statistics) and LayerNorm kernels for gfx950.  All HBM-bound: 16-byte vector
// accesses (8 bf16 channels per lane), per-channel reductions accumulated in registers, then LDS (fixed order), then one
// partial row per block that a second small launch sums in block order - no atomics, results are deterministic.
#include "common.h"
#include "../../../include/cris_hip.h"
#include "p2p_ll.h"
#include <string.h>

// ------------------------------------------------------------------------------------------------
// BN coefficients
// ------------------------------------------------------------------------------------------------
#define BN_WIDE_MIN 128         // partial lists longer than this are merged by 64 part lanes per channel (1024-thread blocks)
#define BN_MERGE_MIN 512        // ... and lists longer than this by a first-level merge launch into BN_MERGE_SLICES slices
#define BN_MERGE_SLICES 64

// sum of one value per (part lane, channel) over the PL part lanes of a block, returned to every lane of the channel; a fixed
// order (PL = 64: eight groups of eight lanes, then the eight group sums), so the result is deterministic
template <int PL>
__device__ __forceinline__ float bn_lane_sum(float (*sh)[17], float v, int pl, int cl) {
    __syncthreads();                        // the previous use of sh is over
    sh[pl][cl] = v;
    __syncthreads();
    if (PL == 64) {
        float g = 0.f;
        if (pl < 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g += sh[pl * 8 + j][cl];
        }
        __syncthreads();
        if (pl < 8) sh[pl][cl] = g;
        __syncthreads();
    }
    float t = sh[0][cl];
#pragma unroll
    for (int j = 1; j < (PL == 64 ? 8 : PL); ++j) t += sh[j][cl];
    return t;
}

// level 1 for the longest lists (the stem and layer-1 outputs: 676 - 2704 parts): slice s merges parts [s*pps, (s+1)*pps) into
// one (sum, M2 about the slice mean) row, so that many CUs read the list - a single finalize launch has only C/16 blocks (4 at
// C = 64) and one CU's load throughput then bounds it (measured: 50 us for 2704 parts against ~8 + 5 us for the two launches).
// Block = 16 channels x 16 part lanes (as the finalize kernel: 64 x C/16 blocks), the same two-pass, division-free merge.
__global__ __launch_bounds__(256) void bn_merge_kernel(const float* __restrict__ psum, const float* __restrict__ pm2, int nparts,
                                                       int rows_per_part, int M, int C, int pps, float* __restrict__ osum,
                                                       float* __restrict__ om2) {
    __shared__ float sh[16][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl, cc = min(c, C - 1);
    const int s = blockIdx.y;
    const int i_begin = s * pps, i_end = min(nparts, (s + 1) * pps), last = nparts - 1;
    const float n_full = (float)rows_per_part, n_last = (float)(M - last * rows_per_part);
    const float n_slice = (float)(min(M, i_end * rows_per_part) - i_begin * rows_per_part);
    // both columns of this lane's first four entries (a slice is a few entries per lane: normally ALL of them) are requested up
    // front and kept in registers for the second pass (round 6: its M2 loads used to wait for the mean - a second cold round trip)
    float ps0[4], pq0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t o = (size_t)min(i_begin + pl + 16 * u, i_end - 1) * C + cc;
        ps0[u] = psum[o];
        pq0[u] = pm2[o];
    }
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (i_begin + pl + 16 * u < i_end) t += ps0[u];
    for (int i0 = i_begin + pl + 64; i0 < i_end; i0 += 64) {
        float ps[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ps[u] = psum[(size_t)min(i0 + 16 * u, i_end - 1) * C + cc];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 16 * u < i_end) t += ps[u];
    }
    const float total = bn_lane_sum<16>(sh, t, pl, cl);
    const float mean = total / n_slice;
    const float inv_full = 1.f / n_full, inv_last = 1.f / n_last;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i_begin + pl + 16 * u;
        if (i < i_end) {
            const float d = ps0[u] * (i == last ? inv_last : inv_full) - mean;
            q += pq0[u] + (i == last ? n_last : n_full) * d * d;
        }
    }
    for (int i0 = i_begin + pl + 64; i0 < i_end; i0 += 64) {
        float ps[4], pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = (size_t)min(i0 + 16 * u, i_end - 1) * C + cc;
            ps[u] = psum[o];
            pq[u] = pm2[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 16 * u;
            if (i < i_end) {
                const float d = ps[u] * (i == last ? inv_last : inv_full) - mean;
                q += pq[u] + (i == last ? n_last : n_full) * d * d;
            }
        }
    }
    const float m2 = bn_lane_sum<16>(sh, q, pl, cl);
    if (pl == 0 && c < C) {
        osum[(size_t)s * C + c] = total;
        om2[(size_t)s * C + c] = m2;
    }
}

// The arithmetic of the single-exchange SyncBN, shared by the separate pack / unpack kernels and by the finalize kernel that
// exchanges in place: explicit roundings (no FMA contraction left to the compiler), so every form gives the same bits.
//   pack:    S1 = sum - n c,  S2 = M2 + n (mean_l - c)^2        (moments about the reference c = running mean)
//   unpack:  global sum = S1 + N c,  M2 about the global mean = max(S2 - S1^2 / N, 0)
__device__ __forceinline__ void bn_sync_pack_vals(float total, float m2, float mean_l, float ref, float n, float& s1, float& s2) {
    const float d = __fsub_rn(mean_l, ref);
    s1 = __fmaf_rn(-n, ref, total);
    s2 = __fmaf_rn(__fmul_rn(n, d), d, m2);
}
__device__ __forceinline__ void bn_sync_unpack_vals(float s1, float s2, float ref, float count, float& gsum, float& gm2) {
    gsum = __fmaf_rn(count, ref, s1);
    gm2 = fmaxf(__fsub_rn(s2, __fdiv_rn(__fmul_rn(s1, s1), count)), 0.f);
}

// merge + coefficients in ONE launch: block = 16 channels x PL part lanes (coalesced 64-B rows).  Two passes over the partial
// list (sum, M2 about the part mean; all parts rows_per_part rows, the last one what is left of count_local):
//   mean = sum_i S_i / n,   M2 = sum_i [ M2_i + n_i (S_i / n_i - mean)^2 ]
// - the exact decomposition of the sum of squares about the common mean (no E[x^2] - mean^2 cancellation) and, unlike a chain
// of pairwise (Chan) updates, free of divisions inside the loops: with 64 lanes x 16 channels in a block the chain's three
// divisions per entry made the kernel VALU-bound on its one CU (14 us per launch, 71 us for the 2704-part list of the stem).
// The second pass re-reads the list (L2 hits).  PL = 64 serves lists of 129 - 512 parts; longer ones come through bn_merge_kernel.
// The kernel is a handful of blocks on an otherwise idle chip, i.e. memory latency: the per-channel parameters are requested
// before the list is walked, and the list in batches of eight entries whose loads are issued together.
template <int PL>
__global__ __launch_bounds__(16 * PL) void bn_finalize_kernel(const float* psum, const float* pm2, int nparts, int rows_per_part, float count_local,
                                   float count, const float* gamma, const float* beta, float* rmean, float* rvar,
                                   float momentum, float eps, int C, float* scale, float* shift, float* mean_o,
                                   float* invstd_o, float* merged /* optional [2*C]: local (sum, M2) for SyncBN */,
                                   const float* global_stats /* optional [2*C]: (sum, M2 about the global mean) */,
                                   const cris_p2p_link link /* world > 1: SyncBN, the exchange happens in here */) {
    __shared__ float sh[PL][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int cc = min(c, C - 1);            // lanes beyond C read channel C-1 and store nothing
    float mean, m2;
    float gam = 0.f, bet = 0.f, rm0 = 0.f, rv0 = 0.f;
    if (pl == 0 && c < C && !merged) {
        gam = gamma[c];
        bet = beta[c];
        if (rmean) { rm0 = rmean[c]; rv0 = rvar[c]; }
    }
    if (!global_stats) {
        // Round 6: the launcher guarantees nparts <= 8 * PL (longer lists come through bn_merge_kernel), so each lane owns at
        // most eight entries: BOTH columns of all of them are requested up front and kept in registers - the second pass used to
        // issue its M2 loads only after the mean was known, a second cold round trip to memory (the partials were written by
        // other CUs' GEMM epilogues) in a kernel that is nothing but latency.  Same operations in the same order: same bits.
        float ps[8], pq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t o = (size_t)min(pl + PL * u, nparts - 1) * C + cc;
            ps[u] = psum[o];
            pq[u] = pm2[o];
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (pl + PL * u < nparts) s += ps[u];
        const float total = bn_lane_sum<PL>(sh, s, pl, cl);
        mean = total / count_local;
        const int last = nparts - 1;
        const float n_full = (float)rows_per_part, n_last = count_local - (float)last * n_full;
        const float inv_full = 1.f / n_full, inv_last = 1.f / n_last;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = pl + PL * u;
            if (i < nparts) {
                const float d = ps[u] * (i == last ? inv_last : inv_full) - mean;
                q += pq[u] + (i == last ? n_last : n_full) * d * d;
            }
        }
        m2 = bn_lane_sum<PL>(sh, q, pl, cl);
        if (pl != 0 || c >= C) return;
        if (merged) {                       // hand the local (sum, M2) to the SyncBN exchange; finalize runs again after it
            merged[c] = total;
            merged[C + c] = m2;
            mean_o[c] = mean;               // local mean, needed to re-centre M2 about the global mean
            return;
        }
        if (link.world > 0) {
            // SyncBatchNorm in this launch: the arithmetic of bn_sync_pack / rank-order sum / bn_sync_unpack / the
            // global_stats branch below, expression for expression (the two paths give the same bits), with the exchange done by
            // the lane that owns the channel: moments about the running mean (identical on every rank) out, sums over ranks in
            const int gen = p2p_link_gen(link);
            float s1, s2, gsum, gm2;
            bn_sync_pack_vals(total, m2, mean, rm0, count_local, s1, s2);
            p2p_ll_send(link, gen, c, s1);
            p2p_ll_send(link, gen, C + c, s2);
            bool bad = false;
            s1 = p2p_ll_recv_sum(link, gen, c, bad);
            s2 = p2p_ll_recv_sum(link, gen, C + c, bad);
            bn_sync_unpack_vals(s1, s2, rm0, count, gsum, gm2);
            mean = __fdiv_rn(gsum, count);
            m2 = gm2;
            if (bad) {                                                      // a missing peer must not pass silently
                mean = __int_as_float(0x7fc00000);
                if (link.err) link.err[0] = 1;
            }
        }
    } else {
        if (pl != 0 || c >= C) return;
        mean = __fdiv_rn(global_stats[c], count);
        m2 = global_stats[C + c];
    }
    const float var = fmaxf(m2 / count, 0.f);
    const float inv = rsqrtf(var + eps);
    const float sc = gam * inv;
    scale[c] = sc;
    shift[c] = bet - mean * sc;
    mean_o[c] = mean;
    invstd_o[c] = inv;
    if (rmean) {
        const float unb = count > 1.f ? var * (count / (count - 1.f)) : var;
        rmean[c] = (1.f - momentum) * rm0 + momentum * mean;
        rvar[c] = (1.f - momentum) * rv0 + momentum * unb;
    }
}

// `link.world > 1` (SyncBatchNorm backward): the column sums are this rank's (sum g, sum g xhat): they are added into `local_acc`
// (the gradient arena's d beta / d gamma block), exchanged, and the sums over all ranks are written to `out`
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int nparts, int ncol, float* __restrict__ out,
                                                           float* __restrict__ local_acc, const cris_p2p_link link) {
    __shared__ float sh[16][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool sync = link.world > 0;        // (a world of ONE still goes through the mailbox: tools/dist1_check.py measures the exchange's cost that way)
    // latency-bound (a few blocks, a few loads each): the value to add to and this lane's rows are all requested up front
    float o = 0.f;
    if (pl == 0 && c < ncol) o = sync ? (local_acc ? local_acc[c] : 0.f) : out[c];
    float a = 0.f;
    if (c < ncol) {
        for (int i0 = pl; i0 < nparts; i0 += 64) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = part[(size_t)min(i0 + 16 * u, nparts - 1) * ncol + c];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + 16 * u < nparts) a += v[u];
        }
    }
    sh[pl][cl] = a;
    __syncthreads();
    if (pl == 0 && c < ncol) {
#pragma unroll
        for (int j = 1; j < 16; ++j) a += sh[j][cl];
        if (!sync) {
            out[c] = o + a;
        } else {
            const float loc = 0.f + a;                  // (what the plain form leaves in a zeroed `out`)
            if (local_acc) local_acc[c] = o + 1.0f * loc;   // (the axpy the exchange-by-collective path runs: dst += 1 * src)
            const int gen = p2p_link_gen(link);
            p2p_ll_send(link, gen, c, loc);
            bool bad = false;
            const float g = p2p_ll_recv_sum(link, gen, c, bad);
            out[c] = bad ? __int_as_float(0x7fc00000) : g;
            if (bad && link.err) link.err[0] = 1;
        }
    }
}
static cris_p2p_link cris_no_link() {
    cris_p2p_link l;
    memset(&l, 0, sizeof(l));
    return l;
}
void cris_launch_sum_partials(const float* part, int nparts, int ncol, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(cris_cdiv(ncol, 16)), dim3(256), 0, stream, part, nparts, ncol, out, (float*)nullptr,
                       cris_no_link());
}
// rows the caller must allocate for a partials buffer of `nparts` parts (room for the level-1 merge output of long lists)
extern "C" int cris_bn_partials_rows(int nparts) { return nparts > BN_MERGE_MIN ? nparts + BN_MERGE_SLICES : nparts; }

static int bn_finalize_launch(const float* psum, const float* pm2, int nparts, int rows_per_part, float count_local,
                              float count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, int C, float* scale, float* shift, float* mean, float* invstd,
                              float* merged, const float* global_stats, const cris_p2p_link& link, void* stream) {
    CRIS_CHECK_ARG((global_stats || (psum && pm2 && nparts > 0 && rows_per_part > 0)) && gamma && beta && mean && C > 0 && count > 0.f, "bad args");
    CRIS_CHECK_ARG(merged || (scale && shift && invstd), "bad args");
    CRIS_CHECK_ARG(global_stats || ((double)(nparts - 1) * rows_per_part < count_local && (double)nparts * rows_per_part >= count_local),
                   "the partial list must cover count_local rows: nparts = ceil(count_local / rows_per_part)");
    if (!global_stats && nparts > BN_MERGE_MIN) {
        // two levels: BN_MERGE_SLICES slices written behind the partials (rows nparts .. nparts+slices), merged below
        const int pps = cris_cdiv(nparts, BN_MERGE_SLICES);
        const int slices = cris_cdiv(nparts, pps);
        float* osum = const_cast<float*>(psum) + (size_t)nparts * C;
        float* om2 = const_cast<float*>(pm2) + (size_t)nparts * C;
        hipLaunchKernelGGL(bn_merge_kernel, dim3(cris_cdiv(C, 16), slices), dim3(256), 0, (hipStream_t)stream, psum, pm2, nparts,
                           rows_per_part, (int)count_local, C, pps, osum, om2);
        CRIS_LAUNCH_CHECK();
        psum = osum;
        pm2 = om2;
        nparts = slices;
        rows_per_part *= pps;
    }
    CRIS_CHECK_ARG(global_stats || nparts <= 8 * 64, "partial list longer than one finalize launch covers");
    if (!global_stats && nparts > BN_WIDE_MIN)
        hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3(cris_cdiv(C, 16)), dim3(1024), 0, (hipStream_t)stream, psum, pm2, nparts, rows_per_part,
                           count_local, count, gamma, beta, running_mean, running_var, momentum, eps, C, scale, shift, mean, invstd,
                           merged, global_stats, link);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3(cris_cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, psum, pm2, nparts, rows_per_part,
                           count_local, count, gamma, beta, running_mean, running_var, momentum, eps, C, scale, shift, mean, invstd,
                           merged, global_stats, link);
    CRIS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cris_bn_finalize(const float* psum, const float* pm2, int nparts, int rows_per_part, float count_local,
                                float count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float momentum, float eps, int C, float* scale, float* shift, float* mean, float* invstd,
                                float* merged, const float* global_stats, void* stream) {
    return bn_finalize_launch(psum, pm2, nparts, rows_per_part, count_local, count, gamma, beta, running_mean, running_var, momentum, eps,
                              C, scale, shift, mean, invstd, merged, global_stats, cris_no_link(), stream);
}

static int p2p_link_check(const cris_p2p_link& l, int n, const char* fn) {
    if (l.world <= 0) return 0;
    if (!l.boxes || l.world > 64 || l.rank < 0 || l.rank >= l.world || l.slot < 0 || l.slot >= l.slots || n > l.max_floats) {
        cris_set_error("%s: bad mailbox link (world %d rank %d slot %d / %d, %d values for a capacity of %d)", fn, l.world, l.rank, l.slot,
                       l.slots, n, l.max_floats);
        return -1;
    }
    return 0;
}

extern "C" int cris_bn_finalize_sync(const float* psum, const float* pm2, int nparts, int rows_per_part, float count_local, float count,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                     float eps, int C, float* scale, float* shift, float* mean, float* invstd, const cris_p2p_link* link,
                                     void* stream) {
    CRIS_CHECK_ARG(link && psum && pm2, "null argument");
    CRIS_CHECK_ARG(link->world <= 0 || (running_mean && running_var), "the exchange takes its moments about the running mean");
    if (p2p_link_check(*link, 2 * C, __func__)) return -1;
    return bn_finalize_launch(psum, pm2, nparts, rows_per_part, count_local, count, gamma, beta, running_mean, running_var, momentum, eps,
                              C, scale, shift, mean, invstd, nullptr, nullptr, *link, stream);
}

// SyncBN in ONE exchange: every rank shifts its local (sum, M2 about the local mean) to moments about a reference c that is
// identical on all ranks (the running mean before this step's update):  S1 = sum - n c,  S2 = M2 + n (mean_l - c)^2.
// [S1 | S2] is all-reduced as a single 2C message; the sums convert back to (sum, M2 about the GLOBAL mean):
//   mean = c + S1/N,  M2 = S2 - S1^2/N   (c tracks the batch mean, so the subtraction loses nothing that matters).
__global__ void bn_sync_pack_kernel(float* merged, const float* mean_local, const float* ref, float n_local, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s1, s2;
    bn_sync_pack_vals(merged[c], merged[C + c], mean_local[c], ref[c], n_local, s1, s2);
    merged[c] = s1;
    merged[C + c] = s2;
}
__global__ void bn_sync_unpack_kernel(float* merged, const float* ref, float count_global, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float gsum, gm2;
    bn_sync_unpack_vals(merged[c], merged[C + c], ref[c], count_global, gsum, gm2);
    merged[c] = gsum;                                          // global sum
    merged[C + c] = gm2;                                       // M2 about the global mean
}
extern "C" int cris_bn_sync_pack(float* merged, const float* mean_local, const float* ref, float n_local, int C, void* stream) {
    CRIS_CHECK_ARG(merged && mean_local && ref && C > 0, "bad args");
    hipLaunchKernelGGL(bn_sync_pack_kernel, dim3(cris_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, merged, mean_local, ref, n_local, C);
    CRIS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cris_bn_sync_unpack(float* merged, const float* ref, float count_global, int C, void* stream) {
    CRIS_CHECK_ARG(merged && ref && C > 0 && count_global > 0.f, "bad args");
    hipLaunchKernelGGL(bn_sync_unpack_kernel, dim3(cris_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, merged, ref, count_global, C);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// per-row-block column statistics of a bf16 matrix in the same partial format as the GEMM epilogue
__global__ void colstats_kernel(const bf16_t* x, int ldx, int coff, int M, int C, int rows_per_part, float* psum, float* pm2) {
    const int cv = blockIdx.x * blockDim.x + threadIdx.x;
    const int part = blockIdx.y;
    if (cv >= (C >> 3)) return;
    const int r0 = part * rows_per_part, r1 = min(M, r0 + rows_per_part);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
    for (int m = r0; m < r1; ++m) {
        unpack8(*reinterpret_cast<const uint4*>(x + (size_t)m * ldx + coff + cv * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
    const float inv = 1.f / (float)(r1 - r0);
    for (int m = r0; m < r1; ++m) {
        unpack8(*reinterpret_cast<const uint4*>(x + (size_t)m * ldx + coff + cv * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[j] - s[j] * inv;
            q[j] += d * d;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        psum[(size_t)part * C + cv * 8 + j] = s[j];
        pm2[(size_t)part * C + cv * 8 + j] = q[j];
    }
}
extern "C" int cris_colstats_bf16(const cris_bf16* x, int ldx, int coff, int M, int C, int rows_per_part, float* psum, float* pm2,
                                  void* stream) {
    CRIS_CHECK_ARG(x && psum && pm2 && M > 0 && !(C & 7) && !(ldx & 7) && !(coff & 7) && rows_per_part > 0, "bad args");
    dim3 grid(cris_cdiv(C / 8, 64), cris_cdiv(M, rows_per_part));
    hipLaunchKernelGGL(colstats_kernel, grid, dim3(64), 0, (hipStream_t)stream, x, ldx, coff, M, C, rows_per_part, psum, pm2);
    CRIS_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                               int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

extern "C" int cris_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, int C, float* scale, float* shift, void* stream) {
    CRIS_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bad args");
    hipLaunchKernelGGL(bn_eval_kernel, dim3(cris_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BN apply (+ second branch, identity, ReLU, multiplier, 2x2 average pool, output statistics)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8f(const float* p, float* f) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void load8bf(const bf16_t* p, float* f) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    unpack8(v, f);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const cris_bn_apply_params p) {
    const int CV = p.C >> 3;
    const int OH = p.pool ? p.H / 2 : p.H, OW = p.pool ? p.W / 2 : p.W;
    const long total = (long)p.Bn * OH * OW * CV;
    const bool small = total < (1L << 24);           // (reciprocal index arithmetic: common.h cris_div24)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int mo = small ? cris_div24((int)idx, CV, true) : (int)(idx / CV);
        const int cv = (int)(idx - (long)mo * CV);
        const int c0 = cv * 8;
        float sc[8], sh[8], o[8];
        load8f(p.scale + c0, sc);
        load8f(p.shift + c0, sh);
        if (!p.pool) {
            float y[8];
            load8bf(p.y + (size_t)mo * p.ldy + p.y_coff + c0, y);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = y[j] * sc[j] + sh[j];
            if (p.y2) {
                float y2[8], s2[8], h2[8];
                load8bf(p.y2 + (size_t)mo * p.ldy2 + p.y2_coff + c0, y2);
                load8f(p.scale2 + c0, s2);
                load8f(p.shift2 + c0, h2);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += y2[j] * s2[j] + h2[j];
            }
            if (p.ident) {
                float id[8];
                load8bf(p.ident + (size_t)mo * p.ldi + p.i_coff + c0, id);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += id[j];
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.f);
            }
            if (p.mul) {
                const int b = mo / (p.H * p.W);
                float mu[8];
                load8f(p.mul + (size_t)b * p.C + c0, mu);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] *= mu[j];
            }
        } else {
            const int b = cris_div24(mo, OH * OW, small);
            const int r = mo - b * OH * OW;
            const int oh = cris_div24(r, OW, small), ow = r - oh * OW;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t m = ((size_t)b * p.H + (oh * 2 + dy)) * p.W + (ow * 2 + dx);
                    float y[8];
                    load8bf(p.y + m * p.ldy + p.y_coff + c0, y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float a = y[j] * sc[j] + sh[j];
                        if (p.relu) a = fmaxf(a, 0.f);
                        o[j] += a;
                    }
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= 0.25f;
        }
        cris_st16(p.z + (size_t)mo * p.ldz + p.z_coff + c0, pack8(o));
    }
}

// Fast path of the apply kernel for the launches of the real networks (no pooling, no multiplier, C/8 a power of two <= 256):
// a thread keeps ONE 8-channel vector for its whole life, so scale / shift (and the second branch's) are loaded once instead
// of once per row, there is no index division in the loop, the configuration is a template parameter (one basic block per
// row) and U rows are in flight per thread, all of their 16-byte loads issued before the first use.  The generic kernel
// spent 7 coefficient vectors (224 B from L1) and two 64-bit divisions per 16 bytes of activation: 3.4 TB/s on the largest
// tensors, where the Adam kernel streams at 5.7.
template <bool Y2, bool IDENT>
__global__ __launch_bounds__(256) void bn_apply_fast_kernel(const cris_bn_apply_params p, int cv_shift) {
    constexpr int U = (Y2 && IDENT) ? 2 : 4;
    const int CV = 1 << cv_shift;
    const int c0 = ((int)threadIdx.x & (CV - 1)) * 8;
    const int rows_per_pass = 256 >> cv_shift;
    const int M = p.Bn * p.H * p.W;
    float sc[8], sh[8], s2[8], h2[8];
    load8f(p.scale + c0, sc);
    load8f(p.shift + c0, sh);
    if (Y2) {
        load8f(p.scale2 + c0, s2);
        load8f(p.shift2 + c0, h2);
    }
    const bf16_t* yb = p.y + p.y_coff + c0;
    const bf16_t* y2b = Y2 ? p.y2 + p.y2_coff + c0 : nullptr;
    const bf16_t* ib = IDENT ? p.ident + p.i_coff + c0 : nullptr;
    bf16_t* zb = p.z + p.z_coff + c0;
    const int step = (int)gridDim.x * rows_per_pass;
    for (int m = (int)blockIdx.x * rows_per_pass + ((int)threadIdx.x >> cv_shift); m < M; m += step * U) {
        uint4 ry[U], ry2[U], ri[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int mc = min(m + u * step, M - 1);          // rows past the end re-read the last row (not stored)
            ry[u] = *reinterpret_cast<const uint4*>(yb + (size_t)mc * p.ldy);
            if (Y2) ry2[u] = *reinterpret_cast<const uint4*>(y2b + (size_t)mc * p.ldy2);
            if (IDENT) ri[u] = *reinterpret_cast<const uint4*>(ib + (size_t)mc * p.ldi);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int mu = m + u * step;
            if (mu >= M) break;
            float y[8], o[8];
            unpack8(ry[u], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = y[j] * sc[j] + sh[j];
            if (Y2) {
                float y2[8];
                unpack8(ry2[u], y2);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += y2[j] * s2[j] + h2[j];
            }
            if (IDENT) {
                float id[8];
                unpack8(ri[u], id);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += id[j];
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.f);
            }
            cris_st16(zb + (size_t)mu * p.ldz, pack8(o));
        }
    }
}
// blocks for `total` 8-channel vectors at U rows per thread and pass: U vectors per thread once the tensor is large enough to
// fill the chip that way (at most 2048 blocks, 8 per CU), one vector per thread (up to 1024 blocks) for the small ones
static int bn_fast_grid(long total, int U) {
    const int few = cris_grid_1d(total, 256 * U, 2048), many = cris_grid_1d(total, 256, 1024);
    return few > many ? few : many;
}
// log2(C/8) when C/8 is a power of two <= 256 (the fast kernels' thread mapping), else -1
static int bn_cv_shift(int C) {
    const int CV = C >> 3;
    if (CV < 1 || CV > 256 || (CV & (CV - 1))) return -1;
    int s = 0;
    while ((1 << s) < CV) ++s;
    return s;
}

extern "C" int cris_bn_apply(const cris_bn_apply_params* pp, void* stream) {
    const cris_bn_apply_params& p = *pp;
    CRIS_CHECK_ARG(p.y && p.scale && p.shift && p.z, "null operand");
    CRIS_CHECK_ARG((p.C & 7) == 0 && (p.ldy & 7) == 0 && (p.y_coff & 7) == 0 && (p.ldz & 7) == 0 && (p.z_coff & 7) == 0,
                   "channels / ld / offsets must be multiples of 8");
    CRIS_CHECK_ARG(!p.pool || (!(p.H & 1) && !(p.W & 1) && !p.y2 && !p.ident && !p.mul), "pool needs even H,W and a plain BN");
    CRIS_CHECK_ARG(!p.y2 || ((p.ldy2 & 7) == 0 && (p.y2_coff & 7) == 0 && p.scale2 && p.shift2), "branch 2");
    CRIS_CHECK_ARG(!p.ident || ((p.ldi & 7) == 0 && (p.i_coff & 7) == 0), "identity ld/offset");
    const long rows = (long)p.Bn * (p.pool ? (p.H / 2) * (p.W / 2) : p.H * p.W);
    const long total = rows * (p.C >> 3);
    static const int use_fast = cris_env_int("CRIS_BN_APPLY_FAST", 1);
    const int cvs = bn_cv_shift(p.C);
    if (use_fast && cvs >= 0 && !p.pool && !p.mul) {
        const int grid = bn_fast_grid(total, (p.y2 && p.ident) ? 2 : 4);
        if (p.y2 && p.ident) hipLaunchKernelGGL((bn_apply_fast_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, cvs);
        else if (p.y2) hipLaunchKernelGGL((bn_apply_fast_kernel<true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, cvs);
        else if (p.ident) hipLaunchKernelGGL((bn_apply_fast_kernel<false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, cvs);
        else hipLaunchKernelGGL((bn_apply_fast_kernel<false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, cvs);
        CRIS_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(cris_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BN backward: reduce (sum g, sum g*xhat [, branch 2]) then apply
// ------------------------------------------------------------------------------------------------
// gradient entering the BN output at full-resolution row m, channels c0..c0+7; also xhat (and xhat2)
__device__ __forceinline__ void bn_bwd_point(const cris_bn_bwd_params& p, int m, int c0, float* g, float* xh, float* xh2) {
    float y[8], mean[8], inv[8];
    load8bf(p.y + (size_t)m * p.ldy + p.y_coff + c0, y);
    load8f(p.mean + c0, mean);
    load8f(p.invstd + c0, inv);
#pragma unroll
    for (int j = 0; j < 8; ++j) xh[j] = (y[j] - mean[j]) * inv[j];
    if (p.y2) {
        float y2[8], m2[8], i2[8];
        load8bf(p.y2 + (size_t)m * p.ldy2 + p.y2_coff + c0, y2);
        load8f(p.mean2 + c0, m2);
        load8f(p.invstd2 + c0, i2);
#pragma unroll
        for (int j = 0; j < 8; ++j) xh2[j] = (y2[j] - m2[j]) * i2[j];
    }
    const int HW = p.H * p.W;
    int mo = m;
    float gscale = 1.f;
    const bool small = (long)p.Bn * HW < (1L << 24);       // (pixel indices fit 24 bits: reciprocal divisions, common.h)
    if (p.pool) {
        const int b = cris_div24(m, HW, small);
        const int r = m - b * HW;
        const int h = cris_div24(r, p.W, small), w = r - h * p.W;
        mo = (b * (p.H / 2) + (h >> 1)) * (p.W / 2) + (w >> 1);
        gscale = 0.25f;
    }
    float dz[8];
    load8bf(p.dz + (size_t)mo * p.lddz + p.dz_coff + c0, dz);
    // ReLU mask
    bool pos[8];
    if (!p.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pos[j] = true;
    } else if (!p.pool && (p.y2 || p.z)) {
        float z[8];
        load8bf(p.z + (size_t)m * p.ldz + p.z_coff + c0, z);
#pragma unroll
        for (int j = 0; j < 8; ++j) pos[j] = z[j] > 0.f;
    } else {
        float sc[8], sh[8];
        load8f(p.scale + c0, sc);
        load8f(p.shift + c0, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) pos[j] = (y[j] * sc[j] + sh[j]) > 0.f;
    }
    if (p.mul) {
        const int b = cris_div24(m, HW, small);
        float mu[8];
        load8f(p.mul + (size_t)b * p.C + c0, mu);
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] *= mu[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = pos[j] ? dz[j] * gscale : 0.f;
}

// Geometry shared by the reduce and apply kernels: a block owns ONE chunk of `chv` 8-channel vectors (up to 64 channels) and a
// range of rows; its 256 threads are chv vector lanes x 256/chv row lanes.  The reduce grid is chunks x row blocks with at most
// 64 row blocks, so a channel's partial sums form a column of <= 64 entries that one small launch adds up in order
// (deterministic, no atomics).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const cris_bn_bwd_params p, int rows_per_block, int chv, int chunks) {
    // Each thread owns one 8-channel vector and a strided subset of the block's rows (register accumulation); the RS row
    // lanes of a vector are then combined through a plain LDS table + a column sum in lane order.
    __shared__ float spart[3][256 * 8];            // [sum][thread-major: rsub * cvn*8 + cv_local*8 + j]
    const int CV = p.C >> 3;
    const int M = p.Bn * p.H * p.W;
    const int chunk = blockIdx.x % chunks, rb = blockIdx.x / chunks;
    const int r0 = rb * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    {
        const int cvb = chunk * chv;
        const int cvn = min(chv, CV - cvb);
        const int RS = 256 / cvn;
        const int cvl = (int)threadIdx.x % cvn;
        const int cv = cvb + cvl;
        const int rsub = threadIdx.x / cvn;
        float a0[8], a1[8], a3[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a0[j] = a1[j] = a3[j] = 0.f;
        if (rsub < RS) {
            for (int m = r0 + rsub; m < r1; m += RS) {
                float g[8], xh[8], xh2[8];
                bn_bwd_point(p, m, cv * 8, g, xh, xh2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a0[j] += g[j];
                    a1[j] += g[j] * xh[j];
                    if (p.y2) a3[j] += g[j] * xh2[j];
                }
            }
        }
        if (rsub < RS) {
            const int base = (rsub * cvn + cvl) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                spart[0][base + j] = a0[j];
                spart[1][base + j] = a1[j];
                spart[2][base + j] = a3[j];
            }
        }
        __syncthreads();
        const int ncol = cvn * 8;                  // columns of this chunk
        for (int c = threadIdx.x; c < ncol; c += 256) {
            float s0 = 0.f, s1 = 0.f, s3 = 0.f;
            for (int r = 0; r < RS; ++r) {
                s0 += spart[0][r * ncol + c];
                s1 += spart[1][r * ncol + c];
                s3 += spart[2][r * ncol + c];
            }
            const int col = cvb * 8 + c;
            float* part = p.part + (size_t)rb * (p.y2 ? 4 : 2) * p.C;      // this row block's row of the partials table
            part[col] = s0;
            part[p.C + col] = s1;
            if (p.y2) {
                part[2 * p.C + col] = s0;
                part[3 * p.C + col] = s3;
            }
        }
    }
}

// Specialised reduce for the launches without a forward multiplier (all but one BatchNorm of the network).  Same block /
// thread mapping and the same fixed-order LDS combine as the generic kernel above, but (a) the per-channel constants are
// loaded ONCE per thread instead of once per row, (b) the configuration flags are template parameters, so the row loop is
// one basic block, and (c) U rows are in flight per thread: all of their 16-byte loads are issued before the first use
// (rows past the block's range re-read its last row with weight 0 - an unconditional load, not a branch).
//   MASK 0: no ReLU; 1: ReLU mask from the stored forward output z; 2: ReLU mask recomputed from scale*y + shift.
template <int MASK, bool Y2, bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_fast_kernel(const cris_bn_bwd_params p, int rows_per_block, int chv, int chunks) {
    constexpr int U = Y2 ? 2 : 4;
    __shared__ float spart[3][256 * 8];
    const int CV = p.C >> 3;
    const int M = p.Bn * p.H * p.W;
    const int chunk = blockIdx.x % chunks, rb = blockIdx.x / chunks;
    const int r0 = rb * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const int HW = p.H * p.W, W2 = p.W >> 1, HW2 = (p.H >> 1) * W2;
    const bool small_px = M < (1 << 24);
    const float gscale = POOL ? 0.25f : 1.f;
    {
        const int cvb = chunk * chv;
        const int cvn = min(chv, CV - cvb);
        const int RS = 256 / cvn;
        const int cvl = (int)threadIdx.x % cvn;
        const int c0 = (cvb + cvl) * 8;
        const int rsub = threadIdx.x / cvn;
        float a0[8], a1[8], a3[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a0[j] = a1[j] = a3[j] = 0.f;
        if (rsub < RS) {
            float mean[8], inv[8], sc[8], sh[8], mean2[8], inv2[8];
            load8f(p.mean + c0, mean);
            load8f(p.invstd + c0, inv);
            if (MASK == 2) {
                load8f(p.scale + c0, sc);
                load8f(p.shift + c0, sh);
            }
            if (Y2) {
                load8f(p.mean2 + c0, mean2);
                load8f(p.invstd2 + c0, inv2);
            }
            const bf16_t* yb = p.y + p.y_coff + c0;
            const bf16_t* dzb = p.dz + p.dz_coff + c0;
            const bf16_t* zb = MASK == 1 ? p.z + p.z_coff + c0 : nullptr;
            const bf16_t* y2b = Y2 ? p.y2 + p.y2_coff + c0 : nullptr;
            for (int m = r0 + rsub; m < r1; m += RS * U) {
                uint4 ry[U], rdz[U], rz[U], ry2[U];
                float wgt[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int mu = m + u * RS;
                    wgt[u] = mu < r1 ? gscale : 0.f;
                    const int mc = min(mu, r1 - 1);
                    int mo = mc;
                    if (POOL) {
                        const int b = cris_div24(mc, HW, small_px);
                        const int r = mc - b * HW;
                        const int h = cris_div24(r, p.W, small_px), w = r - h * p.W;
                        mo = b * HW2 + (h >> 1) * W2 + (w >> 1);
                    }
                    ry[u] = *reinterpret_cast<const uint4*>(yb + (size_t)mc * p.ldy);
                    rdz[u] = *reinterpret_cast<const uint4*>(dzb + (size_t)mo * p.lddz);
                    if (MASK == 1) rz[u] = *reinterpret_cast<const uint4*>(zb + (size_t)mc * p.ldz);
                    if (Y2) ry2[u] = *reinterpret_cast<const uint4*>(y2b + (size_t)mc * p.ldy2);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float y[8], dz[8], z[8], y2[8];
                    unpack8(ry[u], y);
                    unpack8(rdz[u], dz);
                    if (MASK == 1) unpack8(rz[u], z);
                    if (Y2) unpack8(ry2[u], y2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        bool pos = true;
                        if (MASK == 1) pos = z[j] > 0.f;
                        if (MASK == 2) pos = (y[j] * sc[j] + sh[j]) > 0.f;
                        const float g = pos ? dz[j] * wgt[u] : 0.f;
                        a0[j] += g;
                        a1[j] += g * ((y[j] - mean[j]) * inv[j]);
                        if (Y2) a3[j] += g * ((y2[j] - mean2[j]) * inv2[j]);
                    }
                }
            }
        }
        if (rsub < RS) {
            const int base = (rsub * cvn + cvl) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                spart[0][base + j] = a0[j];
                spart[1][base + j] = a1[j];
                spart[2][base + j] = a3[j];
            }
        }
        __syncthreads();
        const int ncol = cvn * 8;
        for (int c = threadIdx.x; c < ncol; c += 256) {
            float s0 = 0.f, s1 = 0.f, s3 = 0.f;
            for (int r = 0; r < RS; ++r) {
                s0 += spart[0][r * ncol + c];
                s1 += spart[1][r * ncol + c];
                if (Y2) s3 += spart[2][r * ncol + c];
            }
            const int col = cvb * 8 + c;
            float* part = p.part + (size_t)rb * (Y2 ? 4 : 2) * p.C;
            part[col] = s0;
            part[p.C + col] = s1;
            if (Y2) {
                part[2 * p.C + col] = s0;
                part[3 * p.C + col] = s3;
            }
        }
    }
}

typedef void (*bn_bwd_reduce_fn)(const cris_bn_bwd_params, int, int, int);
// [MASK][Y2][POOL]; combinations the path never produces stay on the generic kernel
static bn_bwd_reduce_fn bn_bwd_reduce_fast_table(int mask, bool y2, bool pool) {
    if (pool) {
        if (y2 || mask == 1) return nullptr;
        return mask == 2 ? bn_bwd_reduce_fast_kernel<2, false, true> : bn_bwd_reduce_fast_kernel<0, false, true>;
    }
    if (y2) {
        if (mask == 2) return nullptr;
        return mask == 1 ? bn_bwd_reduce_fast_kernel<1, true, false> : bn_bwd_reduce_fast_kernel<0, true, false>;
    }
    if (mask == 1) return bn_bwd_reduce_fast_kernel<1, false, false>;
    if (mask == 2) return bn_bwd_reduce_fast_kernel<2, false, false>;
    return bn_bwd_reduce_fast_kernel<0, false, false>;
}

// gradient of the per-sample multiplier (FPN: f5 = relu(bn(y)) * state, model/layers.py:289): dmul[b][c] = sum over the pixels of
// sample b of dz * relu(bn(y)).  Block = (sample, 64 channels): 8 channel vectors x 32 pixel lanes, each lane adds its pixels
// in order, the lanes are added in lane order through LDS: deterministic.
__global__ __launch_bounds__(256) void bn_dmul_kernel(const cris_bn_bwd_params p) {
    __shared__ float sh[32][8][8 + 1];
    const int CV = p.C >> 3, HW = p.H * p.W;
    const int chunks = (CV + 7) / 8;
    const int b = blockIdx.x / chunks, cvb = (blockIdx.x - b * chunks) * 8;
    const int cvl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int cv = cvb + cvl;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cv < CV) {
        const int c0 = cv * 8;
        float sc[8], shf[8];
        load8f(p.scale + c0, sc);
        load8f(p.shift + c0, shf);
        for (int r = pl; r < HW; r += 32) {
            const size_t m = (size_t)b * HW + r;
            float y[8], dz[8];
            load8bf(p.y + m * p.ldy + p.y_coff + c0, y);
            load8bf(p.dz + m * p.lddz + p.dz_coff + c0, dz);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += dz[j] * fmaxf(y[j] * sc[j] + shf[j], 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[pl][cvl][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x >> 3, j = threadIdx.x & 7;
        if (cvb + l < CV) {
            float a = 0.f;
            for (int q = 0; q < 32; ++q) a += sh[q][l][j];
            p.dmul[(size_t)b * p.C + (cvb + l) * 8 + j] = a;
        }
    }
}

// chunk width (8-channel vectors per block, a power of two <= 8), number of chunks, rows per row block, row blocks (<= 64)
struct bn_bwd_geom {
    int chv, chunks, rpb, rbs;
};
static bn_bwd_geom bn_bwd_geometry(int M, int C) {
    static const int max_blocks = cris_env_int("CRIS_BN_RED_BLOCKS", 512);
    static const int min_rows = cris_env_int("CRIS_BN_RED_ROWS", 32);
    bn_bwd_geom g;
    const int CV = C >> 3;
    g.chv = 1;
    while (g.chv < 8 && g.chv * 2 <= CV / 4) g.chv *= 2;       // ~C/4 channels per chunk, at most 64
    g.chunks = cris_cdiv(CV, g.chv);
    int rbs = max_blocks / g.chunks;
    if (rbs > 64) rbs = 64;
    if (rbs < 1) rbs = 1;
    g.rpb = cris_cdiv(M, rbs);
    if (g.rpb < min_rows) g.rpb = min_rows;
    g.rbs = cris_cdiv(M, g.rpb);
    return g;
}
extern "C" long cris_bn_bwd_ws_floats(const cris_bn_bwd_params* p) {
    return (long)bn_bwd_geometry(p->Bn * p->H * p->W, p->C).rbs * (p->y2 ? 4 : 2) * p->C;
}

static int bn_bwd_reduce_launch(const cris_bn_bwd_params* pp, float* local_sums, const cris_p2p_link& link, void* stream) {
    const cris_bn_bwd_params& p = *pp;
    CRIS_CHECK_ARG(p.dz && p.y && p.mean && p.invstd && p.sums && p.part, "null operand");
    CRIS_CHECK_ARG((p.C & 7) == 0 && p.C <= 8192, "C");
    CRIS_CHECK_ARG(!p.relu || p.pool || p.z || (p.scale && p.shift), "relu mask source");
    CRIS_CHECK_ARG(!p.pool || (p.scale && p.shift && !p.y2 && !p.mul), "pool backward needs scale/shift, plain BN");
    CRIS_CHECK_ARG(!p.mul || !p.dmul || (p.relu && !p.pool && !p.y2 && p.scale && p.shift && (p.lddz & 7) == 0 && (p.dz_coff & 7) == 0),
                   "multiplier gradient: plain BN + ReLU");
    const int M = p.Bn * p.H * p.W;
    const bn_bwd_geom g = bn_bwd_geometry(M, p.C);
    static const int use_fast = cris_env_int("CRIS_BN_RED_FAST", 1);
    bn_bwd_reduce_fn fast = nullptr;
    if (use_fast && !p.mul && (p.ldy & 7) == 0 && (p.y_coff & 7) == 0 && (p.lddz & 7) == 0 && (p.dz_coff & 7) == 0 &&
        (!p.y2 || ((p.ldy2 & 7) == 0 && (p.y2_coff & 7) == 0 && p.z))) {
        const int mask = !p.relu ? 0 : (!p.pool && (p.y2 || p.z)) ? 1 : 2;
        if (mask != 1 || ((p.ldz & 7) == 0 && (p.z_coff & 7) == 0)) fast = bn_bwd_reduce_fast_table(mask, p.y2 != nullptr, p.pool != 0);
    }
    hipLaunchKernelGGL(fast ? fast : bn_bwd_reduce_kernel, dim3(g.chunks * g.rbs), dim3(256), 0, (hipStream_t)stream, p, g.rpb, g.chv, g.chunks);
    CRIS_LAUNCH_CHECK();
    // the row blocks' partial rows, summed in block order (deterministic) into the [2C] ([4C]) sums (+=)
    const int ncol = (p.y2 ? 4 : 2) * p.C;
    if (p2p_link_check(link, ncol, __func__)) return -1;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(cris_cdiv(ncol, 16)), dim3(256), 0, (hipStream_t)stream, p.part, g.rbs, ncol, p.sums,
                       local_sums, link);
    CRIS_LAUNCH_CHECK();
    if (p.mul && p.dmul) {
        hipLaunchKernelGGL(bn_dmul_kernel, dim3(p.Bn * cris_cdiv(p.C >> 3, 8)), dim3(256), 0, (hipStream_t)stream, p);
        CRIS_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int cris_bn_bwd_reduce(const cris_bn_bwd_params* pp, void* stream) { return bn_bwd_reduce_launch(pp, nullptr, cris_no_link(), stream); }

static int bn_bwd_sum_launch(const cris_bn_bwd_params* pp, int nparts, float* local_sums, const cris_p2p_link& link, void* stream) {
    const cris_bn_bwd_params& p = *pp;
    CRIS_CHECK_ARG(p.sums && p.part && nparts > 0 && (p.C & 7) == 0, "bad args");
    CRIS_CHECK_ARG(!p.y2 && !p.mul, "partial rows from a GEMM epilogue exist for the plain conv -> BatchNorm -> ReLU chain only");
    const int ncol = 2 * p.C;
    if (p2p_link_check(link, ncol, __func__)) return -1;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(cris_cdiv(ncol, 16)), dim3(256), 0, (hipStream_t)stream, p.part, nparts, ncol, p.sums,
                       local_sums, link);
    CRIS_LAUNCH_CHECK();
    return 0;
}
extern "C" int cris_bn_bwd_sum(const cris_bn_bwd_params* pp, int nparts, void* stream) {
    CRIS_CHECK_ARG(pp, "null argument");
    return bn_bwd_sum_launch(pp, nparts, nullptr, cris_no_link(), stream);
}
extern "C" int cris_bn_bwd_sum_sync(const cris_bn_bwd_params* pp, int nparts, float* local_sums, const cris_p2p_link* link, void* stream) {
    CRIS_CHECK_ARG(pp && link, "null argument");
    return bn_bwd_sum_launch(pp, nparts, local_sums, *link, stream);
}

extern "C" int cris_bn_bwd_reduce_sync(const cris_bn_bwd_params* pp, float* local_sums, const cris_p2p_link* link, void* stream) {
    CRIS_CHECK_ARG(pp && link, "null argument");
    return bn_bwd_reduce_launch(pp, local_sums, *link, stream);
}

// (a channel-chunked apply geometry, and letting its blocks add up the reduce kernel's partial rows themselves to save the
// summation launch, were both measured slower: 13.0 against 11.3 us per launch, 16.0 against 15.2 ms per step)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const cris_bn_bwd_params p) {
    const int CV = p.C >> 3;
    const long total = (long)p.Bn * p.H * p.W * CV;
    const bool small = total < (1L << 24);
    const float invc = 1.0f / p.count;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = small ? cris_div24((int)idx, CV, true) : (int)(idx / CV);
        const int cv = (int)(idx - (long)m * CV);
        const int c0 = cv * 8;
        float g[8], xh[8], xh2[8];
        bn_bwd_point(p, m, c0, g, xh, xh2);
        float s0[8], s1[8], sc[8], o[8];
        load8f(p.sums + c0, s0);
        load8f(p.sums + p.C + c0, s1);
        load8f(p.scale + c0, sc);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = sc[j] * (g[j] - s0[j] * invc - xh[j] * s1[j] * invc);
        cris_st16(p.dy + (size_t)m * p.lddy + p.dy_coff + c0, pack8(o));
        if (p.y2 && p.dy2) {
            float s3[8], sc2[8];
            load8f(p.sums + 3 * p.C + c0, s3);
            load8f(p.scale2 + c0, sc2);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = sc2[j] * (g[j] - s0[j] * invc - xh2[j] * s3[j] * invc);
            cris_st16(p.dy2 + (size_t)m * p.lddy2 + p.dy2_coff + c0, pack8(o));
        }
        if (p.dident) {
            bf16_t* dst = p.dident + (size_t)m * p.lddi + p.di_coff + c0;
            if (p.dident_accum) {
                float old[8];
                load8bf(dst, old);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] += old[j];
            }
            cris_st16(dst, pack8(g));
        }
    }
}

// Fast path of the backward apply (same idea as bn_apply_fast_kernel: one 8-channel vector per thread, constants loaded once,
// flags as template parameters, U rows in flight).  MASK as in bn_bwd_reduce_fast_kernel.
template <int MASK, bool Y2>
__global__ __launch_bounds__(256) void bn_bwd_apply_fast_kernel(const cris_bn_bwd_params p, int cv_shift) {
    constexpr int U = Y2 ? 2 : 4;
    const int CV = 1 << cv_shift;
    const int c0 = ((int)threadIdx.x & (CV - 1)) * 8;
    const int rows_per_pass = 256 >> cv_shift;
    const int M = p.Bn * p.H * p.W;
    const float invc = 1.0f / p.count;
    float mean[8], inv[8], sc[8], sh[8], a0[8], s1[8], mean2[8], inv2[8], s3[8], sc2[8];
    load8f(p.mean + c0, mean);
    load8f(p.invstd + c0, inv);
    load8f(p.scale + c0, sc);
    if (MASK == 2) load8f(p.shift + c0, sh);
    load8f(p.sums + c0, a0);
    load8f(p.sums + p.C + c0, s1);
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] *= invc;
    const bool want2 = Y2 && p.dy2;
    if (Y2) {
        load8f(p.mean2 + c0, mean2);
        load8f(p.invstd2 + c0, inv2);
        load8f(p.sums + 3 * p.C + c0, s3);
        load8f(p.scale2 + c0, sc2);
    }
    const bf16_t* yb = p.y + p.y_coff + c0;
    const bf16_t* dzb = p.dz + p.dz_coff + c0;
    const bf16_t* zb = MASK == 1 ? p.z + p.z_coff + c0 : nullptr;
    const bf16_t* y2b = Y2 ? p.y2 + p.y2_coff + c0 : nullptr;
    bf16_t* dyb = p.dy + p.dy_coff + c0;
    bf16_t* dy2b = want2 ? p.dy2 + p.dy2_coff + c0 : nullptr;
    bf16_t* dib = p.dident ? p.dident + p.di_coff + c0 : nullptr;
    const bool di_acc = p.dident && p.dident_accum;
    const int step = (int)gridDim.x * rows_per_pass;
    for (int m = (int)blockIdx.x * rows_per_pass + ((int)threadIdx.x >> cv_shift); m < M; m += step * U) {
        uint4 ry[U], rdz[U], rz[U], ry2[U], rold[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int mc = min(m + u * step, M - 1);
            ry[u] = *reinterpret_cast<const uint4*>(yb + (size_t)mc * p.ldy);
            rdz[u] = *reinterpret_cast<const uint4*>(dzb + (size_t)mc * p.lddz);
            if (MASK == 1) rz[u] = *reinterpret_cast<const uint4*>(zb + (size_t)mc * p.ldz);
            if (Y2) ry2[u] = *reinterpret_cast<const uint4*>(y2b + (size_t)mc * p.ldy2);
            if (di_acc) rold[u] = *reinterpret_cast<const uint4*>(dib + (size_t)mc * p.lddi);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int mu = m + u * step;
            if (mu >= M) break;
            float y[8], dz[8], z[8], g[8], o[8];
            unpack8(ry[u], y);
            unpack8(rdz[u], dz);
            if (MASK == 1) unpack8(rz[u], z);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bool pos = true;
                if (MASK == 1) pos = z[j] > 0.f;
                if (MASK == 2) pos = (y[j] * sc[j] + sh[j]) > 0.f;
                g[j] = pos ? dz[j] : 0.f;
                const float xh = (y[j] - mean[j]) * inv[j];
                o[j] = sc[j] * (g[j] - a0[j] - xh * s1[j] * invc);
            }
            cris_st16(dyb + (size_t)mu * p.lddy, pack8(o));
            if (want2) {
                float y2[8];
                unpack8(ry2[u], y2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh2 = (y2[j] - mean2[j]) * inv2[j];
                    o[j] = sc2[j] * (g[j] - a0[j] - xh2 * s3[j] * invc);
                }
                cris_st16(dy2b + (size_t)mu * p.lddy2, pack8(o));
            }
            if (dib) {
                if (di_acc) {
                    float old[8];
                    unpack8(rold[u], old);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += old[j];
                }
                cris_st16(dib + (size_t)mu * p.lddi, pack8(g));
            }
        }
    }
}

extern "C" int cris_bn_bwd_apply(const cris_bn_bwd_params* pp, void* stream) {
    const cris_bn_bwd_params& p = *pp;
    CRIS_CHECK_ARG(p.dz && p.y && p.mean && p.invstd && p.sums && p.scale && p.dy, "null operand");
    CRIS_CHECK_ARG((p.C & 7) == 0 && (p.lddy & 7) == 0 && (p.dy_coff & 7) == 0 && p.count > 0.f, "geometry");
    const long total = (long)p.Bn * p.H * p.W * (p.C >> 3);
    static const int use_fast = cris_env_int("CRIS_BN_APPLY_FAST", 1);
    const int cvs = bn_cv_shift(p.C);
    const bool aligned = (p.ldy & 7) == 0 && (p.y_coff & 7) == 0 && (p.lddz & 7) == 0 && (p.dz_coff & 7) == 0 &&
                         (!p.y2 || ((p.ldy2 & 7) == 0 && (p.y2_coff & 7) == 0 && p.z && (!p.dy2 || ((p.lddy2 & 7) == 0 && (p.dy2_coff & 7) == 0)))) &&
                         (!p.dident || ((p.lddi & 7) == 0 && (p.di_coff & 7) == 0));
    if (use_fast && cvs >= 0 && !p.pool && !p.mul && aligned) {
        const int mask = !p.relu ? 0 : (p.y2 || p.z) ? 1 : 2;            // the same rule as bn_bwd_point / cris_bn_bwd_reduce
        if (mask != 1 || ((p.ldz & 7) == 0 && (p.z_coff & 7) == 0)) {
            const int grid = bn_fast_grid(total, p.y2 ? 2 : 4);
            typedef void (*fn_t)(const cris_bn_bwd_params, int);
            static const fn_t tab[2][3] = {{bn_bwd_apply_fast_kernel<0, false>, bn_bwd_apply_fast_kernel<1, false>, bn_bwd_apply_fast_kernel<2, false>},
                                           {bn_bwd_apply_fast_kernel<0, true>, bn_bwd_apply_fast_kernel<1, true>, bn_bwd_apply_fast_kernel<2, true>}};
            hipLaunchKernelGGL(tab[p.y2 ? 1 : 0][mask], dim3(grid), dim3(256), 0, (hipStream_t)stream, p, cvs);
            CRIS_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cris_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, up to 4 x 8 channels per lane (C <= 2048)
// ------------------------------------------------------------------------------------------------
#define LN_MAXV 4

template <bool WANT_MASK, int V = LN_MAXV>
__device__ __forceinline__ void ln_load_row(const void* x, int x_f32, size_t rowoff, int C, int lane, int in_relu,
                                            uint32_t in_thresh, float in_scale, uint32_t in_key, uint32_t row,
                                            float (&v)[V][8], float (&mask_out)[V][8]) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c0 = (lane + 64 * i) * 8;
        if (c0 < C) {
            if (x_f32) load8f(reinterpret_cast<const float*>(x) + rowoff + c0, v[i]);
            else load8bf(reinterpret_cast<const bf16_t*>(x) + rowoff + c0, v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float m = 1.f;
                if (in_relu && !(v[i][j] > 0.f)) m = 0.f;
                if (in_thresh) m = cris_keep(in_key, row * (uint32_t)C + (uint32_t)(c0 + j), in_thresh) ? m * in_scale : 0.f;
                v[i][j] *= m;
                if (WANT_MASK) mask_out[i][j] = m;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = 0.f;
                if (WANT_MASK) mask_out[i][j] = 0.f;
            }
        }
    }
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const cris_ln_fwd_params p) {
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const uint32_t sdev = p.seed_dev ? p.seed_dev[0] : 0u;
    const uint32_t in_key = cris_drop_key(p.in_seed + sdev, p.in_stream);
    const uint32_t out_key = cris_drop_key(p.out_seed + sdev, p.out_stream);
    const float in_scale = p.in_thresh ? 1.f / (1.f - p.in_drop_p) : 1.f;
    const float out_scale = p.out_thresh ? 1.f / (1.f - p.out_drop_p) : 1.f;
    const float invC = 1.f / (float)p.C;
    for (int row = wave_g; row < p.rows; row += nwaves) {
        float v[LN_MAXV][8];
        ln_load_row<false>(p.x, p.x_f32, (size_t)row * p.ldx, p.C, lane, p.in_relu, p.in_thresh, in_scale, in_key, (uint32_t)row,
                           v, v);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c0 = (lane + 64 * i) * 8;
            if (c0 < p.C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[i][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + p.eps);
        if (lane == 0) {
            p.mean[row] = mean;
            p.rstd[row] = rstd;
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c0 = (lane + 64 * i) * 8;
            if (c0 >= p.C) continue;
            float ga[8], be[8], o[8];
            load8f(p.gamma + c0, ga);
            load8f(p.beta + c0, be);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * ga[j] + be[j];
            const size_t oo = (size_t)row * p.C + c0;
            if (p.y) cris_st16(p.y + oo, pack8(o));
            if (p.ypos) {
                float pe[8], t[8];
                load8f(p.pos + (size_t)(row % p.pos_rows) * p.C + c0, pe);
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = o[j] + pe[j];
                cris_st16(p.ypos + oo, pack8(t));
            }
            if (p.out_f32) {
                float r[8];
                if (p.resid) load8f(p.resid + oo, r);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = o[j];
                    if (p.out_thresh) t = cris_keep(out_key, (uint32_t)row * (uint32_t)p.C + (uint32_t)(c0 + j), p.out_thresh) ? t * out_scale : 0.f;
                    r[j] += t;
                }
                *reinterpret_cast<float4*>(p.out_f32 + oo) = make_float4(r[0], r[1], r[2], r[3]);
                *reinterpret_cast<float4*>(p.out_f32 + oo + 4) = make_float4(r[4], r[5], r[6], r[7]);
            }
        }
    }
}

extern "C" int cris_ln_fwd(const cris_ln_fwd_params* pp, void* stream) {
    const cris_ln_fwd_params& p = *pp;
    CRIS_CHECK_ARG(p.x && p.gamma && p.beta && p.mean && p.rstd && p.rows > 0, "null operand");
    CRIS_CHECK_ARG((p.C & 7) == 0 && p.C <= 64 * 8 * LN_MAXV && (p.ldx & 7) == 0, "C must be a multiple of 8, <= 2048");
    CRIS_CHECK_ARG(!p.ypos || (p.pos && p.pos_rows > 0), "ypos needs pos");
    CRIS_CHECK_ARG((long)p.rows * p.C < (1L << 32), "dropout index overflow");
    const int grid = cris_grid_1d(p.rows, 4, 2048);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// V = 8-channel vectors per lane (C <= 512 V).  V = 4 serves every C <= 2048 and is what runs by default; the narrower instantiations
// (V = 1 / 2 for C <= 512 / 1024; CRIS_LN_BWD_V=0 switches them off) keep a quarter / half of the registers - the V = 4 kernel holds
// 238 VGPRs, two waves per SIMD, for rows of which a C = 512 LayerNorm uses one vector.  Measured in round 4 (call r04a): 12.188
// against 12.280 ms per step with the default grid (a grid of 1024 blocks gives the gain back: 12.276); default since.
template <int V>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const cris_ln_bwd_params p) {
    __shared__ float sg[2][64 * 8 * V];        // dgamma / dbeta block accumulators
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    for (int i = threadIdx.x; i < 2 * 64 * 8 * V; i += 256) (&sg[0][0])[i] = 0.f;
    __syncthreads();
    const uint32_t sdev = p.seed_dev ? p.seed_dev[0] : 0u;
    const uint32_t in_key = cris_drop_key(p.in_seed + sdev, p.in_stream);
    const uint32_t out_key = cris_drop_key(p.out_seed + sdev, p.out_stream);
    const float in_scale = p.in_thresh ? 1.f / (1.f - p.in_drop_p) : 1.f;
    const float out_scale = p.out_thresh ? 1.f / (1.f - p.out_drop_p) : 1.f;
    const float invC = 1.f / (float)p.C;
    float dga[V][8], dbe[V][8];
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) dga[i][j] = dbe[i][j] = 0.f;

    for (int row = wave_g; row < p.rows; row += nwaves) {
        float v[V][8], msk[V][8];
        ln_load_row<true, V>(p.x, p.x_f32, (size_t)row * p.ldx, p.C, lane, p.in_relu, p.in_thresh, in_scale, in_key, (uint32_t)row,
                          v, msk);
        const float mean = p.mean[row], rstd = p.rstd[row];
        float a[V][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c0 = (lane + 64 * i) * 8;
            if (c0 < p.C) {
                const size_t oo = (size_t)row * p.C + c0;
                float g[8], t[8], ga[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = 0.f;
                if (p.dy) {
                    load8bf(p.dy + oo, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += t[j];
                }
                if (p.dypos) {
                    load8bf(p.dypos + oo, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += t[j];
                }
                if (p.dout_f32) {
                    load8f(p.dout_f32 + oo, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float d = t[j];
                        if (p.out_thresh) d = cris_keep(out_key, (uint32_t)row * (uint32_t)p.C + (uint32_t)(c0 + j), p.out_thresh) ? d * out_scale : 0.f;
                        g[j] += d;
                    }
                }
                load8f(p.gamma + c0, ga);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (v[i][j] - mean) * rstd;
                    v[i][j] = xh;
                    dga[i][j] += g[j] * xh;
                    dbe[i][j] += g[j];
                    a[i][j] = g[j] * ga[j];
                    s1 += a[i][j];
                    s2 += a[i][j] * xh;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[i][j] = 0.f;
            }
        }
        const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c0 = (lane + 64 * i) * 8;
            if (c0 >= p.C) continue;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (a[i][j] - m1 - v[i][j] * m2) * msk[i][j];
            const size_t xo = (size_t)row * p.ldx + c0;
            if (p.dx_f32) {
                float* d = reinterpret_cast<float*>(p.dx) + xo;
                if (p.dx_accum) {
                    float old[8];
                    load8f(d, old);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += old[j];
                }
                *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(d + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
                cris_st16(reinterpret_cast<bf16_t*>(p.dx) + xo, pack8(o));
            }
        }
    }
    // parameter gradients: registers -> LDS, the 4 waves adding one after the other (fixed order) -> this block's row of the
    // partials table [grid][dgamma C | dbeta C]; cris_sum_tables adds the rows in block order (deterministic, no atomics)
    for (int w = 0; w < 4; ++w) {
        if ((int)(threadIdx.x >> 6) == w) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int c0 = (lane + 64 * i) * 8;
                if (c0 < p.C) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        sg[0][c0 + j] += dga[i][j];
                        sg[1][c0 + j] += dbe[i][j];
                    }
                }
            }
        }
        __syncthreads();
    }
    float* part = p.part + (size_t)blockIdx.x * 2 * p.C;
    for (int c = threadIdx.x; c < p.C; c += 256) {
        part[c] = sg[0][c];
        part[p.C + c] = sg[1][c];
    }
}

static int ln_bwd_grid(int rows) {
    static const int max_grid = cris_env_int("CRIS_LN_BWD_BLOCKS", 512);
    return cris_grid_1d(rows, 4, max_grid);
}
extern "C" int cris_ln_bwd_parts(int rows) { return ln_bwd_grid(rows); }

extern "C" int cris_ln_bwd(const cris_ln_bwd_params* pp, void* stream) {
    const cris_ln_bwd_params& p = *pp;
    CRIS_CHECK_ARG(p.x && p.gamma && p.mean && p.rstd && p.dx && p.part && p.rows > 0, "null operand");
    CRIS_CHECK_ARG(p.dy || p.dypos || p.dout_f32, "no incoming gradient");
    CRIS_CHECK_ARG((p.C & 7) == 0 && p.C <= 64 * 8 * LN_MAXV && (p.ldx & 7) == 0, "C must be a multiple of 8, <= 2048");
    CRIS_CHECK_ARG(!p.dx_accum || p.dx_f32, "accumulate only into fp32");
    static const int narrow = cris_env_int("CRIS_LN_BWD_V", 1);
    const dim3 grid(ln_bwd_grid(p.rows));        // (= the rows of the partials table the caller sized with cris_ln_bwd_parts)
    if (narrow && p.C <= 512) hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (narrow && p.C <= 1024) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(ln_bwd_kernel<LN_MAXV>, grid, dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// grouped ordered column sums: out[c] = sum_p part[p*ld + c], p = 0 .. nparts-1 in order, for up to CRIS_SUM_GROUP_MAX
// tables in one launch (LayerNorm parameter gradients of a whole arena stage; table passed by value)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_group_kernel(const cris_sum_group g) {
    __shared__ float sh[4][64];
    int ei = 0;                                    // block-uniform; entries >= g.n hold the total block count
#pragma unroll
    for (int i = 1; i < CRIS_SUM_GROUP_MAX; ++i) ei += g.block_start[i] <= (int)blockIdx.x ? 1 : 0;
    const cris_sum_entry e = g.e[ei];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = (blockIdx.x - g.block_start[ei]) * 64 + cl;
    float a = 0.f;
    if (c < e.ncol)
        for (int i = pl; i < e.nparts; i += 4) a += e.part[(size_t)i * e.ld + c];
    sh[pl][cl] = a;
    __syncthreads();
    if (pl == 0 && c < e.ncol) e.out[c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}

extern "C" int cris_sum_tables(const cris_sum_group* gp, void* stream) {
    CRIS_CHECK_ARG(gp && gp->n > 0 && gp->n <= CRIS_SUM_GROUP_MAX, "1 .. CRIS_SUM_GROUP_MAX tables per launch");
    cris_sum_group g = *gp;
    int start = 0;
    for (int i = 0; i < g.n; ++i) {
        CRIS_CHECK_ARG(g.e[i].part && g.e[i].out && g.e[i].nparts > 0 && g.e[i].ncol > 0 && g.e[i].ld >= g.e[i].ncol, "bad table");
        g.block_start[i] = start;
        start += cris_cdiv(g.e[i].ncol, 64);
    }
    for (int i = g.n; i <= CRIS_SUM_GROUP_MAX; ++i) g.block_start[i] = start;
    hipLaunchKernelGGL(sum_group_kernel, dim3(start), dim3(256), 0, (hipStream_t)stream, g);
    CRIS_LAUNCH_CHECK();
    return 0;
}
