// Implicit-GEMM convolution / linear kernels (forward and input gradient) for gfx950: the 4-wave tile family and the skinny
// kernel.  (The 8-wave ping-pong tiles for the largest layers are in gemm8.hip, the weight-gradient kernels in wgrad.hip.)
//
// out[M,N] = A_im2col[M,K] * Wt[N,K]^T, bf16 operands, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//   tiles 128x128 / 64x128 / 128x64 / 64x64 (x 64 in K), 256 threads; operands go HBM/L2 -> LDS by LDS-DMA
//   (buffer_load_dwordx4 ... lds; the im2col gather and the zero fill are per-lane SOURCE offsets, the LDS image stays
//   lane-linear) through a 2-3 deep ring with counted vmcnt + one raw barrier per K-step; LDS rows of 128 B with the
//   16-byte chunk index XOR-swizzled by (row>>1)&7 so that ds_read_b128 fragment reads hit distinct 16-B slots
//   (conflict free, see MI355X_MICROARCH.md section LDS); XCD-aware tile order (consecutive tiles of one XCD share a
//   panel).  M <= 144 linear problems (text encoder, per-sample vectors) take a latency-oriented kernel: no staging, K
//   split over the waves, deterministic LDS reduction.  No atomics anywhere.
#include "gemm_common.h"
// LDS-DMA ring depth per tile variant (K-steps in flight = depth - 1); -D switches for the A/B in profiles/r02_ab_experiments.md
#ifndef ST_64x64
#define ST_64x64 3
#endif
#ifndef ST_64x128
#define ST_64x128 3
#endif
#ifndef ST_128x64
#define ST_128x64 3
#endif
#ifndef ST_128x128
#define ST_128x128 2
#endif
#define SKINNY_MAX_M 144      // M <= this and a 1x1 geometry -> skinny kernel (no LDS staging, K split over the waves)

// Phase stamps of the 4-wave tile (tools/probe/gemm4_probe.hip compiles this file with -DG4_PROBE; the library never does): the
// shader clock (s_memtime) of wave 0 of every block at  0 entry, 1 prologue DMAs issued, 2 first K-step landed (first barrier
// passed), 3 K loop done, 4 epilogue stores issued, 5 stores acknowledged - kept in SGPRs and written once at the end, so the
// counted vmcnt waits of the loop see no extra memory operation.  G4_ABL: pieces compiled out (results wrong by construction):
// 1 no MFMAs, 2 no DMA refills inside the loop, 4 no epilogue, 8 nothing at all (the launch floor of this grid / LDS size).
#ifdef G4_PROBE
__device__ unsigned long long* g4_stamps;       // [blocks][8]
#define G4_T(k) t_stamp[k] = __builtin_amdgcn_s_memtime()
#ifndef G4_ABL
#define G4_ABL 0
#endif
#else
#define G4_T(k)
#define G4_ABL 0
#endif


// ------------------------------------------------------------------------------------------------
// forward / dgrad (4-wave tiles; lds_off and gemm_epilogue live in gemm_common.h)
// ------------------------------------------------------------------------------------------------
// Main kernel: BM x BN x 64 tiles, 4 waves, operands staged by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave
// instruction = 8 tile rows of 128 B) into a ring of STAGES LDS buffers with STAGES-1 K-steps in flight: one counted
// s_waitcnt vmcnt + one raw s_barrier per K-step, never a full drain inside the loop.  The LDS image is lane-linear
// (DMA rule), so the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE side:
// the lane that lands on 16-B slot `cpos` of row r fetches logical K-chunk cpos ^ ((r>>1)&7) - a permutation inside one
// 128-B row, so global coalescing is unchanged.  The DMAs are buffer loads (buffer_load_dwordx4 ... lds) through raw
// descriptors over the activation / weight extents: zero fill (spatial padding, M / N / K tails) = an out-of-range byte
// offset, which the hardware returns as 0 - a branch-free per-lane select, so every wave issues exactly NA + NB DMAs per
// K-step and the counted vmcnt below is exact.
// KS = 2 (intra-block split-K): a second group of four waves runs the same loop over the ODD K-steps in its own LDS ring (the
// first group takes the even ones), so a block holds two independent MFMA / LDS dependency chains per SIMD instead of one; the
// second group's accumulators are added to the first's through LDS at the end (fixed order: even steps + odd steps) and the
// first group runs the epilogue.  For the mid-size problems whose grids put barely one block on a CU.
// `bid`: the tile of the problem this block computes, in the XCD-aware order of the launch (conv_gemm_kernel: one problem
// per launch; conv_gemm_group_kernel: several independent problems, see cris_conv_gemm_group).
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, int MT_ = 32, int KS = 1>
__device__ __forceinline__ void conv_gemm_tile(const cris_conv_gemm_params& p, int bid, unsigned char* smem) {
    constexpr int WTM = BM / WAVES_M;          // wave tile rows
    constexpr int WTN = BN / WAVES_N;
    // v_mfma_f32_32x32x16_bf16: one 16-B A chunk + one 16-B B chunk per lane feed 32x32x16 MACs - half the LDS read
    // traffic per flop of the 16x16x32 shape (LDS read bandwidth, 128 B/clk/CU, is what caps the 16x16 form at 64x64
    // wave tiles).  A/B lane layout: row = lane&31, k = (lane>>5)*8 .. +8.  MT_ = 16 selects the 16x16x32 shape (four
    // independent accumulators on a 32x32 wave tile); measured no better on the 64x64 block tile, so unused.
    constexpr int MT = MT_;
    using acc_t = typename std::conditional<MT == 32, f32x16, f32x4>::type;
    constexpr int FM = WTM / MT;
    constexpr int FN = WTN / MT;
    constexpr int NA = BM / 32;                // DMA instructions per wave per K-step (A): 8 rows each, 4 waves
    constexpr int NB = BN / 32;
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE_BYTES = (BM + BN) * 128;

#ifdef G4_PROBE
    unsigned long long t_stamp[6];
#endif
    G4_T(0);
    if (G4_ABL & 8) {                               // (probe: the launch alone - same grid, LDS size and kernel arguments, no work)
        if (p.M < 0) p.colsum[0] = 0.f;
        return;
    }
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = (t >> 6) & 3;             // role inside the group of four (DMA rows, wave tile)
    const int grp = KS > 1 ? (t >> 8) : 0;     // K-split group: K-steps grp, grp + KS, ...
    unsigned char* const ring = smem + grp * (STAGES * STAGE_BYTES);
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;

    // XCD-aware tile order: blocks b, b+8, b+16.. run on the same XCD (round-robin dispatch); each XCD gets a contiguous
    // run of tiles so that its private 4 MB L2 keeps the panel the run shares.  Which index runs fastest decides what is
    // re-streamed: n fastest keeps an A (activation) panel and sweeps the weights - right while the whole weight matrix
    // fits the L2; once it does not (N*K*2 B > 2 MB: the 3x3 / wide 1x1 layers of the neck, decoder and layers 3-4), m
    // fastest keeps ONE 128-column weight panel resident and streams the (much smaller per tile) activation rows once.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    int tile_m, tile_n;
    if ((long)p.N * p.K > (1L << 20)) {
        tile_n = cris_fast_div(bid, tiles_m, __builtin_amdgcn_rcpf((float)tiles_m));
        tile_m = bid - tile_n * tiles_m;
    } else {
        tile_m = cris_fast_div(bid, tiles_n, __builtin_amdgcn_rcpf((float)tiles_n));
        tile_n = bid - tile_m * tiles_n;
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- this lane's DMA role: rows (wave + 4j)*8 + (lane>>3), LDS slot lane&7, logical K-chunk kc ----
    const int rsub = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);       // == (lane&7) ^ ((row>>1)&7) for every j
    const int OHW = p.OH * p.OW;
    // 1x1 / stride 1 / no padding (every linear layer, conv1 / conv3 / downsample of a Bottleneck: most launches of the step): the
    // pixel of output row m IS m - no (image, row, column) decomposition; otherwise two reciprocal divisions per DMA row
    const bool lin = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W;     // wave-uniform
    const float r_ohw = __builtin_amdgcn_rcpf((float)OHW), r_ow = __builtin_amdgcn_rcpf((float)p.OW);
    int a_pix[NA], a_ih[NA], a_iw[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + (wave + 4 * i) * 8 + rsub;
        if (m >= p.M) {
            a_pix[i] = 0; a_ih[i] = -(1 << 28); a_iw[i] = 0;       // never in range -> zeros
        } else if (lin) {
            a_pix[i] = m; a_ih[i] = 0; a_iw[i] = 0;                // (pixel index = a_pix + ih * W + iw with ih = iw = 0)
        } else {
            const int b = cris_fast_div(m, OHW, r_ohw);
            const int r = m - b * OHW;
            const int oh = cris_fast_div(r, p.OW, r_ow);
            const int ow = r - oh * p.OW;
            a_pix[i] = b * p.H * p.W;
            a_ih[i] = oh * p.stride - p.pad;
            a_iw[i] = ow * p.stride - p.pad;
        }
    }
    unsigned b_off[NB];                        // byte offset of this lane's weight rows (rows >= N are out of range -> 0)
#pragma unroll
    for (int i = 0; i < NB; ++i) b_off[i] = (unsigned)(n0 + (wave + 4 * i) * 8 + rsub) * (unsigned)p.ldb * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.A), 0, (int)((size_t)p.Bn * p.H * p.W * p.lda * 2), CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.Wt), 0, (int)((size_t)p.N * p.ldb * 2), CRIS_BUF_FLAGS);
    // running (tap, c) of this lane's chunk (the general K-step issue only: C not a multiple of 64)
    const bool fastk = (p.C & 63) == 0;             // wave-uniform
    int kcur = kc * 8 + grp * BK;
    int c_cur = 0, kh = 0, kw = 0;
    if (!fastk) {
        const int tap = kcur / p.C;
        c_cur = kcur - tap * p.C;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }

    // general K-step issue: the 8-channel chunk of a lane may sit in any tap (C not a multiple of 64)
    auto issue_gen = [&](int buf) {
        unsigned char* sa = ring + buf * STAGE_BYTES + wave * 1024;          // wave-uniform LDS base of DMA j: + j*4096
        unsigned char* sb = sa + A_BYTES;
        const bool kvalid = kcur < p.K;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int ih = a_ih[i] + kh, iw = a_iw[i] + kw;
            const bool ok = kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned off = ((unsigned)(a_pix[i] + ih * p.W + iw) * (unsigned)p.lda + (unsigned)(p.a_coff + c_cur)) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(sa + i * 4096), 16, ok ? off : CRIS_OOB, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned off = b_off[i] + (unsigned)kcur * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(sb + i * 4096), 16, kvalid ? off : CRIS_OOB, 0, 0, 0);
        }
        // advance to this group's next K step
        kcur += BK * KS;
        c_cur += BK * KS;
        while (c_cur >= p.C) {
            c_cur -= p.C;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };
    // fast K-step issue for C % 64 == 0 (every layer of the real networks except the stem): a whole 64-wide K-step lies in
    // ONE tap, so the tap is wave-uniform; the per-row pixel offsets / padding validity are recomputed only when the tap
    // changes (every C/64 steps) and a K-step costs one v_add + v_or per DMA instead of ~20 VALU instructions - the main
    // loop of these kernels is otherwise bound by address arithmetic, not by MFMA or memory.
    int f_kh = 0, f_kw = 0, f_c = 0, f_k = grp * BK;                        // wave-uniform
    bool f_newtap = true;
    if (KS > 1) {                                                           // this group's first step may lie in a later tap
        f_c = f_k;
        while (f_c >= p.C) {
            f_c -= p.C;
            if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
        }
    }
    unsigned a_base[NA];
    const unsigned lane_k = (unsigned)kc * 16u;                             // byte offset of this lane's chunk inside a K-step
    auto issue_fast = [&](int buf) {
        unsigned char* sa = ring + buf * STAGE_BYTES + wave * 1024;
        unsigned char* sb = sa + A_BYTES;
        if (f_newtap) {
            f_newtap = false;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int ih = a_ih[i] + f_kh, iw = a_iw[i] + f_kw;
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                a_base[i] = ok ? ((unsigned)(a_pix[i] + ih * p.W + iw) * (unsigned)p.lda + (unsigned)p.a_coff) * 2u + lane_k : CRIS_OOB;
            }
        }
        const unsigned kvm = f_k < p.K ? 0u : CRIS_OOB;                      // steps beyond K read zeros
        const unsigned ca = (unsigned)f_c * 2u, kb = (unsigned)f_k * 2u + lane_k;
#pragma unroll
        for (int i = 0; i < NA; ++i) {       // (braces matter: hipcc 7.2's host pass drops the kernel stub for a brace-less body)
            const unsigned off = (a_base[i] + ca) | kvm;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(sa + i * 4096), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned off = (b_off[i] + kb) | kvm;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(sb + i * 4096), 16, off, 0, 0, 0);
        }
        f_k += BK * KS;
        f_c += BK * KS;
        while (f_c >= p.C) {                 // (C is a multiple of 64 here: f_c lands on 0 whenever KS = 1)
            f_c -= p.C;
            f_newtap = true;
            if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
        }
    };

    acc_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < (MT == 32 ? 16 : 4); ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    const int fr = lane & (MT - 1), fh = lane / MT;             // fragment row, 16-B chunk inside a k-slice
    constexpr int KSL = MT == 32 ? 4 : 2;                       // k-slices per 64-wide step (16 or 32 deep)
    constexpr int CPS = 8 / KSL;                                // 16-B chunks per slice
    auto issue_stage = [&](int b_) {
        if (fastk) issue_fast(b_);
        else issue_gen(b_);
    };
    {
        // prologue: STAGES-1 K-steps in flight (steps beyond nk read zeros: keeps the vmcnt arithmetic uniform)
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) issue_stage(s);
        G4_T(1);
        int buf = 0;
        const int nkg = (nk + KS - 1) / KS;              // K-steps per group (a group's steps beyond nk read zeros)
        for (int kt = 0; kt < nkg; ++kt) {
            CRIS_VMCNT((STAGES - 2) * (NA + NB));       // this wave's share of K-step kt has landed ...
            __builtin_amdgcn_s_barrier();               // ... and everyone's; everyone is also done reading step kt-1
#ifdef G4_PROBE
            if (kt == 0) G4_T(2);
#endif
            if (!(G4_ABL & 2) || kt == 0) {
                int nb = buf + STAGES - 1;
                if (nb >= STAGES) nb -= STAGES;
                issue_stage(nb);                        // refill the buffer of step kt-1 with step kt+STAGES-1
            }
            const unsigned char* sa = ring + buf * STAGE_BYTES;
            const unsigned char* sb = sa + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) {
                bf16x8 af[FM], bfr[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = wm * WTM + i * MT + fr;
                    af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(row, ks * CPS + fh));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = wn * WTN + j * MT + fr;
                    bfr[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(row, ks * CPS + fh));
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        if (G4_ABL & 1) {               // (probe ablation: keep the fragment reads alive without the MFMA)
                            acc[i][j][0] += (float)af[i][0] + (float)bfr[j][0];
                        } else if constexpr (MT == 32) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);          // keep this step's LDS reads / MFMAs ahead of the next barrier
            if (++buf == STAGES) buf = 0;
        }
    }
    CRIS_VMCNT(0);                                  // drain the (out-of-range) tail DMAs before the block retires
    G4_T(3);

    if constexpr (KS > 1) {
        // group 1's partial sums -> LDS (over the drained rings) -> group 0: acc(even K-steps) + acc(odd K-steps)
        constexpr int NR = MT == 32 ? 16 : 4;
        float* xch = reinterpret_cast<float*>(smem) + (wave * 64 + lane) * (FM * FN * NR + 1);
        __syncthreads();                            // every wave is done reading the operand rings
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < NR; ++r) xch[(i * FN + j) * NR + r] = acc[i][j][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[i][j][r] += xch[(i * FN + j) * NR + r];
    }
    if (!(G4_ABL & 4)) gemm_epilogue<EPI, MT, FM, FN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, tile_m * WAVES_M + wm, lane);
#ifdef G4_PROBE
    if (G4_ABL & 4) {                               // keep the accumulators alive
        float sink = 0.f;
        for (int i = 0; i < FM; ++i)
            for (int j = 0; j < FN; ++j) sink += acc[i][j][0] + acc[i][j][7];
        if (sink == 123.456f) p.colsum[0] = sink;
    }
    G4_T(4);
    CRIS_VMCNT(0);
    G4_T(5);
    if (t == 0 && g4_stamps) {
        unsigned long long* d = g4_stamps + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 6; ++k) d[k] = t_stamp[k];
        d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        d[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    }
#endif
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, int MT_ = 32, int KS = 1>
__global__ __launch_bounds__(256 * KS) void conv_gemm_kernel(const cris_conv_gemm_params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv_gemm_tile<BM, BN, WAVES_M, WAVES_N, STAGES, EPI, MT_, KS>(p, cris_xcd_logical_block(blockIdx.x, gridDim.x), smem);
}

// Several INDEPENDENT problems in one launch (cris_conv_gemm_group): the problem table travels by value in the kernel arguments
// (scalar loads from the kernarg segment: the index is block-uniform), each block finds its problem from the prefix sums of the
// tile counts.  What it buys: the mid-size layers (M <= 5408) put 1 - 3 blocks on a CU and spend 40 - 60 % of their 13 - 30 us in
// launch latency, prologue and epilogue; q / k / v projections, the three f4_proj convolutions of the neck, a Bottleneck's
// conv1 + downsample ... do not depend on each other, so their tiles share one grid, one dispatch and each other's tails.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI>
__global__ __launch_bounds__(256) void conv_gemm_group_kernel(const cris_conv_gemm_group g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lb = cris_xcd_logical_block(blockIdx.x, gridDim.x);
    int pi = 0;                                    // block-uniform; entries >= g.n hold the total block count (> lb)
#pragma unroll
    for (int i = 1; i < CRIS_GEMM_GROUP_MAX; ++i) pi += g.block_start[i] <= lb ? 1 : 0;
    const cris_conv_gemm_params p = g.prob[pi];
    conv_gemm_tile<BM, BN, WAVES_M, WAVES_N, STAGES, EPI>(p, lb - g.block_start[pi], smem);
}

// Skinny kernel (M <= FM*16 rows, 1x1 geometry: text encoder / per-sample vectors).  Such GEMMs are pure latency: one
// block owns all rows x 32 columns and its 8 waves split K (interleaved 32-wide chunks, so the block reads contiguous
// 512-B runs); fragments go straight from L2/HBM into registers (no LDS staging, no barrier in the loop), two K-chunks
// per trip with all loads issued before the first MFMA; the partial accumulators are summed through LDS in a fixed wave
// order (deterministic); full epilogue by wave 0.
#ifndef SK_WAVES
#define SK_WAVES 8
#endif
template <int FM>
__global__ __launch_bounds__(64 * SK_WAVES) void skinny_gemm_kernel(const cris_conv_gemm_params p) {
    constexpr int FN = 2;
    __shared__ float red[FM * FN * 4][64];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * (FN * 16);
    // raw buffer loads: rows >= M / columns >= N / K tails are out-of-range offsets (hardware zeros), no branches
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)((size_t)p.M * p.lda * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wt), 0, (int)((size_t)p.N * p.ldb * 2),
                                                                        CRIS_BUF_FLAGS);
    unsigned aoff[FM], boff[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = i * 16 + fr;
        aoff[i] = m < p.M ? ((unsigned)m * (unsigned)p.lda + (unsigned)p.a_coff) * 2u : CRIS_OOB;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + j * 16 + fr;
        boff[j] = n < p.N ? (unsigned)n * (unsigned)p.ldb * 2u : CRIS_OOB;
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int KSTEP = 32 * SK_WAVES;
    for (int kb = wave * 32; kb < p.K; kb += 2 * KSTEP) {              // wave-uniform trip count (MFMA ignores EXEC)
        const int k0 = kb + fg * 8, k1 = k0 + KSTEP;                  // K % 8 == 0: a chunk is either whole or absent
        const unsigned o0 = k0 < p.K ? (unsigned)k0 * 2u : CRIS_OOB, o1 = k1 < p.K ? (unsigned)k1 * 2u : CRIS_OOB;
        u32x4 av0[FM], bv0[FN], av1[FM], bv1[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) av0[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (aoff[i] | o0) >= CRIS_OOB ? CRIS_OOB : aoff[i] + o0, 0, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) bv0[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (boff[j] | o0) >= CRIS_OOB ? CRIS_OOB : boff[j] + o0, 0, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i) av1[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (aoff[i] | o1) >= CRIS_OOB ? CRIS_OOB : aoff[i] + o1, 0, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) bv1[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (boff[j] | o1) >= CRIS_OOB ? CRIS_OOB : boff[j] + o1, 0, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&av0[i]), *reinterpret_cast<bf16x8*>(&bv0[j]),
                                                                    acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&av1[i]), *reinterpret_cast<bf16x8*>(&bv1[j]),
                                                                    acc[i][j], 0, 0, 0);
            }
    }
    // fixed-order sum over the waves (deterministic): SK_WAVES-1 .. 0 add into the LDS table one after the other
    for (int w = SK_WAVES - 1; w >= 0; --w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* slot = &red[(i * FN + j) * 4 + r][lane];
                        *slot = (w == SK_WAVES - 1) ? acc[i][j][r] : *slot + acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    // the epilogue (thousands of instructions per fragment in its general form) is spread over all waves: wave w finishes
    // the 16x16 fragments w, w+8, ... ; one BatchNorm-statistics part per 16-row fragment row
#pragma unroll 1
    for (int u = wave; u < FM * FN; u += SK_WAVES) {
        const int i = u / FN, j = u - i * FN;
        f32x4 one[1][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) one[0][0][r] = red[u * 4 + r][lane];
        gemm_epilogue<0, 16, 1, 1>(p, one, i * 16, n0 + j * 16, i, lane);
    }
}

// Split-K skinny kernels (M <= 144 rows: the text encoder's 12 layers, forward and input gradient - 108 launches per step).
// The single-pass kernel above puts N/32 = 16 .. 64 blocks on a 256-CU part and walks K in four dependent round trips per wave,
// then eight serial LDS rounds and the general epilogue: 19 - 42 us per launch for 1 - 2 MB of weights.  Here K is ALSO split
// over blocks: block (column pair of fragments, K slice) = 4 waves x one or two 64-deep chunks each - every load of a wave
// is in flight at once - the four partial accumulators are summed through LDS in wave order, and the slice's fp32 partial
// goes to a workspace slab [slice][fragment][lane][4]; a second small launch adds the slabs in slice order (deterministic)
// and runs the general epilogue once per output fragment.  128 - 256 blocks per launch instead of 16 - 64.
#define SK2_WAVES 4
template <int FM>
__global__ __launch_bounds__(64 * SK2_WAVES) void skinny_split_kernel(const cris_conv_gemm_params p, int kslice) {
    constexpr int FN = 2;
    __shared__ float red[SK2_WAVES - 1][FM * FN][64][4];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int nb = blockIdx.x, ks = blockIdx.y;
    const int n0 = nb * (FN * 16);
    const int k_begin = ks * kslice, k_end = min(p.K, k_begin + kslice);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)((size_t)p.M * p.lda * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wt), 0, (int)((size_t)p.N * p.ldb * 2),
                                                                        CRIS_BUF_FLAGS);
    unsigned aoff[FM], boff[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = i * 16 + fr;
        aoff[i] = m < p.M ? ((unsigned)m * (unsigned)p.lda + (unsigned)p.a_coff) * 2u : CRIS_OOB;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + j * 16 + fr;
        boff[j] = n < p.N ? (unsigned)n * (unsigned)p.ldb * 2u : CRIS_OOB;
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // 32-wide chunks of the slice, interleaved over the waves (the block reads contiguous 256-B runs of every row)
    for (int kb = k_begin + wave * 32; kb < k_end; kb += 2 * 32 * SK2_WAVES) {
        const int k0 = kb + fg * 8, k1 = k0 + 32 * SK2_WAVES;
        const unsigned o0 = k0 < k_end ? (unsigned)k0 * 2u : CRIS_OOB, o1 = k1 < k_end ? (unsigned)k1 * 2u : CRIS_OOB;
        u32x4 av0[FM], bv0[FN], av1[FM], bv1[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) av0[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (aoff[i] | o0) >= CRIS_OOB ? CRIS_OOB : aoff[i] + o0, 0, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) bv0[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (boff[j] | o0) >= CRIS_OOB ? CRIS_OOB : boff[j] + o0, 0, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i) av1[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (aoff[i] | o1) >= CRIS_OOB ? CRIS_OOB : aoff[i] + o1, 0, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) bv1[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (boff[j] | o1) >= CRIS_OOB ? CRIS_OOB : boff[j] + o1, 0, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&av0[i]), *reinterpret_cast<bf16x8*>(&bv0[j]),
                                                                    acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&av1[i]), *reinterpret_cast<bf16x8*>(&bv1[j]),
                                                                    acc[i][j], 0, 0, 0);
            }
    }
    // waves 1 .. 3 park their partials in LDS, wave 0 adds them in wave order and stores the slice's slab
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(&red[wave - 1][i * FN + j][lane][0]) = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
        float* slab = p.ws + ((size_t)ks * gridDim.x + nb) * (FM * FN * 256);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 v = acc[i][j];
#pragma unroll
                for (int w = 0; w < SK2_WAVES - 1; ++w) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(&red[w][i * FN + j][lane][0]);
                    v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                }
                *reinterpret_cast<f32x4*>(slab + (i * FN + j) * 256 + lane * 4) = v;
            }
    }
}

// second pass: out fragment (i, column fragment c) = sum over the K slices of its slab entries, then the general epilogue
template <int FM>
__global__ __launch_bounds__(256) void skinny_finish_kernel(const cris_conv_gemm_params p, int nslices, int nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x * 4 + wave;                       // fragment index over [FM][columns / 16]
    const int nfc = nblocks * 2;
    if (u >= FM * nfc) return;
    const int i = u / nfc, c = u - i * nfc;
    const int nb = c >> 1, j = c & 1;
    f32x4 one[1][1];
    one[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (slab loads in batches of four issued together: the kernel is a few waves on an idle chip, i.e. load latency; the
    // additions stay in slice order)
    const float* src = p.ws + (size_t)nb * (FM * 2 * 256) + (i * 2 + j) * 256 + lane * 4;
    const size_t slab = (size_t)nblocks * (FM * 2 * 256);
    for (int k0 = 0; k0 < nslices; k0 += 4) {
        f32x4 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = *reinterpret_cast<const f32x4*>(src + (size_t)min(k0 + q, nslices - 1) * slab);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k0 + q < nslices) { one[0][0][0] += o[q][0]; one[0][0][1] += o[q][1]; one[0][0][2] += o[q][2]; one[0][0][3] += o[q][3]; }
    }
    gemm_epilogue<0, 16, 1, 1>(p, one, i * 16, c * 16, i, lane);
}

// K slices of the split-K skinny launch: enough blocks for the chip, at least 128 of K per slice
static int skinny_split_slices(const cris_conv_gemm_params& p) {
    static const int target = cris_env_int("CRIS_SKINNY_BLOCKS", 192);
    const int nblocks = cris_cdiv(p.N, 32);
    int ks = target / nblocks;
    const int kmax = p.K / 128;
    if (ks > kmax) ks = kmax;
    if (ks < 1) ks = 1;
    return ks;
}
static int skinny_kslice(const cris_conv_gemm_params& p, int slices) { return cris_cdiv(cris_cdiv(p.K, slices), 32) * 32; }

// tile selection (host)
enum { V_SKINNY1 = 0, V_SKINNY9, V_SKINNY9S, V_128x64, V_64x64, V_64x128, V_128x128, V_8W_256x256, V_8W_256x128, V_8W_128x256, V_8W_128x128, V_64x64_K2, V_COUNT };
int cris_launch_gemm8(int variant, const cris_conv_gemm_params& p, int epi, hipStream_t s);       // gemm8.hip

// which epilogue instantiation a problem takes: 0 general, 1 lean, 2 lean + bias / ReLU (see gemm_epilogue)
static int epilogue_kind(const cris_conv_gemm_params& p) {
    const bool plain = p.drop_thresh == 0u && !p.outT && p.out && !p.out_f32 && !(p.resid && p.resid_f32);
    if (plain && !p.bias && p.act == 0) return p.bnr_y ? 3 : 1;          // 3: lean + BatchNorm-backward partials
    return !plain ? 0 : (p.act == 0 || p.act == 1 || p.act == 3) ? 2 : 0;
}

static bool variant_applicable(int v, const cris_conv_gemm_params& p) {
    const bool lin = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W;
    switch (v) {
        case V_SKINNY1: return lin && p.M <= 16;
        case V_SKINNY9: case V_SKINNY9S: return lin && p.M <= SKINNY_MAX_M;
        case V_8W_256x256: case V_8W_256x128: case V_8W_128x256: case V_8W_128x128: return (p.C & 63) == 0;
        default: return v >= 0 && v < V_COUNT;
    }
}

static int pick_variant(const cris_conv_gemm_params& p) {
    const bool lin = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W;
    if (lin && p.M <= 16) return V_SKINNY1;
    // M = 17 .. 144 rows (text encoder): standalone, call r03j - K 2048: split-K skinny 13.0 us, 64x64 tile 16.4, single-pass
    // skinny 28.0; K 512: 64x64 tile 8.3 - 8.5, split-K 9.2 - 10.5, single-pass 14.1 - 14.5.  CRIS_SKINNY_SPLIT=0: single-pass.
    static const int skinny_split = cris_env_int("CRIS_SKINNY_SPLIT", 1);
    if (lin && p.M <= SKINNY_MAX_M) {
        if (!skinny_split) return V_SKINNY9;
        // (an eight-deep ring for these few-block launches - every operand byte of a K <= 512 problem requested in the
        // prologue - changed nothing: 9.0 against 8.3 us standalone, 12.50 against 12.50 ms per step, call r03v: the launches
        // sit on the ~4-5 us floor of any dependent kernel plus the general epilogue, not on operand latency)
        return p.K >= 1024 ? V_SKINNY9S : V_64x64;
    }
    // (Measured and removed, calls r03h / r03i: a persistent streaming kernel for the K <= 256 1x1 convolutions of the large
    // feature maps - weight panel resident in LDS, activation ring running across tile boundaries - ran level with the 128x128
    // tile (22.2 against 21.2 us at M 86528 / N 256 / K 64): those layers were bound by the epilogue's VALU work, which
    // gemm_epilogue_fast32 cut instead - 27.8 -> 21.2 us.)
    // 8-wave ping-pong tiles (gemm8.hip; one block per CU): chosen from the per-shape A/B of tools/gemm_variants.py
    // (profiles/r03_gemm_variants.tsv).  CRIS_GEMM8=0 switches the family off.
    static const int g8 = cris_env_int("CRIS_GEMM8", 1);
    static const int g8_min = cris_env_int("CRIS_GEMM8_MIN_TILES", 150);
    static const int g8_min_k = cris_env_int("CRIS_GEMM8_MIN_K", 256);
    // (upper bound 200 -> 216 in round 5: takes in the M 5408 / N 514 / K 4608 CoordConv layer, 43 x 5 tiles - 37.4 against 47.3 us
    // standalone, 12.012 / 12.003 against 12.061 / 12.026 ms per step, call r05j)
    static const int g8_t128_lo = cris_env_int("CRIS_GEMM8_T128_LO", 100), g8_t128_hi = cris_env_int("CRIS_GEMM8_T128_HI", 216);
    // (lean epilogues only: the general epilogue on a 64x32 .. 128x64 wave tile spills and runs 1.3 - 1.6x longer than on the
    // 4-wave tiles - 25.8 against 15.8 us for the decoder's M 5408 / N 512 / K 512 projections, in-step kernel trace of call r03f)
    if (g8 && epilogue_kind(p) != 0 && (p.C & 63) == 0 && p.N >= 128) {
        const long t128 = (long)cris_cdiv(p.M, 128) * cris_cdiv(p.N, 128);
        // 128x128 with a five-deep ring: problems of 0.4 - 0.8 tiles per CU whose loop is bound by operand latency (mid-size
        // layers: M 5408 x N 512, M 1352 x N 2048, M 21632 x N 128): 36 against 51 us at M 5408 / N 512 / K 4608
        if (p.K >= 512 && t128 >= g8_t128_lo && t128 <= g8_t128_hi) return V_8W_128x128;
        if (p.N > 128 && p.K >= g8_min_k) {
            const long t256 = (long)cris_cdiv(p.M, 256) * cris_cdiv(p.N, 256);
            // (a grid of 1.0 .. 1.5 waves of 256x256 tiles idles half the chip in its second round: halve the rows instead;
            // with K = 256 the 128x128 tile is level or ahead except where the 128x256 tiles fit in one round)
            if (p.K >= 512 && t256 >= g8_min && !(t256 > 256 && t256 <= 400)) return V_8W_256x256;
            const long t128x256 = (long)cris_cdiv(p.M, 128) * cris_cdiv(p.N, 256);
            if (t128x256 >= g8_min && !(t128x256 > 256 && t128x256 <= 400) && (p.K >= 512 || t128x256 <= 256)) return V_8W_128x256;
        }
    }
    // (Measured and rejected, call r03e: the 64x64 / 64x128 tiles with an 8- / 6-deep ring and one block per CU for the M 1352
    // layers - 30.1 against 29.3 us at M 1352 / N 512 / K 4608, the wider one 41 against 29: those layers are not short of
    // operands in flight.)
    // N <= 64: the 64x64 tile beats 128x64 on every such layer of the step (per-shape A/B: 20.8 against 23.8 us at M 86528 /
    // N 64 / K 576, 47.8 against 49.3 at M 346112 / N 32 / K 288); CRIS_GEMM_NARROW_128=1 brings the 128x64 tile back
    static const int narrow128 = cris_env_int("CRIS_GEMM_NARROW_128", 0);
    if (p.N <= 64) return narrow128 ? V_128x64 : V_64x64;
    // too few 128x128 tiles to occupy the chip: halve the tile (2 blocks of 72 KB LDS fit a CU).  Mid-size problems
    // (M <= 8192 rows: layers 3-4, neck, decoder) never take the 128x128 tile even when N is wide: measured on the decoder FFN
    // (M 5408, N 2048, K 512) 37 us with 64-row tiles against 48 us.
    static const int t128_min = cris_env_int("CRIS_GEMM_T128_MIN", 448);
    if (p.M <= 8192 || (long)cris_cdiv(p.M, 128) * cris_cdiv(p.N, 128) < t128_min) {
        // latency-bound mid-size problems: more, smaller blocks (3 per CU at 48 KB LDS) hide each other's pipeline fill,
        // barriers and epilogues - except for long reductions (K >= 4096) over enough rows, where the 64x64 tile's LDS read
        // traffic (two fragment loads per MFMA) is the limit and the 64x128 tile (1.5 per MFMA) wins (measured: M 5408, N 512,
        // K 9216: 96 us against 119 us; K 4608: 54 against 57)
        static const int t64_max = cris_env_int("CRIS_GEMM_T64_MAX", 1024);
        const bool long_k = p.K >= 4096 && p.M >= 4096 && p.N >= 256;
        if (!long_k && (long)cris_cdiv(p.M, 64) * cris_cdiv(p.N, 128) < t64_max) {
            // grids of at most two blocks per CU: the K-steps split over two wave groups inside the block (CRIS_GEMM_KS2=1; K >= 256
            // so that each group has at least two steps).  Measured, call r03af: 12.27 ms per step (grids <= 512 blocks) and 12.20
            // (<= 256) against 12.17 without - twice the waves do not pay for the two-deep rings and the combine; stays off,
            // available by name ("64x64k2")
            static const int ks2 = cris_env_int("CRIS_GEMM_KS2", 0), ks2_max = cris_env_int("CRIS_GEMM_KS2_MAX_BLOCKS", 512);
            static const int ks2_min_k = cris_env_int("CRIS_GEMM_KS2_MIN_K", 256);
            // (the K-split tile has no BatchNorm-backward epilogue: EPI 3 problems keep the plain 64x64 tile)
            if (ks2 && p.K >= ks2_min_k && epilogue_kind(p) != 3 && (long)cris_cdiv(p.M, 64) * cris_cdiv(p.N, 64) <= ks2_max) return V_64x64_K2;
            return V_64x64;
        }
        return V_64x128;
    }
    return V_128x128;
}
// rows per BatchNorm-statistics partial of a variant (= rows of its wave tile)
static int variant_stat_rows(int v) {
    switch (v) {
        case V_SKINNY1: return 16;
        case V_SKINNY9: return 16;
        case V_SKINNY9S: return 16;
        case V_128x128: return 64;
        case V_8W_256x256: case V_8W_256x128: return 128;
        case V_8W_128x256: case V_8W_128x128: return 64;
        default: return 32;
    }
}
static int resolve_variant(const cris_conv_gemm_params& p, int variant) {
    if (variant < 0) return pick_variant(p);
    return variant_applicable(variant, p) ? variant : -1;
}
extern "C" int cris_conv_gemm_stat_rows(const cris_conv_gemm_params* p) { return variant_stat_rows(pick_variant(*p)); }
extern "C" int cris_conv_gemm_variant_stat_rows(const cris_conv_gemm_params* p, int variant) {
    const int v = resolve_variant(*p, variant);
    return v < 0 ? -1 : variant_stat_rows(v);
}
extern "C" long cris_conv_gemm_ws_floats(const cris_conv_gemm_params* p, int variant) {
    const int v = resolve_variant(*p, variant);
    if (v != V_SKINNY9S) return 0;
    return (long)skinny_split_slices(*p) * cris_cdiv(p->N, 32) * (9 * 2 * 256);
}
extern "C" int cris_conv_gemm_plan(const cris_conv_gemm_params* p, int variant, int* epilogue) {
    if (epilogue) *epilogue = epilogue_kind(*p);
    return resolve_variant(*p, variant);
}
extern "C" int cris_conv_gemm_num_variants(void) { return V_COUNT; }
extern "C" const char* cris_conv_gemm_variant_name(int v) {
    static const char* names[V_COUNT] = {"skinny1", "skinny9", "skinny9s", "128x64", "64x64", "64x128", "128x128", "8w256x256", "8w256x128", "8w128x256", "8w128x128", "64x64k2"};
    return (v >= 0 && v < V_COUNT) ? names[v] : "?";
}

static int set_lds(const void* kern, int bytes) {
    return (int)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

extern "C" int cris_conv_gemm(const cris_conv_gemm_params* pp, void* stream) { return cris_conv_gemm_variant(pp, -1, stream); }

static int conv_gemm_check(const cris_conv_gemm_params& p);
int cris_launch_gemm8_group(const cris_conv_gemm_group& g, int nblocks, int epi, hipStream_t s);       // gemm8.hip

extern "C" int cris_conv_gemm_group_launch(const cris_conv_gemm_group* gp, int variant, void* stream) {
    CRIS_CHECK_ARG(gp && gp->n > 0 && gp->n <= CRIS_GEMM_GROUP_MAX, "1 .. CRIS_GEMM_GROUP_MAX problems per launch");
    CRIS_CHECK_ARG(variant == V_128x64 || variant == V_64x64 || variant == V_64x128 || variant == V_128x128 || variant == V_8W_128x128,
                   "group launches run the 4-wave tiles or the 8-wave 128x128 tile");
    if (gp->n == 1) return cris_conv_gemm_variant(&gp->prob[0], variant, stream);
    cris_conv_gemm_group g = *gp;
    static const int bm[V_COUNT] = {0, 0, 0, 128, 64, 64, 128, 256, 256, 128, 128, 64}, bn[V_COUNT] = {0, 0, 0, 64, 64, 128, 128, 256, 128, 256, 128, 64};
    const int epi = epilogue_kind(g.prob[0]);
    CRIS_CHECK_ARG(epi < 3, "BatchNorm-backward partials are not available in grouped launches");
    int start = 0;
    for (int i = 0; i < g.n; ++i) {
        const cris_conv_gemm_params& p = g.prob[i];
        if (conv_gemm_check(p) != 0) return -1;
        CRIS_CHECK_ARG(epilogue_kind(p) == epi, "the problems of a group must share the epilogue instantiation");
        CRIS_CHECK_ARG(variant_applicable(variant, p), "tile variant not applicable to a problem of the group");
        g.block_start[i] = start;
        start += cris_cdiv(p.M, bm[variant]) * cris_cdiv(p.N, bn[variant]);
    }
    for (int i = g.n; i <= CRIS_GEMM_GROUP_MAX; ++i) g.block_start[i] = start;
    hipStream_t s = (hipStream_t)stream;
    if (variant == V_8W_128x128) return cris_launch_gemm8_group(g, start, epi, s);
    constexpr int LDS_128x64 = ST_128x64 * (128 + 64) * 128, LDS_64x128 = ST_64x128 * (64 + 128) * 128;
    constexpr int LDS_128x128 = ST_128x128 * (128 + 128) * 128, LDS_64x64 = ST_64x64 * (64 + 64) * 128;
    typedef void (*kern_t)(const cris_conv_gemm_group);
    static const kern_t k_128x64[3] = {conv_gemm_group_kernel<128, 64, 4, 1, ST_128x64, 0>, conv_gemm_group_kernel<128, 64, 4, 1, ST_128x64, 1>,
                                       conv_gemm_group_kernel<128, 64, 4, 1, ST_128x64, 2>};
    static const kern_t k_64x128[3] = {conv_gemm_group_kernel<64, 128, 2, 2, ST_64x128, 0>, conv_gemm_group_kernel<64, 128, 2, 2, ST_64x128, 1>,
                                       conv_gemm_group_kernel<64, 128, 2, 2, ST_64x128, 2>};
    static const kern_t k_128x128[3] = {conv_gemm_group_kernel<128, 128, 2, 2, ST_128x128, 0>, conv_gemm_group_kernel<128, 128, 2, 2, ST_128x128, 1>,
                                        conv_gemm_group_kernel<128, 128, 2, 2, ST_128x128, 2>};
    static const kern_t k_64x64[3] = {conv_gemm_group_kernel<64, 64, 2, 2, ST_64x64, 0>, conv_gemm_group_kernel<64, 64, 2, 2, ST_64x64, 1>,
                                      conv_gemm_group_kernel<64, 64, 2, 2, ST_64x64, 2>};
    static const int lds_ready = [&]() {
        int rc = 0;
        for (int e = 0; e < 3; ++e)
            rc |= set_lds((const void*)k_128x64[e], LDS_128x64) | set_lds((const void*)k_64x128[e], LDS_64x128) |
                  set_lds((const void*)k_128x128[e], LDS_128x128) | set_lds((const void*)k_64x64[e], LDS_64x64);
        return rc;
    }();
    if (lds_ready != 0) {
        cris_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (%d)", __func__, lds_ready);
        return lds_ready;
    }
    switch (variant) {
        case V_128x64: hipLaunchKernelGGL(k_128x64[epi], dim3(start), dim3(256), LDS_128x64, s, g); break;
        case V_64x64: hipLaunchKernelGGL(k_64x64[epi], dim3(start), dim3(256), LDS_64x64, s, g); break;
        case V_64x128: hipLaunchKernelGGL(k_64x128[epi], dim3(start), dim3(256), LDS_64x128, s, g); break;
        default: hipLaunchKernelGGL(k_128x128[epi], dim3(start), dim3(256), LDS_128x128, s, g);
    }
    CRIS_LAUNCH_CHECK();
    return 0;
}

static int conv_gemm_check(const cris_conv_gemm_params& p) {
    CRIS_CHECK_ARG(p.A && p.Wt, "null operand");
    CRIS_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "empty problem");
    CRIS_CHECK_ARG((p.C & 7) == 0 && (p.lda & 7) == 0 && (p.a_coff & 7) == 0, "A channels/ld/offset must be multiples of 8");
    CRIS_CHECK_ARG((p.ldb & 7) == 0 && (p.K & 7) == 0 && p.ldb >= p.K, "W ld / K must be multiples of 8");
    CRIS_CHECK_ARG(p.K == p.KH * p.KW * p.C, "K != KH*KW*C");
    CRIS_CHECK_ARG(p.M == p.Bn * p.OH * p.OW, "M != Bn*OH*OW");
    CRIS_CHECK_ARG(p.M < (1 << 24) && (long)cris_cdiv(p.M, 64) * cris_cdiv(p.N, 64) < (1L << 22),
                   "more than 2^24 output rows / 2^22 tiles (the reciprocal index arithmetic of the prologue is exact below that)");
    CRIS_CHECK_ARG(p.out || p.outT || p.colsum, "no output");
    CRIS_CHECK_ARG(!p.bnr_y || (epilogue_kind(p) == 3 && p.colsum && p.colsq && p.bnr_mean && p.bnr_invstd && p.bnr_scale && p.bnr_shift &&
                                (p.bnr_ldy & 7) == 0 && (size_t)p.M * p.bnr_ldy * 2 < (1UL << 31)),
                   "BatchNorm-backward partials need the lean epilogue, both tables and the four coefficient vectors");
    CRIS_CHECK_ARG(p.stat_ld == 0 || p.stat_ld >= p.N, "stat_ld < N");
    CRIS_CHECK_ARG(!p.outT || ((p.T_E & 63) == 0 && p.T_L > 0 && (p.T_Lpad & 3) == 0 && p.T_Lpad >= p.T_L && p.M % p.T_L == 0),
                   "bad transposed-store geometry");
    CRIS_CHECK_ARG((uintptr_t)p.A % 16 == 0 && (uintptr_t)p.Wt % 16 == 0, "operands must be 16-byte aligned");
    CRIS_CHECK_ARG((long)p.M * p.N < (1L << 32) || p.drop_thresh == 0u, "dropout index overflow");
    CRIS_CHECK_ARG((size_t)p.Bn * p.H * p.W * p.lda * 2 < (1UL << 31) && ((size_t)p.N + 256) * p.ldb * 2 < (1UL << 31) &&
                       (!p.out || (size_t)p.M * p.ldc * 4 < (1UL << 31)) && (!p.resid || (size_t)p.M * p.ldr * 4 < (1UL << 31)),
                   "operand extent must stay below 2 GiB (32-bit buffer offsets)");
    return 0;
}

extern "C" int cris_conv_gemm_variant(const cris_conv_gemm_params* pp, int variant, void* stream) {
    const cris_conv_gemm_params& p = *pp;
    if (conv_gemm_check(p) != 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    // <= 72 KB per block: two blocks (8 waves) share a CU's 160 KB LDS and hide each other's barriers / epilogues
    constexpr int LDS_128x64 = ST_128x64 * (128 + 64) * 128, LDS_64x128 = ST_64x128 * (64 + 128) * 128;
    constexpr int LDS_128x128 = ST_128x128 * (128 + 128) * 128, LDS_64x64 = ST_64x64 * (64 + 64) * 128;
    typedef void (*kern_t)(const cris_conv_gemm_params);
    // [variant][epilogue: 0 general, 1 lean, 2 lean + bias / ReLU]
    static const kern_t k_128x64[4] = {conv_gemm_kernel<128, 64, 4, 1, ST_128x64, 0>, conv_gemm_kernel<128, 64, 4, 1, ST_128x64, 1>,
                                       conv_gemm_kernel<128, 64, 4, 1, ST_128x64, 2>, conv_gemm_kernel<128, 64, 4, 1, ST_128x64, 3>};
    static const kern_t k_64x128[4] = {conv_gemm_kernel<64, 128, 2, 2, ST_64x128, 0>, conv_gemm_kernel<64, 128, 2, 2, ST_64x128, 1>,
                                       conv_gemm_kernel<64, 128, 2, 2, ST_64x128, 2>, conv_gemm_kernel<64, 128, 2, 2, ST_64x128, 3>};
    static const kern_t k_128x128[4] = {conv_gemm_kernel<128, 128, 2, 2, ST_128x128, 0>, conv_gemm_kernel<128, 128, 2, 2, ST_128x128, 1>,
                                        conv_gemm_kernel<128, 128, 2, 2, ST_128x128, 2>, conv_gemm_kernel<128, 128, 2, 2, ST_128x128, 3>};
    // (measured alternatives for this variant: 16x16x32 MFMA with four accumulators 19.50 vs 19.39 ms/step, a 2-stage ring
    // with 5 blocks per CU 20.00 ms/step - neither helps)
    static const kern_t k_64x64[4] = {conv_gemm_kernel<64, 64, 2, 2, ST_64x64, 0>, conv_gemm_kernel<64, 64, 2, 2, ST_64x64, 1>,
                                      conv_gemm_kernel<64, 64, 2, 2, ST_64x64, 2>, conv_gemm_kernel<64, 64, 2, 2, ST_64x64, 3>};
    // 64x64 with the K-steps split over two wave groups: two-deep ring per group = 64 KB, two blocks (16 waves) per CU
    constexpr int ST_K2 = 2, LDS_64x64K2 = 2 * ST_K2 * (64 + 64) * 128;
    static const kern_t k_64x64k2[3] = {conv_gemm_kernel<64, 64, 2, 2, ST_K2, 0, 32, 2>, conv_gemm_kernel<64, 64, 2, 2, ST_K2, 1, 32, 2>,
                                        conv_gemm_kernel<64, 64, 2, 2, ST_K2, 2, 32, 2>};
    static const int lds_ready = [&]() {
        int rc = 0;
        for (int e = 0; e < 3; ++e) rc |= set_lds((const void*)k_64x64k2[e], LDS_64x64K2);
        for (int e = 0; e < 4; ++e)
            rc |= set_lds((const void*)k_128x64[e], LDS_128x64) | set_lds((const void*)k_64x128[e], LDS_64x128) |
                  set_lds((const void*)k_128x128[e], LDS_128x128) | set_lds((const void*)k_64x64[e], LDS_64x64);
        return rc;
    }();
    if (lds_ready != 0) {
        cris_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (%d)", __func__, lds_ready);
        return lds_ready;
    }
    const int lean = epilogue_kind(p);
    int v = resolve_variant(p, variant);
    CRIS_CHECK_ARG(v >= 0, "tile variant not applicable to this problem");
    // automatic choice without a workspace (a caller that predates the `ws` field): the single-pass skinny kernel (same rows
    // per statistics partial as the split-K one) instead of an error
    if (variant < 0 && v == V_SKINNY9S && !p.ws) v = V_SKINNY9;
    CRIS_CHECK_ARG(!p.bnr_y || v == V_128x64 || v == V_64x64 || v == V_64x128 || v == V_128x128 || v == V_8W_128x128,
                   "BatchNorm-backward partials: the 4-wave tiles and the 8-wave 128x128 tile only (cris_conv_gemm_plan tells which variant runs)");
    switch (v) {
        case V_8W_256x256: case V_8W_256x128: case V_8W_128x256: case V_8W_128x128:
            return cris_launch_gemm8(v - V_8W_256x256, p, lean, s);
        case V_SKINNY1:
            hipLaunchKernelGGL(skinny_gemm_kernel<1>, dim3(cris_cdiv(p.N, 32)), dim3(64 * SK_WAVES), 0, s, p);
            break;
        case V_SKINNY9S: {
            CRIS_CHECK_ARG(p.ws != nullptr, "the split-K skinny variant needs the workspace (cris_conv_gemm_ws_floats)");
            const int slices = skinny_split_slices(p), nblocks = cris_cdiv(p.N, 32);
            hipLaunchKernelGGL(skinny_split_kernel<9>, dim3(nblocks, slices), dim3(64 * SK2_WAVES), 0, s, p, skinny_kslice(p, slices));
            hipLaunchKernelGGL(skinny_finish_kernel<9>, dim3(cris_cdiv(9 * nblocks * 2, 4)), dim3(256), 0, s, p, slices, nblocks);
            break;
        }
        case V_SKINNY9:
            hipLaunchKernelGGL(skinny_gemm_kernel<9>, dim3(cris_cdiv(p.N, 32)), dim3(64 * SK_WAVES), 0, s, p);
            break;
        case V_128x64:
            hipLaunchKernelGGL(k_128x64[lean], dim3(cris_cdiv(p.M, 128) * cris_cdiv(p.N, 64)), dim3(256), LDS_128x64, s, p);
            break;
        case V_64x64:
            hipLaunchKernelGGL(k_64x64[lean], dim3(cris_cdiv(p.M, 64) * cris_cdiv(p.N, 64)), dim3(256), LDS_64x64, s, p);
            break;
        case V_64x64_K2:
            CRIS_CHECK_ARG(lean < 3, "the K-split 64x64 tile has no BatchNorm-backward epilogue");
            hipLaunchKernelGGL(k_64x64k2[lean], dim3(cris_cdiv(p.M, 64) * cris_cdiv(p.N, 64)), dim3(512), LDS_64x64K2, s, p);
            break;
        case V_64x128:
            hipLaunchKernelGGL(k_64x128[lean], dim3(cris_cdiv(p.M, 64) * cris_cdiv(p.N, 128)), dim3(256), LDS_64x128, s, p);
            break;
        default:
            hipLaunchKernelGGL(k_128x128[lean], dim3(cris_cdiv(p.M, 128) * cris_cdiv(p.N, 128)), dim3(256), LDS_128x128, s, p);
    }
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight packing (batched)
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
            if (d.row_scale) v *= d.row_scale[n];
            const long ldF = d.ldF > 0 ? d.ldF : (long)d.taps * d.Cpad;      // (row stride: see cris_pack_desc.ldF)
            d.dstF[(long)n * ldF + (long)tap * d.Cpad + c] = f2bf(v);
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
                if (c < d.Cin && n < d.N) v = d.src[(size_t)c * d.N + n] * (d.row_scale ? d.row_scale[n] : 1.f);
                tile[tx][rr] = v;                                      // tile[n][c]
            } else {                                                   // src[n][c][tap]: read along c
                const int n = nt * 64 + rr, c = ct * 64 + tx;
                if (n < d.N && c < d.Cin) v = d.src[((size_t)n * d.Cin + c) * d.taps + tap] * (d.row_scale ? d.row_scale[n] : 1.f);
                tile[rr][tx] = v;
            }
        }
        __syncthreads();
        for (int rr = ty; rr < 64; rr += 4) {
            const int c = ct * 64 + rr, n = nt * 64 + tx;
            const size_t ldD = d.ldD > 0 ? (size_t)d.ldD : (size_t)d.taps * d.Npad;
            if (c < d.Cin && n < d.Npad) d.dstD[(size_t)c * ldD + (size_t)tapf * d.Npad + n] = f2bf(tile[tx][rr]);
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
