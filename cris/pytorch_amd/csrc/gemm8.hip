// 8-wave "ping-pong" implicit-GEMM tiles for the largest convolution / linear problems (forward and input gradient) on gfx950.
//
// Why a second kernel family: the 4-wave tiles of gemm.hip top out near 0.9 PFLOP/s - every block re-fetches its operand
// panels through the L2 -> LDS path, whose delivered rate (~11-14 TB/s over the chip) caps a BM x BN tile at roughly
// BM*BN/(BM+BN) flop per byte, and their one-barrier-per-K-step loop leaves the matrix pipe idle while a wave reads its
// fragments.  Here a block is 512 threads = 8 waves = TWO wave groups of four (wave w and w+4 share a SIMD) on a 256x256,
// 256x128 or 128x256 tile (one block per CU, 128-144 KB of LDS):
//   * each group owns half of the tile's rows; a K-tile (64 deep) is worked off in phases of 8 v_mfma_f32_32x32x16_bf16
//     (a 64x32 sub-block of the wave tile x K 64); a phase = MEM segment {issue the LDS-DMAs of one staging piece, read
//     this phase's fragments from LDS, counted wait} - s_barrier - MFMA segment {8 MFMAs at raised priority} - s_barrier;
//   * group 1 runs ONE barrier behind group 0, so on every SIMD one wave is in its MFMA segment while its partner is in
//     its MEM segment: the matrix pipe always has an owner and the fragment reads / DMA issue of one wave hide under the
//     MFMAs of the other (MI355X_MICROARCH.md "Two waves per SIMD": alternate matrix-heavy with memory segments);
//   * operands travel HBM/L2 -> LDS by LDS-DMA exactly as in gemm.hip (im2col gather and zero fill as per-lane SOURCE
//     offsets through a raw buffer descriptor, lane-linear LDS image, XOR swizzle on the source side), but in PIECES of
//     128 tile rows (the rows the next phases need first), issued 3-4 phases ahead of their first read; waits are counted
//     (s_waitcnt vmcnt(6) / (8): three or four pieces stay in flight across the barriers), never a drain in the loop.
// Hazards, by construction (ticks = barrier intervals; group 0 runs MEM_p at tick 2p, group 1 at tick 2p+1):
//   RAW  a piece read in phase p is waited for (own vmcnt) at the END of MEM_{p-1} by every wave: both groups have passed
//        that wait before the barrier in front of group 0's MEM_p;
//   WAR  every wave's LDS reads are complete (lgkmcnt(0)) before the barrier that ends its MEM segment, and the DMA that
//        overwrites a slot is issued at least one phase after the slot's last reading phase.
// Only convolutions with C % 64 == 0 (a 64-wide K-tile lies in one tap; every large layer of the networks) come here.
#include "gemm_common.h"

// Diagnostic build switch of tools/probe/gemm8_probe.hip (ablations: which unit bounds the loop).  0 in the library: every
// `if constexpr` below folds away.  bit 0: no fragment reads in the loop; 1: no DMA issue in the loop; 2: no MFMAs;
// 3: no barriers in the loop; 4: lgkmcnt(0) after the barrier instead of before; 5: no s_setprio; 6: no group stagger;
// 7 (results stay correct): flips where a phase's DMAs are issued - in front of the fragment reads (default of the 256x256
// tile) or inside the MFMA segment after the first MFMAs (default of the 256x128 / 128x256 tiles; each default measured).
#ifndef G8_ABL
#define G8_ABL 0
#endif

#define CRIS_WAIT_VM_LGKM0(N) __builtin_amdgcn_s_waitcnt(((N) & 0xF) | ((((N) >> 4) & 3) << 14) | (0x7 << 4))
#define CRIS_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xF | (3 << 14) | (0x7 << 4))
#define CRIS_BARRIER()                              \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        __builtin_amdgcn_s_barrier();               \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)
// barrier / waits of the main loop (the ablation switches act on these only)
#define LOOP_BARRIER()                              \
    do {                                            \
        if constexpr (!(G8_ABL & 8)) CRIS_BARRIER(); \
        else __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
/* N: DMAs that may stay in flight at the wait; N7: the same when the phase's own pieces are issued later (bit 7) */
#define G8_VM(N, N7) (DMA_IN_MFMA ? (N7) : (N))
#define LOOP_WAIT(N, N7)                                                                      \
    do {                                                                                  \
        if constexpr (G8_ABL & 2) { if constexpr (!(G8_ABL & 16)) CRIS_WAIT_LGKM0(); }    \
        else if constexpr (G8_ABL & 16) CRIS_VMCNT(G8_VM(N, N7));                             \
        else CRIS_WAIT_VM_LGKM0(G8_VM(N, N7));                                                \
    } while (0)

// PA x PB sub-blocks of 64 x 32 per wave tile; wave grid 2 (M) x 4 (N):
//   (2,2): 256x256 tile, 2 phases of 16 MFMAs per K-tile, 2 K-tile buffers (128 KB);  (2,1): 256x128;  (1,2): 128x256 - 2 phases
//   of 8 MFMAs per K-tile, 3 K-tile buffers (144 KB);  (1,1): 128x128, one phase of 8 MFMAs per K-tile, FIVE K-tile buffers
//   (160 KB): the variant for problems of at most ~one tile per CU (mid-size layers), whose loop is bound by the operand
//   latency - three K-tiles (96 KB) stay in flight per CU against one or two with the 4-wave tiles' rings
// `bid`: the tile of the problem this block computes, in the XCD-aware order of the launch
template <int PA, int PB, int EPI>
__device__ __forceinline__ void conv_gemm8_tile(const cris_conv_gemm_params& p, int bid, unsigned char* smem) {
    constexpr int WTM = PA * 64, WTN = PB * 32;
    constexpr int BM = 2 * WTM, BN = 4 * WTN;
    constexpr int FM = PA * 2, FN = PB;
    constexpr bool S4 = PA == 2 && PB == 2;
    constexpr bool S1 = PA == 1 && PB == 1;
    constexpr int NBUF = S4 ? 2 : S1 ? 5 : 3;
    constexpr bool DMA_IN_MFMA = (!S4) != ((G8_ABL & 128) != 0);      // where a phase issues its LDS-DMAs (see G8_ABL bit 7)
    constexpr int A_BYTES = BM * 128, TILE_BYTES = (BM + BN) * 128;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // (XCD-aware tile order, as in gemm.hip: cris_xcd_logical_block in the kernels below)
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // which index runs fastest inside an XCD's run of tiles (probe A/B, profiles/r03_gemm8_probe.md): the 256x256 tile keeps an
    // A panel and sweeps the (few) column tiles - the two blocks that share an activation panel run side by side on one L2,
    // 191 against 198 us at M 86528 / N 512 / K 2304; the 128x128 tile keeps a weight panel (35.5 against 39.8 us at
    // M 5408 / N 512 / K 4608); the two 2-phase tiles follow gemm.hip's rule (weights beyond 2 MB: m fastest)
    int tile_m, tile_n;
#ifndef G8_ORDER
#define G8_ORDER 0                                  // probe: 1 = n fastest always, 2 = m fastest always
#endif
    const bool m_fastest = G8_ORDER == 2 || (G8_ORDER == 0 && (S1 || (!S4 && (long)p.N * p.K > (1L << 20))));
    if (m_fastest) {
        tile_n = cris_fast_div(bid, tiles_m, __builtin_amdgcn_rcpf((float)tiles_m));
        tile_m = bid - tile_n * tiles_m;
    } else {
        tile_m = cris_fast_div(bid, tiles_n, __builtin_amdgcn_rcpf((float)tiles_n));
        tile_n = bid - tile_m * tiles_n;
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- DMA roles.  A piece a (a < PA) = tile rows g*WTM + a*64 + [0,64) of both groups g: two 64-row units, this wave
    // moves rows +wave*8 .. +8 of each (lane>>3 = row, lane&7 = 16-B slot).  B piece b = tile columns c*WTN + b*32 + [0,32)
    // of the four wave columns c: unit v covers c = 2v, 2v+1; this wave's rows: c = 2v + wave/4, + (wave%4)*8.
    const int rsub = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);       // logical K-chunk of this lane's slot: slot ^ ((row>>1)&7)
    const unsigned lane_k = (unsigned)kc * 16u;
    const int OHW = p.OH * p.OW;
    int a_pix[PA * 2], a_ih[PA * 2], a_iw[PA * 2];
#pragma unroll
    for (int i = 0; i < PA * 2; ++i) {
        const int m = m0 + (i & 1) * WTM + (i >> 1) * 64 + wave * 8 + rsub;          // i = a*2 + g
        if (m < p.M) {
            const int b = cris_fast_div(m, OHW, __builtin_amdgcn_rcpf((float)OHW));
            const int r = m - b * OHW;
            const int oh = cris_fast_div(r, p.OW, __builtin_amdgcn_rcpf((float)p.OW));
            const int ow = r - oh * p.OW;
            a_pix[i] = b * p.H * p.W;
            a_ih[i] = oh * p.stride - p.pad;
            a_iw[i] = ow * p.stride - p.pad;
        } else {
            a_pix[i] = 0; a_ih[i] = -(1 << 28); a_iw[i] = 0;
        }
    }
    unsigned b_off[PB * 2];
#pragma unroll
    for (int i = 0; i < PB * 2; ++i) {                                                 // i = b*2 + v
        const int n = n0 + (2 * (i & 1) + (wave >> 2)) * WTN + (i >> 1) * 32 + (wave & 3) * 8 + rsub;
        b_off[i] = (unsigned)n * (unsigned)p.ldb * 2u;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.A), 0, (int)((size_t)p.Bn * p.H * p.W * p.lda * 2), CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.Wt), 0, (int)((size_t)p.N * p.ldb * 2), CRIS_BUF_FLAGS);

    // per piece stream: K position (tap, channel) of the next K-tile to stage and the ring buffer it goes to (wave-uniform)
    int sa_c[PA], sa_kh[PA], sa_kw[PA], sa_k[PA], sa_buf[PA], sb_k[PB], sb_buf[PB];
#pragma unroll
    for (int a = 0; a < PA; ++a) { sa_c[a] = 0; sa_kh[a] = 0; sa_kw[a] = 0; sa_k[a] = 0; sa_buf[a] = 0; }
#pragma unroll
    for (int b = 0; b < PB; ++b) { sb_k[b] = 0; sb_buf[b] = 0; }
    unsigned a_base[PA * 2];

    auto issue_A = [&](int a) {
        if (sa_c[a] == 0) {                                       // tap changed: new pixel offsets / padding validity
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int i = a * 2 + g;
                const int ih = a_ih[i] + sa_kh[a], iw = a_iw[i] + sa_kw[a];
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                a_base[i] = ok ? ((unsigned)(a_pix[i] + ih * p.W + iw) * (unsigned)p.lda + (unsigned)p.a_coff) * 2u + lane_k : CRIS_OOB;
            }
        }
        const unsigned kvm = sa_k[a] < p.K ? 0u : CRIS_OOB;      // K-tiles beyond K read zeros (uniform vmcnt arithmetic)
        const unsigned ca = (unsigned)sa_c[a] * 2u;
        unsigned char* dst = smem + sa_buf[a] + (a * 64 + wave * 8) * 128;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const unsigned off = (a_base[a * 2 + g] + ca) | kvm;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(dst + g * WTM * 128), 16, off, 0, 0, 0);
        }
        sa_k[a] += BK;
        sa_c[a] += BK;
        if (sa_c[a] >= p.C) {
            sa_c[a] = 0;
            if (++sa_kw[a] == p.KW) { sa_kw[a] = 0; ++sa_kh[a]; }
        }
        sa_buf[a] += TILE_BYTES;
        if (sa_buf[a] == NBUF * TILE_BYTES) sa_buf[a] = 0;
    };
    auto issue_B = [&](int b) {
        const unsigned kvm = sb_k[b] < p.K ? 0u : CRIS_OOB;
        const unsigned kb = (unsigned)sb_k[b] * 2u + lane_k;
        unsigned char* dst = smem + sb_buf[b] + A_BYTES + ((wave >> 2) * WTN + b * 32 + (wave & 3) * 8) * 128;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const unsigned off = (b_off[b * 2 + v] + kb) | kvm;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(dst + v * 2 * WTN * 128), 16, off, 0, 0, 0);
        }
        sb_k[b] += BK;
        sb_buf[b] += TILE_BYTES;
        if (sb_buf[b] == NBUF * TILE_BYTES) sb_buf[b] = 0;
    };

    // ---- fragment reads (v_mfma_f32_32x32x16_bf16 A/B layout: row = lane&31, k = (lane>>5)*8 .. +8 of a 16-deep slice)
    const int fr = lane & 31, fh = lane >> 5;
    const int swz = (fr >> 1) & 7;                    // rows differ from fr by multiples of 32: same swizzle term
    int cx[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) cx[ks] = (((ks * 2 + fh) ^ swz) & 7) << 4;
    const int rowA = (wm * WTM + fr) * 128, rowB = A_BYTES + (wn * WTN + fr) * 128;
    int rbuf = 0;                                     // ring buffer of the K-tile being computed
    bf16x8 af[2][4], bfr[PB][4];
    auto read_A = [&](int a) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + rbuf + rowA + (a * 64 + i * 32) * 128 + cx[ks]);
            }
        }
    };
    auto read_B = [&](int b) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bfr[b][ks] = *reinterpret_cast<const bf16x8*>(smem + rbuf + rowB + b * 32 * 128 + cx[ks]);
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one 64x32 sub-block x K 64: two accumulators, interleaved so that dependent MFMAs are one instruction apart.  The empty
    // asm statements pin the MFMAs to their segment: an MFMA is a pure register operation, and without them hipcc sinks the
    // MFMAs of one phase across the barriers into the next phase's segment (seen in the disassembly) - which would put the
    // two wave groups' matrix work back to back on the same SIMD instead of alternating it with their memory segments.
#define CRIS_MFMA_SEG(a, b, ISSUE)                                                                                           \
    do {                                                                                                                \
        if constexpr ((G8_ABL & 16) != 0) CRIS_WAIT_LGKM0();                                                            \
        asm volatile("" : "+v"(acc[(a) * 2 + 0][b]), "+v"(acc[(a) * 2 + 1][b]));                                        \
        if constexpr (!(G8_ABL & 32)) __builtin_amdgcn_s_setprio(1);                                                    \
        if constexpr (!(G8_ABL & 4)) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                 \
            acc[(a) * 2 + 0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks], bfr[b][ks], acc[(a) * 2 + 0][b], 0, 0, 0); \
            acc[(a) * 2 + 1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][ks], bfr[b][ks], acc[(a) * 2 + 1][b], 0, 0, 0); \
            if (ks == 0) { ISSUE; }                                                                                     \
        }                                                                                                               \
        asm volatile("" : "+v"(acc[(a) * 2 + 0][b]), "+v"(acc[(a) * 2 + 1][b]));                                        \
        if constexpr (!(G8_ABL & 32)) __builtin_amdgcn_s_setprio(0);                                                                                  \
    } while (0)

    // a 64-row half of the wave tile x both 32-column halves x K 64: 16 MFMAs on four accumulators (256x256 tile)
#define CRIS_MFMA_SEG16(a, ISSUE)                                                                                       \
    do {                                                                                                                \
        if constexpr ((G8_ABL & 16) != 0) CRIS_WAIT_LGKM0();                                                            \
        asm volatile("" : "+v"(acc[(a) * 2][0]), "+v"(acc[(a) * 2 + 1][0]), "+v"(acc[(a) * 2][1]), "+v"(acc[(a) * 2 + 1][1])); \
        if constexpr (!(G8_ABL & 32)) __builtin_amdgcn_s_setprio(1);                                                    \
        if constexpr (!(G8_ABL & 4)) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                 \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                             \
                acc[(a) * 2 + 0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks], bfr[b][ks], acc[(a) * 2 + 0][b], 0, 0, 0); \
                acc[(a) * 2 + 1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][ks], bfr[b][ks], acc[(a) * 2 + 1][b], 0, 0, 0); \
            }                                                                                                           \
            if (ks == 0) { ISSUE; }                                                                                     \
        }                                                                                                               \
        asm volatile("" : "+v"(acc[(a) * 2][0]), "+v"(acc[(a) * 2 + 1][0]), "+v"(acc[(a) * 2][1]), "+v"(acc[(a) * 2 + 1][1])); \
        if constexpr (!(G8_ABL & 32)) __builtin_amdgcn_s_setprio(0);                                                    \
    } while (0)

#define LI_A(a) do { if constexpr (!(G8_ABL & 2) && !DMA_IN_MFMA) issue_A(a); } while (0)
#define LI_B(b) do { if constexpr (!(G8_ABL & 2) && !DMA_IN_MFMA) issue_B(b); } while (0)
#define MI_A(a) do { if constexpr (!(G8_ABL & 2) && DMA_IN_MFMA) issue_A(a); } while (0)
#define MI_B(b) do { if constexpr (!(G8_ABL & 2) && DMA_IN_MFMA) issue_B(b); } while (0)
#define LR_A(a) do { if constexpr (!(G8_ABL & 1)) read_A(a); } while (0)
#define LR_B(b) do { if constexpr (!(G8_ABL & 1)) read_B(b); } while (0)
    const int nk = p.K / BK;
    if constexpr ((G8_ABL & 1) != 0) {
        read_A(0);
#pragma unroll
        for (int b = 0; b < PB; ++b) read_B(b);
    }
    if constexpr (S4) {
        // 256x256: two phases of 16 MFMAs per K-tile.  Piece stream: A0 B0 B1 A1 | A0 B0 B1 A1 ...; phase 1 of K-tile t issues
        // A1(t+1), phase 2 issues A0(t+2) B0(t+2) B1(t+2) (every piece as early as its ring slot allows: two phases ahead)
        issue_A(0); issue_B(0); issue_B(1); issue_A(1); issue_A(0); issue_B(0); issue_B(1);
        CRIS_WAIT_VM_LGKM0(8);                      // A0(0), B0(0), B1(0) of this wave have landed
        CRIS_BARRIER();
        if ((G8_ABL & 64) == 0 && wm == 1) CRIS_BARRIER();                // group 1 runs one barrier behind from here on
        for (int kt = 0; kt < nk; ++kt) {
            // phase 1: rows 0..63 of the wave tile
            LI_A(1);
            LR_A(0);
            LR_B(0);
            LR_B(1);
            LOOP_WAIT(8, 6);               // fragments in registers; A1(kt) landed (4 pieces stay in flight)
            LOOP_BARRIER();
            CRIS_MFMA_SEG16(0, MI_A(1));
            LOOP_BARRIER();
            // phase 2: rows 64..127
            LI_A(0);
            LI_B(0);
            LI_B(1);
            LR_A(1);
            LOOP_WAIT(8, 2);               // A0(kt+1), B0(kt+1), B1(kt+1) landed
            LOOP_BARRIER();
            CRIS_MFMA_SEG16(1, MI_A(0); MI_B(0); MI_B(1));
            LOOP_BARRIER();
            rbuf = rbuf == 0 ? TILE_BYTES : 0;
        }
    } else if constexpr (S1) {
        // 128x128: K-tile t+4 is issued while K-tile t is computed
        constexpr int D = NBUF - 1;
#pragma unroll
        for (int d = 0; d < D; ++d) { issue_A(0); issue_B(0); }
        CRIS_WAIT_VM_LGKM0((D - 1) * 4);            // K-tile 0 of this wave has landed
        CRIS_BARRIER();
        if ((G8_ABL & 64) == 0 && wm == 1) CRIS_BARRIER();
        for (int kt = 0; kt < nk; ++kt) {
            LI_A(0);
            LI_B(0);
            LR_A(0);
            LR_B(0);
            LOOP_WAIT((D - 1) * 4, (D - 2) * 4);     // K-tile kt+1 landed
            LOOP_BARRIER();
            CRIS_MFMA_SEG(0, 0, MI_A(0); MI_B(0));
            LOOP_BARRIER();
            rbuf += TILE_BYTES;
            if (rbuf == NBUF * TILE_BYTES) rbuf = 0;
        }
    } else if constexpr (PA == 2) {
        // 256x128: pieces A0 B A1 | A0 B A1 ...; q1 issues A1(t+1), A0(t+2); q2 issues B(t+2)
        issue_A(0); issue_B(0); issue_A(1); issue_A(0); issue_B(0);
        CRIS_WAIT_VM_LGKM0(6);
        CRIS_BARRIER();
        if ((G8_ABL & 64) == 0 && wm == 1) CRIS_BARRIER();
        for (int kt = 0; kt < nk; ++kt) {
            LI_A(1);
            LI_A(0);
            LR_A(0);
            LR_B(0);
            LOOP_WAIT(8, 4);                  // A1(kt) landed (4 pieces stay in flight)
            LOOP_BARRIER();
            CRIS_MFMA_SEG(0, 0, MI_A(1); MI_A(0));
            LOOP_BARRIER();
            LI_B(0);
            LR_A(1);
            LOOP_WAIT(6, 4);                  // A0(kt+1), B(kt+1) landed
            LOOP_BARRIER();
            CRIS_MFMA_SEG(1, 0, MI_B(0));
            LOOP_BARRIER();
            rbuf += TILE_BYTES;
            if (rbuf == NBUF * TILE_BYTES) rbuf = 0;
        }
    } else {
        // 128x256: pieces B0 A B1 | B0 A B1 ...; q1 issues B1(t+1), B0(t+2); q2 issues A(t+2)
        issue_B(0); issue_A(0); issue_B(1); issue_B(0); issue_A(0);
        CRIS_WAIT_VM_LGKM0(6);
        CRIS_BARRIER();
        if ((G8_ABL & 64) == 0 && wm == 1) CRIS_BARRIER();
        for (int kt = 0; kt < nk; ++kt) {
            LI_B(1);
            LI_B(0);
            LR_A(0);
            LR_B(0);
            LOOP_WAIT(8, 4);
            LOOP_BARRIER();
            CRIS_MFMA_SEG(0, 0, MI_B(1); MI_B(0));
            LOOP_BARRIER();
            LI_A(0);
            LR_B(1);
            LOOP_WAIT(6, 4);
            LOOP_BARRIER();
            CRIS_MFMA_SEG(0, 1, MI_A(0));
            LOOP_BARRIER();
            rbuf += TILE_BYTES;
            if (rbuf == NBUF * TILE_BYTES) rbuf = 0;
        }
    }
    if ((G8_ABL & 64) == 0 && wm == 0) CRIS_BARRIER();    // same number of barriers for every wave
    CRIS_VMCNT(0);                                  // drain the (out-of-range) tail DMAs before the block retires
#undef CRIS_MFMA_SEG
#undef CRIS_MFMA_SEG16
#undef LI_A
#undef MI_A
#undef MI_B
#undef LI_B
#undef LR_A
#undef LR_B

    // (An epilogue staged through LDS - DPP pair exchange, 16-byte row stores instead of 2-byte stores in the C/D layout - was
    // measured on the probe: no difference, 193.8 against 194.8 us at M 86528 / N 512 / K 2304; the fixed ~16 us per 256x256
    // tile are launch + prologue latency + statistics + store issue in about equal parts, not the store width.)
    gemm_epilogue<EPI, 32, FM, FN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, tile_m * 2 + wm, lane);
}

template <int PA, int PB, int EPI>
__global__ __launch_bounds__(512) void conv_gemm8_kernel(const cris_conv_gemm_params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv_gemm8_tile<PA, PB, EPI>(p, cris_xcd_logical_block(blockIdx.x, gridDim.x), smem);
}

// several independent problems on the 128x128 tile in one launch (cris_conv_gemm_group_launch, see gemm.hip)
template <int EPI>
__global__ __launch_bounds__(512) void conv_gemm8_group_kernel(const cris_conv_gemm_group g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lb = cris_xcd_logical_block(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < CRIS_GEMM_GROUP_MAX; ++i) pi += g.block_start[i] <= lb ? 1 : 0;
    const cris_conv_gemm_params p = g.prob[pi];
    conv_gemm8_tile<1, 1, EPI>(p, lb - g.block_start[pi], smem);
}

static int set_lds8(const void* kern, int bytes) {
    return (int)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// launcher used by cris_conv_gemm (gemm.hip): variant 0 = 256x256, 1 = 256x128, 2 = 128x256, 3 = 128x128; epi as in gemm.hip
int cris_launch_gemm8(int variant, const cris_conv_gemm_params& p, int epi, hipStream_t s) {
    typedef void (*kern_t)(const cris_conv_gemm_params);
    if (epi == 3) {                               // lean + BatchNorm-backward partials: the 128x128 tile only
        static const kern_t k3 = conv_gemm8_kernel<1, 1, 3>;
        static const int ready3 = set_lds8((const void*)k3, 5 * (128 + 128) * 128);
        CRIS_CHECK_ARG(ready3 == 0, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        CRIS_CHECK_ARG(variant == 3 && (p.C & 63) == 0, "BatchNorm-backward partials: 8-wave 128x128 tile only, C % 64 == 0");
        hipLaunchKernelGGL(k3, dim3(cris_cdiv(p.M, 128) * cris_cdiv(p.N, 128)), dim3(512), 5 * (128 + 128) * 128, s, p);
        CRIS_LAUNCH_CHECK();
        return 0;
    }
    static const kern_t k[4][3] = {
        {conv_gemm8_kernel<2, 2, 0>, conv_gemm8_kernel<2, 2, 1>, conv_gemm8_kernel<2, 2, 2>},
        {conv_gemm8_kernel<2, 1, 0>, conv_gemm8_kernel<2, 1, 1>, conv_gemm8_kernel<2, 1, 2>},
        {conv_gemm8_kernel<1, 2, 0>, conv_gemm8_kernel<1, 2, 1>, conv_gemm8_kernel<1, 2, 2>},
        {conv_gemm8_kernel<1, 1, 0>, conv_gemm8_kernel<1, 1, 1>, conv_gemm8_kernel<1, 1, 2>}};
    static const int lds[4] = {2 * (256 + 256) * 128, 3 * (256 + 128) * 128, 3 * (128 + 256) * 128, 5 * (128 + 128) * 128};
    static const int bm[4] = {256, 256, 128, 128}, bn[4] = {256, 128, 256, 128};
    static const int ready = [&]() {
        int rc = 0;
        for (int v = 0; v < 4; ++v)
            for (int e = 0; e < 3; ++e) rc |= set_lds8((const void*)k[v][e], lds[v]);
        return rc;
    }();
    if (ready != 0) {
        cris_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (%d)", __func__, ready);
        return ready;
    }
    CRIS_CHECK_ARG(variant >= 0 && variant < 4, "unknown 8-wave variant");
    CRIS_CHECK_ARG((p.C & 63) == 0, "8-wave tiles need C % 64 == 0");
    hipLaunchKernelGGL(k[variant][epi], dim3(cris_cdiv(p.M, bm[variant]) * cris_cdiv(p.N, bn[variant])), dim3(512), lds[variant], s, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

int cris_launch_gemm8_group(const cris_conv_gemm_group& g, int nblocks, int epi, hipStream_t s) {
    typedef void (*kern_t)(const cris_conv_gemm_group);
    static const kern_t k[3] = {conv_gemm8_group_kernel<0>, conv_gemm8_group_kernel<1>, conv_gemm8_group_kernel<2>};
    constexpr int LDS = 5 * (128 + 128) * 128;
    static const int ready = set_lds8((const void*)k[0], LDS) | set_lds8((const void*)k[1], LDS) | set_lds8((const void*)k[2], LDS);
    if (ready != 0) {
        cris_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (%d)", __func__, ready);
        return ready;
    }
    for (int i = 0; i < g.n; ++i) CRIS_CHECK_ARG((g.prob[i].C & 63) == 0, "8-wave tiles need C % 64 == 0");
    hipLaunchKernelGGL(k[epi], dim3(nblocks), dim3(512), LDS, s, g);
    CRIS_LAUNCH_CHECK();
    return 0;
}
