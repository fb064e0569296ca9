// Streaming GEMM for the HBM-bound 1x1 convolutions of the large feature maps (K = C_in <= 256, M = 21632 .. 346112 pixels:
// layer1 / layer2 bottleneck 1x1 convolutions and their input gradients).
//
// Why: with K <= 256 a 128x128 output tile is one to four K-steps of work - the tile kernels of gemm.hip / gemm8.hip spend
// their time in per-tile fixed latency (prologue DMA round trip, statistics, store issue) multiplied by the 3-5 rounds of
// blocks such a layer needs: 27-35 us for M 86528 / N 256 / K 64 against ~9 us of HBM traffic (probe, call r03g: 23.6 us remain
// with the main loop AND the stores removed).  Here a block is persistent over a run of consecutive 128-row tiles of one
// 128- (or 64-) column panel:
//   * the weight panel [BN][K] is loaded into LDS ONCE per block and stays resident;
//   * the activation rows stream through an LDS-DMA ring of 16 KB steps (128 rows x 64 k) that runs across tile boundaries -
//     the steps of the next tiles are in flight while a tile's epilogue stores its outputs (counted vmcnt, one raw barrier
//     per step, exactly the ring of conv_gemm_kernel);
//   * epilogue = gemm_epilogue of the tile kernels (same rounding, same BatchNorm partials of 64 / 32 rows), accumulators
//     cleared per tile.
// Results are bit-identical to the tile kernels (same K order).  Lean epilogues, 1x1 / stride 1 / no padding, C % 64 == 0.
#include "gemm_common.h"

#define GS_BM 128
#define GS_STAGES 4

template <int BN, int EPI>
__global__ __launch_bounds__(256) void conv_gemm_stream_kernel(const cris_conv_gemm_params p, int tiles_per_block) {
    constexpr int WAVES_N = BN == 128 ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int WTM = GS_BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    constexpr int NA = GS_BM / 32, NB = BN / 32;               // DMA instructions per wave per 64-wide K-step
    constexpr int A_STEP = GS_BM * 128, B_STEP = BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [B: nk x BN x 128 B][A ring: STAGES x 128 x 128 B]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int nk = p.K / BK;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + GS_BM - 1) / GS_BM;
    // blocks that share activation rows (the column panels of one run of tiles) are neighbours in the grid
    const int chunk = blockIdx.x / tiles_n, tile_n = blockIdx.x - chunk * tiles_n;
    const int tm0 = chunk * tiles_per_block;
    const int ntl = min(tiles_per_block, tiles_m - tm0);       // tiles of this block
    const int n0 = tile_n * BN;
    unsigned char* sB = smem;
    unsigned char* sA = smem + nk * B_STEP;

    const int rsub = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    const unsigned lane_k = (unsigned)kc * 16u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)((size_t)p.M * p.lda * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wt), 0, (int)((size_t)p.N * p.ldb * 2),
                                                                        CRIS_BUF_FLAGS);
    // weight panel: every K-step image, once
    for (int ks = 0; ks < nk; ++ks) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned off = (unsigned)(n0 + (wave + 4 * i) * 8 + rsub) * (unsigned)p.ldb * 2u + (unsigned)ks * 128u + lane_k;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(sB + ks * B_STEP + wave * 1024 + i * 4096), 16, off, 0, 0, 0);
        }
    }
    // activation stream: flat step index s = (local tile) * nk + K-step; rows beyond M / steps beyond the run read zeros
    const int nsteps = ntl * nk;
    int is_tile = 0, is_ks = 0, is_buf = 0;                    // next step to issue
    auto issue_step = [&]() {
        const bool live = is_tile < ntl;
        const int m_base = (tm0 + is_tile) * GS_BM;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m_base + (wave + 4 * i) * 8 + rsub;
            const unsigned off = (live && m < p.M) ? ((unsigned)m * (unsigned)p.lda + (unsigned)p.a_coff) * 2u + (unsigned)is_ks * 128u + lane_k
                                                   : CRIS_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(sA + is_buf * A_STEP + wave * 1024 + i * 4096), 16, off, 0, 0, 0);
        }
        if (++is_ks == nk) { is_ks = 0; ++is_tile; }
        if (++is_buf == GS_STAGES) is_buf = 0;
    };
#pragma unroll
    for (int s = 0; s < GS_STAGES - 1; ++s) {
        issue_step();
    }

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    int buf = 0, ks = 0, tile = 0;
    for (int s = 0; s < nsteps; ++s) {
        CRIS_VMCNT((GS_STAGES - 2) * NA);           // this wave's share of step s (and, the first time, of the weight panel) has landed
        __builtin_amdgcn_s_barrier();
        issue_step();                               // refills the buffer of step s-1
        const unsigned char* sa = sA + buf * A_STEP;
        const unsigned char* sb = sB + ks * B_STEP;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(wm * WTM + i * 32 + fr, q * 2 + fh));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bfr[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(wn * WTN + j * 32 + fr, q * 2 + fh));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++buf == GS_STAGES) buf = 0;
        if (++ks == nk) {                           // tile complete: store it while the next tiles' steps are in flight
            const int tile_m = tm0 + tile;
            gemm_epilogue<EPI, 32, FM, FN>(p, acc, tile_m * GS_BM + wm * WTM, n0 + wn * WTN, tile_m * WAVES_M + wm, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            ks = 0;
            ++tile;
        }
    }
    CRIS_VMCNT(0);
}

// rows per BatchNorm-statistics partial: the wave tile's rows
int cris_gemm_stream_stat_rows(int bn) { return bn == 128 ? 64 : 32; }

// bn: 128 or 64 columns per block; epi 1 / 2 (lean)
int cris_launch_gemm_stream(int bn, const cris_conv_gemm_params& p, int epi, hipStream_t s) {
    typedef void (*kern_t)(const cris_conv_gemm_params, int);
    static const kern_t k128[3] = {nullptr, conv_gemm_stream_kernel<128, 1>, conv_gemm_stream_kernel<128, 2>};
    static const kern_t k64[3] = {nullptr, conv_gemm_stream_kernel<64, 1>, conv_gemm_stream_kernel<64, 2>};
    CRIS_CHECK_ARG(epi == 1 || epi == 2, "streaming GEMM: lean epilogues only");
    CRIS_CHECK_ARG((p.C & 63) == 0 && p.K == p.C && p.K <= 256 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H &&
                       p.OW == p.W, "streaming GEMM: 1x1 convolution with C % 64 == 0, C <= 256");
    const int nk = p.K / BK;
    const int lds = nk * bn * 128 + GS_STAGES * GS_BM * 128;
    const kern_t kern = bn == 128 ? k128[epi] : k64[epi];
    static int attr_done[2][3][5] = {};
    int& done = attr_done[bn == 128][epi][nk];
    if (!done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 128 + GS_STAGES * GS_BM * 128) != hipSuccess) {
            cris_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", __func__);
            return -1;
        }
        done = 1;
    }
    const int tiles_n = cris_cdiv(p.N, bn), tiles_m = cris_cdiv(p.M, GS_BM);
    // about three blocks per CU in flight over the chip, at least two tiles per block where the layer has them
    static const int target_blocks = cris_env_int("CRIS_GEMM_STREAM_BLOCKS", 768);
    int chunks = target_blocks / tiles_n;
    if (chunks < 1) chunks = 1;
    if (chunks > tiles_m) chunks = tiles_m;
    const int tpb = cris_cdiv(tiles_m, chunks);
    chunks = cris_cdiv(tiles_m, tpb);
    hipLaunchKernelGGL(kern, dim3(chunks * tiles_n), dim3(256), lds, s, p, tpb);
    CRIS_LAUNCH_CHECK();
    return 0;
}
