// Weight-gradient kernels for gfx950:  dW[N,K] = dY[M,N]^T * X_im2col[M,K]  (+ bias gradient = column sums of dY).
//
// The reduction runs over pixels - the ROW dimension of both operands - so both tiles are staged in their NATURAL row-major
// layout ([pixel][128 columns], 256-B rows) by LDS-DMA (buffer_load_dwordx4 ... lds: no register pass, no ds_write), and the
// MFMA fragments, which need 8 consecutive PIXELS of one column per lane, are fetched with gfx950's transposing LDS read
// (ds_read_b64_tr_b16: the 16 lanes of a group address a [4 rows][16 cols] block and receive its columns; lane mapping and
// layout cost measured in profiles/r01_lds_tr_probe.md).  The bank-conflict swizzle (16-B chunk index ^ ((row & 7) << 1)) is
// applied on the global side of the DMA, whose LDS destination is lane-linear.  Ring of STAGES buffers of MS pixel rows,
// counted vmcnt + one raw barrier per step like conv_gemm_kernel; v_mfma_f32_32x32x16_bf16, 128 x 128 output tile per
// block, wave tile 64 x 64.
//
// Results are DETERMINISTIC (no atomics): a block that owns the whole pixel range of its tile stores the tile; when the
// range is split over several blocks each split stores its partial tile into a workspace slab and cris_wgrad_reduce sums
// the slabs in split order.  Two launch forms:
//   cris_conv_wgrad        one problem, optional split of the pixel range (large-M layers: stem, layer1/2, projector)
//   cris_conv_wgrad_group  up to CRIS_WGRAD_GROUP_MAX problems in ONE launch, problem table passed by value: the mid-size
//                          layers (M <= a few thousand pixels: layer3/4, neck, decoder, text encoder) have 16-600 output
//                          tiles each - too few to fill 256 CUs alone, so the engine queues them per arena stage and
//                          launches them together, longest reductions first, instead of splitting every one of them.
// The gradient is kept in the GEMM layout [n][tap*C + c] (coalesced 128-B runs from the MFMA C/D layout); the optimizer /
// cris_unpack_grads map it back to the parameter layout [n][c][tap].
#include "common.h"
#include "../../../include/cris_hip.h"

#define WG_T 128          // output tile (n) x (k)
#define WG_MS 32          // pixel rows per pipeline step
#define WG_STAGES 3       // ring depth: 3 x 2 x 32 x 256 B = 48 KB LDS -> 3 blocks per CU
#define WG_LDS (WG_STAGES * 2 * WG_MS * 256)
#ifndef WG_RUN
#define WG_RUN 8          // consecutive tiles handed to one XCD (they share the dY tile)
#endif

typedef __attribute__((ext_vector_type(4))) short wg_s16x4;
typedef __attribute__((ext_vector_type(8))) short wg_s16x8;

// The transposing read is issued as inline asm: with the builtin, hipcc's waitcnt pass puts an s_waitcnt vmcnt(0) in front
// of the first read that follows an LDS-DMA (it cannot tell that the DMA targets another ring slot), which would serialise
// the ring.  The asm is opaque to that pass, so the consumer waits are explicit: wg_wait_lds() is an s_waitcnt lgkmcnt(0)
// that carries the fragment registers as in/out operands, which orders every use after it.
template <int OFF>
__device__ __forceinline__ wg_s16x4 wg_tr_read(unsigned lds_addr) {
    wg_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
struct wg_frags {
    wg_s16x4 y[2][2], x[2][2];                     // [fragment][q]
};
template <int KS>
__device__ __forceinline__ void wg_read_slice(wg_frags& f, const unsigned (&ay)[2][2], const unsigned (&ax)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f.y[i][0] = wg_tr_read<KS * 4096>(ay[i][0]);
        f.y[i][1] = wg_tr_read<KS * 4096>(ay[i][1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f.x[j][0] = wg_tr_read<KS * 4096>(ax[j][0]);
        f.x[j][1] = wg_tr_read<KS * 4096>(ax[j][1]);
    }
}
__device__ __forceinline__ void wg_wait_lds(wg_frags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.y[0][0]), "+v"(f.y[0][1]), "+v"(f.y[1][0]), "+v"(f.y[1][1]), "+v"(f.x[0][0]), "+v"(f.x[0][1]),
                   "+v"(f.x[1][0]), "+v"(f.x[1][1]));
}
__device__ __forceinline__ bf16x8 wg_join(const wg_s16x4& lo, const wg_s16x4& hi) {
    const wg_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
template <int KS, int KSL>
__device__ __forceinline__ void wg_slices(f32x16 (&acc)[2][2], wg_frags& cur, const unsigned (&ay)[2][2], const unsigned (&ax)[2][2]) {
    // `cur` holds slice KS (already waited for); fetch slice KS+1 underneath this slice's MFMAs
    wg_frags nxt;
    if constexpr (KS + 1 < KSL) wg_read_slice<KS + 1>(nxt, ay, ax);
    bf16x8 af[2], bfr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = wg_join(cur.y[i][0], cur.y[i][1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) bfr[j] = wg_join(cur.x[j][0], cur.x[j][1]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if constexpr (KS + 1 < KSL) {
        wg_wait_lds(nxt);
        wg_slices<KS + 1, KSL>(acc, nxt, ay, ax);
    }
}

// rows of the pixel range one split covers (host and device agree through this one function)
__host__ __device__ __forceinline__ int wg_rows_per_split(int M, int splits) {
    const int r = (M + splits - 1) / splits;
    return (r + WG_T - 1) / WG_T * WG_T;
}
// floats per workspace slab of one split: the partial dW [N][ldw] followed by the partial bias gradient [pad8(N)]
__host__ __device__ __forceinline__ long wg_slab_floats(int N, int ldw) { return (long)N * ldw + ((N + 7) & ~7); }

// One output tile (k-tile bx, n-tile by) over the pixel range of split bz.  Every split 0 .. p.splits-1 has a non-empty
// range (the launchers normalise p.splits), so every workspace slab is fully written.
template <int MS, int STAGES>
__device__ __forceinline__ void wgrad_tile(const cris_wgrad_params& p, int bx, int by, int bz, unsigned char* smem) {
    constexpr int IMG_BYTES = MS * 256;            // one operand image: MS pixel rows x 128 bf16
    constexpr int STAGE_BYTES = 2 * IMG_BYTES;     // dY image, then X image
    constexpr int ND = MS / 16;                    // DMA instructions per wave per operand per step (4 rows each, 4 waves)
    constexpr int KSL = MS / 16;                   // 16-pixel MFMA slices per step

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int k0 = bx * WG_T;
    const int n0 = by * WG_T;
    const int rows_per = wg_rows_per_split(p.M, p.splits);
    const int m_begin = bz * rows_per;
    const int m_end = min(p.M, m_begin + rows_per);
    const int nsteps = (m_end - m_begin + MS - 1) / MS;

    // ---- DMA role: instruction i of this wave fills rows (wave + 4i)*4 + (lane>>4), LDS slot lane&15 of the row; the
    // 16-B chunk that belongs there is slot ^ ((row & 7) << 1), and row & 7 = 4*(wave&1) + (lane>>4) for every i ----
    const int rsub = lane >> 4;
    const int row7 = ((wave & 1) << 2) + rsub;
    const int cg = (lane & 15) ^ (row7 << 1);      // global 8-column chunk of this lane, 0..15
    const int OHW = p.OH * p.OW;
    const int yn = n0 + cg * 8;
    const bool yvalid = yn < p.N_ld;
    const int xk = k0 + cg * 8;
    const bool xvalid = xk < p.K;
    const int xtap = xvalid ? xk / p.C : 0;
    const int xc = xvalid ? xk - xtap * p.C : 0;
    const int xkh = xtap / p.KW, xkw = xtap - xkh * p.KW;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dY), 0, (int)((size_t)p.M * p.ldy * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.X), 0, (int)((size_t)p.Bn * p.H * p.W * p.ldx * 2), CRIS_BUF_FLAGS);
    const bool lin = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;      // pixel index == row index
    // (b, oh, ow) of this lane's first row of the NEXT step to issue, advanced without divisions
    int rb, roh, row_;
    {
        const int m = m_begin + wave * 4 + rsub;
        rb = m / OHW;
        const int r = m - rb * OHW;
        roh = r / p.OW;
        row_ = r - roh * p.OW;
    }
    int m_issue = m_begin;                         // first pixel row of the next step to issue
    const int d16b = 16 / OHW, d16q = (16 - d16b * OHW) / p.OW, d16r = (16 - d16b * OHW) - d16q * p.OW;
    const int dMSb = MS / OHW, dMSq = (MS - dMSb * OHW) / p.OW, dMSr = (MS - dMSb * OHW) - dMSq * p.OW;
    auto issue_step = [&](int buf) {
        unsigned char* sy = smem + buf * STAGE_BYTES + wave * 1024;               // wave-uniform LDS base of DMA i: + i*4096
        unsigned char* sx = sy + IMG_BYTES;
        int b = rb, oh = roh, ow = row_;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int m = m_issue + (wave + 4 * i) * 4 + rsub;
            const bool mv = m < m_end;
            const unsigned yo = ((unsigned)m * (unsigned)p.ldy + (unsigned)(p.y_coff + yn)) * 2u;
            unsigned xo;
            bool xv = mv && xvalid;
            if (lin) {
                xo = ((unsigned)m * (unsigned)p.ldx + (unsigned)(p.x_coff + xc)) * 2u;
            } else {
                const int ih = oh * p.stride - p.pad + xkh, iw = ow * p.stride - p.pad + xkw;
                xv = xv && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                xo = ((unsigned)((b * p.H + ih) * p.W + iw) * (unsigned)p.ldx + (unsigned)(p.x_coff + xc)) * 2u;
                // this lane's next row is 16 pixels further: 16 = d16b images + d16q rows + d16r pixels, each carry at most 1
                b += d16b; oh += d16q; ow += d16r;
                if (ow >= p.OW) { ow -= p.OW; ++oh; }
                if (oh >= p.OH) { oh -= p.OH; ++b; }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void_t*)(sy + i * 4096), 16, (mv && yvalid) ? yo : CRIS_OOB, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(sx + i * 4096), 16, xv ? xo : CRIS_OOB, 0, 0, 0);
        }
        m_issue += MS;
        if (!lin) {                                                              // first row of the following step
            rb += dMSb; roh += dMSq; row_ += dMSr;
            if (row_ >= p.OW) { row_ -= p.OW; ++roh; }
            if (roh >= p.OH) { roh -= p.OH; ++rb; }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addressing: lane l, read q (= reduction elements 4q..4q+3 of the lane's 8): row 8*(l>>5) + 4q + ((l&15)>>2)
    // of the slice, columns col0 + 16*((l>>4)&1) + 4*(l&3) .. +3; (row & 7) = 4q + ((l&15)>>2) whatever the slice ----
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    int offY[2][2], offX[2][2];
    {
        const int tl = lane & 15, gb = (lane >> 4) & 1;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r7 = 4 * q + (tl >> 2);
                const int row = 8 * fh + r7;
                const int cy = wr * 64 + f * 32 + 16 * gb + 4 * (tl & 3);
                const int cx = wc * 64 + f * 32 + 16 * gb + 4 * (tl & 3);
                offY[f][q] = row * 256 + ((((cy >> 3) ^ (r7 << 1)) & 15) << 4) + (cy & 7) * 2;
                offX[f][q] = row * 256 + ((((cx >> 3) ^ (r7 << 1)) & 15) << 4) + (cx & 7) * 2;
            }
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;               // LDS byte address of the ring
    const bool do_bias = p.dbias != nullptr && bx == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // columns 8*bcg .. +7 of dY over this thread's rows
    const int bcg = (t & 15) ^ ((((t >> 4) & 7) << 1) & 15);

#pragma unroll
    for (int s_ = 0; s_ < STAGES - 1; ++s_) {
        issue_step(s_);
    }
    int buf = 0;
    for (int st = 0; st < nsteps; ++st) {
        CRIS_VMCNT((STAGES - 2) * 2 * ND);          // this wave's share of step st has landed ...
        __builtin_amdgcn_s_barrier();               // ... and everyone's; everyone is also done reading step st-1
        {
            int nb = buf + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            issue_step(nb);                         // steps beyond the split's range read zeros (uniform DMA count)
        }
        const unsigned char* sy = smem + buf * STAGE_BYTES;
        {
            const unsigned sbase = lds_base + (unsigned)(buf * STAGE_BYTES);
            unsigned ay[2][2], ax[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    ay[f][q] = sbase + (unsigned)offY[f][q];
                    ax[f][q] = sbase + (unsigned)(IMG_BYTES + offX[f][q]);
                }
            wg_frags cur;
            wg_read_slice<0>(cur, ay, ax);
            wg_wait_lds(cur);
            wg_slices<0, KSL>(acc, cur, ay, ax);
        }
        if (do_bias) {                              // block-uniform: only the blocks of the first k-tile
            // thread t owns LDS slot t&15 of rows (t>>4) + 16j: row & 7 is the same for all of them, so the slot always holds
            // the same global 8-column chunk (bcg) - MS/16 16-byte reads per step instead of one 2-byte read per row
#pragma unroll
            for (int j = 0; j < MS / 16; ++j) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(sy + ((t >> 4) + 16 * j) * 256 + (t & 15) * 16), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += f[e];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++buf == STAGES) buf = 0;
    }
    CRIS_VMCNT(0);                                  // drain the (out-of-range) tail DMAs before LDS is reused / the block retires

    // destination: the gradient itself when this block owns the whole reduction, else this split's workspace slab
    const bool single = p.splits == 1;
    float* dW = single ? p.dW : p.ws + (size_t)bz * wg_slab_floats(p.N, p.ldw);
    float* dB = single ? p.dbias : dW + (size_t)p.N * p.ldw;
    if (do_bias) {                                   // block-wide column sums: [16 row groups][128 n] through LDS, fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(t >> 4) * 128 + bcg * 8 + e] = bsum[e];
        __syncthreads();
        if (t < 128 && n0 + t < p.N) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sacc += red[g * 128 + t];
            dB[n0 + t] = sacc;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {                       // C/D: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
            const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (n >= p.N) continue;
            float* row = dW + (size_t)n * p.ldw;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + wc * 64 + j * 32 + fr;
                if (k < p.ldw) row[k] = acc[i][j][r];               // (columns K .. ldw are padding: zeros)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 8-wave 256 x 256 tile (round 3).  The 128x128 tile above moves 64 flop per operand byte through the L2 -> LDS path and tops
// out at 0.6 - 0.75 PFLOP/s like the 4-wave forward tiles did; this is the forward family's answer (csrc/gemm8.hip) applied to
// the weight gradient: 512 threads = two wave groups of four (group = 128 output rows n), group 1 one barrier behind group
// 0, one phase per 32-pixel step = MEM segment {4 LDS-DMAs (two 128-column images per operand, each in the layout of the tile
// above, so the transposing-read addressing carries over), 24 ds_read_b64_tr_b16 of BOTH 16-pixel slices, counted wait} -
// barrier - 16 MFMAs - barrier; ring of four steps (128 KB), two steps (64 KB) in flight.  Wave tile 128 (n) x 64 (k).
// Hazards as in gemm8.hip: a step is waited for at the end of the MEM segment one phase before its reads; its buffer is
// refilled one phase after them.
// ------------------------------------------------------------------------------------------------
#define WG8_T 256
#define WG8_STAGES 4
#define WG8_IMG (WG_MS * 256)
#define WG8_STEP (4 * WG8_IMG)                     // [dY n 0..127][dY n 128..255][X k 0..127][X k 128..255]
#define WG8_LDS (WG8_STAGES * WG8_STEP)
#define WG8_BARRIER()                               \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        __builtin_amdgcn_s_barrier();               \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)

template <bool DMA_IN_MFMA>
__device__ __forceinline__ void wgrad8_tile(const cris_wgrad_params& p, int bx, int by, int bz, unsigned char* smem) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int k0 = bx * WG8_T;
    const int n0 = by * WG8_T;
    const int rows_per = wg_rows_per_split(p.M, p.splits);
    const int m_begin = bz * rows_per;
    const int m_end = min(p.M, m_begin + rows_per);
    const int nsteps = (m_end - m_begin + WG_MS - 1) / WG_MS;

    // ---- DMA role: one instruction per image and step; this wave fills rows wave*4 + (lane>>4), LDS slot lane&15 of the row;
    // the 16-B chunk that belongs there is slot ^ ((row & 7) << 1), row & 7 = 4*(wave&1) + (lane>>4)
    const int rsub = lane >> 4;
    const int row7 = ((wave & 1) << 2) + rsub;
    const int cg = (lane & 15) ^ (row7 << 1);
    const int OHW = p.OH * p.OW;
    unsigned ycol[2];
    int xc[2], xkh[2], xkw[2];
    bool xvalid[2];
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int yn = n0 + im * 128 + cg * 8;
        ycol[im] = yn < p.N_ld ? (unsigned)(p.y_coff + yn) * 2u : CRIS_OOB;
        const int xk = k0 + im * 128 + cg * 8;
        xvalid[im] = xk < p.K;
        const int tap = xvalid[im] ? xk / p.C : 0;
        xc[im] = xvalid[im] ? xk - tap * p.C : 0;
        xkh[im] = tap / p.KW;
        xkw[im] = tap - xkh[im] * p.KW;
    }
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dY), 0, (int)((size_t)p.M * p.ldy * 2),
                                                                        CRIS_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.X), 0, (int)((size_t)p.Bn * p.H * p.W * p.ldx * 2), CRIS_BUF_FLAGS);
    const bool lin = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
    int rb, roh, row_;                              // (b, oh, ow) of this lane's row of the NEXT step to issue
    {
        const int m = m_begin + wave * 4 + rsub;
        rb = m / OHW;
        const int r = m - rb * OHW;
        roh = r / p.OW;
        row_ = r - roh * p.OW;
    }
    int m_issue = m_begin, ibuf = 0;
    const int dMSb = WG_MS / OHW, dMSq = (WG_MS - dMSb * OHW) / p.OW, dMSr = (WG_MS - dMSb * OHW) - dMSq * p.OW;
    auto issue_step = [&]() {
        unsigned char* dst = smem + ibuf * WG8_STEP + wave * 1024;
        const int m = m_issue + wave * 4 + rsub;
        const bool mv = m < m_end;
        const unsigned yrow = (unsigned)m * (unsigned)p.ldy * 2u;
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const unsigned off = (mv && ycol[im] != CRIS_OOB) ? yrow + ycol[im] : CRIS_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void_t*)(dst + im * WG8_IMG), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            unsigned xo;
            bool xv = mv && xvalid[im];
            if (lin) {
                xo = ((unsigned)m * (unsigned)p.ldx + (unsigned)(p.x_coff + xc[im])) * 2u;
            } else {
                const int ih = roh * p.stride - p.pad + xkh[im], iw = row_ * p.stride - p.pad + xkw[im];
                xv = xv && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                xo = ((unsigned)((rb * p.H + ih) * p.W + iw) * (unsigned)p.ldx + (unsigned)(p.x_coff + xc[im])) * 2u;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(dst + (2 + im) * WG8_IMG), 16, xv ? xo : CRIS_OOB, 0, 0, 0);
        }
        m_issue += WG_MS;
        if (!lin) {
            rb += dMSb; roh += dMSq; row_ += dMSr;
            if (row_ >= p.OW) { row_ -= p.OW; ++roh; }
            if (roh >= p.OH) { roh -= p.OH; ++rb; }
        }
        if (++ibuf == WG8_STAGES) ibuf = 0;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addressing (as in wgrad_tile): lane l, read q: row 8*(l>>5) + 4q + ((l&15)>>2) of the slice, columns
    // col0 + 16*((l>>4)&1) + 4*(l&3) .. +3 of the image
    const int fr = lane & 31, fh = lane >> 5;
    int offY[4][2], offX[2][2];
    {
        const int tl = lane & 15, gb = (lane >> 4) & 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r7 = 4 * q + (tl >> 2);
            const int row = 8 * fh + r7;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int cy = f * 32 + 16 * gb + 4 * (tl & 3);
                offY[f][q] = wr * WG8_IMG + row * 256 + ((((cy >> 3) ^ (r7 << 1)) & 15) << 4) + (cy & 7) * 2;
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int cx = (wc & 1) * 64 + f * 32 + 16 * gb + 4 * (tl & 3);
                offX[f][q] = (2 + (wc >> 1)) * WG8_IMG + row * 256 + ((((cx >> 3) ^ (r7 << 1)) & 15) << 4) + (cx & 7) * 2;
            }
        }
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;
    const bool do_bias = p.dbias != nullptr && bx == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // columns 8*bcg .. +7 of dY image t>>8 over this thread's rows
    const int bt = t & 255, bimg = t >> 8;
    const int bcg = (bt & 15) ^ ((((bt >> 4) & 7) << 1) & 15);

#pragma unroll
    for (int s_ = 0; s_ < WG8_STAGES - 1; ++s_) {
        issue_step();
    }
    CRIS_VMCNT((WG8_STAGES - 2) * 4);                // step 0 of this wave has landed
    WG8_BARRIER();
    if (wr == 1) WG8_BARRIER();                     // group 1 runs one barrier behind from here on
    int buf = 0;
    for (int st = 0; st < nsteps; ++st) {
        // ---- MEM segment
        if constexpr (!DMA_IN_MFMA) {
            issue_step();                           // step st+3 -> the buffer of step st-1 (steps beyond the range read zeros)
        }
        const unsigned sbase = lds_base + (unsigned)(buf * WG8_STEP);
        wg_s16x4 fy[2][4][2], fx[2][2][2];          // [slice][fragment][q]
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            fy[0][f][0] = wg_tr_read<0>(sbase + (unsigned)offY[f][0]);
            fy[0][f][1] = wg_tr_read<0>(sbase + (unsigned)offY[f][1]);
            fy[1][f][0] = wg_tr_read<4096>(sbase + (unsigned)offY[f][0]);
            fy[1][f][1] = wg_tr_read<4096>(sbase + (unsigned)offY[f][1]);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            fx[0][f][0] = wg_tr_read<0>(sbase + (unsigned)offX[f][0]);
            fx[0][f][1] = wg_tr_read<0>(sbase + (unsigned)offX[f][1]);
            fx[1][f][0] = wg_tr_read<4096>(sbase + (unsigned)offX[f][0]);
            fx[1][f][1] = wg_tr_read<4096>(sbase + (unsigned)offX[f][1]);
        }
        if (do_bias) {                              // block-uniform: only the blocks of the first k-tile
            const unsigned char* sy = smem + buf * WG8_STEP + bimg * WG8_IMG;
#pragma unroll
            for (int j = 0; j < WG_MS / 16; ++j) {
                float f8[8];
                unpack8(*reinterpret_cast<const uint4*>(sy + ((bt >> 4) + 16 * j) * 256 + (bt & 15) * 16), f8);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += f8[e];
            }
        }
        // fragments in registers (the asm carries them: every use is ordered after it), step st+1 landed (2 steps stay in flight)
#define WG8_WAIT_OPERANDS                                                                                                              \
    "+v"(fy[0][0][0]), "+v"(fy[0][0][1]), "+v"(fy[0][1][0]), "+v"(fy[0][1][1]), "+v"(fy[0][2][0]), "+v"(fy[0][2][1]), "+v"(fy[0][3][0]), \
        "+v"(fy[0][3][1]), "+v"(fy[1][0][0]), "+v"(fy[1][0][1]), "+v"(fy[1][1][0]), "+v"(fy[1][1][1]), "+v"(fy[1][2][0]),                \
        "+v"(fy[1][2][1]), "+v"(fy[1][3][0]), "+v"(fy[1][3][1])
        if constexpr (DMA_IN_MFMA) {                // step st+3 is issued later, inside the MFMA segment: one step stays in flight here
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : WG8_WAIT_OPERANDS : : "memory");
        } else
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)"
                     : "+v"(fy[0][0][0]), "+v"(fy[0][0][1]), "+v"(fy[0][1][0]), "+v"(fy[0][1][1]), "+v"(fy[0][2][0]), "+v"(fy[0][2][1]),
                       "+v"(fy[0][3][0]), "+v"(fy[0][3][1]), "+v"(fy[1][0][0]), "+v"(fy[1][0][1]), "+v"(fy[1][1][0]), "+v"(fy[1][1][1]),
                       "+v"(fy[1][2][0]), "+v"(fy[1][2][1]), "+v"(fy[1][3][0]), "+v"(fy[1][3][1])
                     :
                     : "memory");
        asm volatile("" : "+v"(fx[0][0][0]), "+v"(fx[0][0][1]), "+v"(fx[0][1][0]), "+v"(fx[0][1][1]), "+v"(fx[1][0][0]), "+v"(fx[1][0][1]),
                          "+v"(fx[1][1][0]), "+v"(fx[1][1][1]));
        WG8_BARRIER();
        // ---- MFMA segment
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]),
                          "+v"(acc[3][1]));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = wg_join(fx[ks][j][0], fx[ks][j][1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 af = wg_join(fy[ks][i][0], fy[ks][i][1]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
                if constexpr (DMA_IN_MFMA) {
                    if (ks == 0 && i == 0) {
                        issue_step();               // the DMAs of step st+3 issue underneath the MFMAs
                    }
                }
            }
        }
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]),
                          "+v"(acc[3][1]));
        __builtin_amdgcn_s_setprio(0);
        WG8_BARRIER();
        if (++buf == WG8_STAGES) buf = 0;
    }
    if (wr == 0) WG8_BARRIER();
    CRIS_VMCNT(0);                                  // drain the (out-of-range) tail DMAs before LDS is reused / the block retires

    const bool single = p.splits == 1;
    float* dW = single ? p.dW : p.ws + (size_t)bz * wg_slab_floats(p.N, p.ldw);
    float* dB = single ? p.dbias : dW + (size_t)p.N * p.ldw;
    if (do_bias) {                                   // block-wide column sums: [16 row groups][256 n] through LDS, fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(bt >> 4) * 256 + bimg * 128 + bcg * 8 + e] = bsum[e];
        __syncthreads();
        if (t < 256 && n0 + t < p.N) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sacc += red[g * 256 + t];
            dB[n0 + t] = sacc;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {                       // C/D: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
            const int n = n0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (n >= p.N) continue;
            float* row = dW + (size_t)n * p.ldw;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + wc * 64 + j * 32 + fr;
                if (k < p.ldw) row[k] = acc[i][j][r];
            }
        }
    }
}

// XCD-aware block order over a flattened grid of T blocks: workgroups are dispatched round-robin over the 8 XCDs, so
// physical block p runs on XCD p & 7.  Logical ids are handed out in runs of WG_RUN consecutive tiles per XCD (consecutive
// k-tiles of one dY tile, which then stays in that XCD's L2) while every XCD still sees every part of a sorted problem list.
__device__ __forceinline__ int wg_logical_block(int pb, int total) {
    const int full = total / (8 * WG_RUN) * (8 * WG_RUN);
    if (pb >= full) return pb;
    const int xcd = pb & 7, j = pb >> 3;
    return ((j / WG_RUN) * 8 + xcd) * WG_RUN + j % WG_RUN;
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const cris_wgrad_params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tk = (p.K + WG_T - 1) / WG_T, tn = (p.N + WG_T - 1) / WG_T;
    const int lb = wg_logical_block(blockIdx.x, gridDim.x);
    const int bx = lb % tk, by = (lb / tk) % tn, bz = lb / (tk * tn);
    wgrad_tile<WG_MS, WG_STAGES>(p, bx, by, bz, smem);
}

__global__ __launch_bounds__(256) void conv_wgrad_group_kernel(const cris_wgrad_group g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lb = wg_logical_block(blockIdx.x, gridDim.x);
    int pi = 0;                                    // block-uniform; entries >= g.n hold the total block count (> lb)
#pragma unroll
    for (int i = 1; i < CRIS_WGRAD_GROUP_MAX; ++i) pi += g.block_start[i] <= lb ? 1 : 0;
    const cris_wgrad_params p = g.prob[pi];
    const int l = lb - g.block_start[pi];
    const int tk = (p.K + WG_T - 1) / WG_T, tn = (p.N + WG_T - 1) / WG_T;
    const int bx = l % tk, by = (l / tk) % tn, bz = l / (tk * tn);
    wgrad_tile<WG_MS, WG_STAGES>(p, bx, by, bz, smem);
}

template <bool MI>
__global__ __launch_bounds__(512) void conv_wgrad8_kernel(const cris_wgrad_params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tk = (p.K + WG8_T - 1) / WG8_T, tn = (p.N + WG8_T - 1) / WG8_T;
    const int lb = wg_logical_block(blockIdx.x, gridDim.x);
    const int bx = lb % tk, by = (lb / tk) % tn, bz = lb / (tk * tn);
    wgrad8_tile<MI>(p, bx, by, bz, smem);
}

__global__ __launch_bounds__(512) void conv_wgrad8_group_kernel(const cris_wgrad_group g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lb = wg_logical_block(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < CRIS_WGRAD_GROUP_MAX; ++i) pi += g.block_start[i] <= lb ? 1 : 0;
    const cris_wgrad_params p = g.prob[pi];
    const int l = lb - g.block_start[pi];
    const int tk = (p.K + WG8_T - 1) / WG8_T, tn = (p.N + WG8_T - 1) / WG8_T;
    const int bx = l % tk, by = (l / tk) % tn, bz = l / (tk * tn);
    wgrad8_tile<true>(p, bx, by, bz, smem);
}

// dW (and dbias) = sum over the splits' workspace slabs (deterministic: a fixed tree - 16 split lanes each add their splits
// in order, then the lanes are added in lane order).  Block = 16 float4 columns x 16 split lanes.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, long slab, float* __restrict__ dW,
                                                          long n_dw, float* __restrict__ dbias, int N) {
    __shared__ float4 sh[16][17];
    const long n4 = n_dw >> 2;                     // n_dw = N * ldw is a multiple of 4 (checked by the launcher)
    const long nb4 = dbias ? (N + 3) >> 2 : 0;     // the bias part of a slab is padded to 8 floats
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + cl;     // float4 index over [dW | dbias]
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4 + nb4) {
        const float* src = ws + (i < n4 ? i * 4 : n_dw + (i - n4) * 4);
        for (int s = pl; s < splits; s += 16) {
            const float4 b = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    }
    sh[pl][cl] = a;
    __syncthreads();
    if (pl == 0 && i < n4 + nb4) {
#pragma unroll
        for (int j = 1; j < 16; ++j) {
            const float4 b = sh[j][cl];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (i < n4) {
            *reinterpret_cast<float4*>(dW + i * 4) = a;
        } else {
            const long n = (i - n4) * 4;
            const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < N) dbias[n + j] = v[j];
        }
    }
}

// the same reduction for a table of problems in one launch (table by value; block -> problem by a scan of <= 24 starts)
struct wgrad_reduce_job {
    const float* ws; float* dW; float* dbias;
    long slab, n_dw;
    int splits, N, block_start, pad_;
};
struct wgrad_reduce_table {
    int n, pad_;
    wgrad_reduce_job job[CRIS_WGRAD_GROUP_MAX];
};
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const wgrad_reduce_table t) {
    __shared__ float4 sh[16][17];
    int j = 0;
    while (j + 1 < t.n && (int)blockIdx.x >= t.job[j + 1].block_start) ++j;
    const wgrad_reduce_job& q = t.job[j];
    const long n4 = q.n_dw >> 2;
    const long nb4 = q.dbias ? (q.N + 3) >> 2 : 0;
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const long i = (long)((int)blockIdx.x - q.block_start) * 16 + cl;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4 + nb4) {
        const float* src = q.ws + (i < n4 ? i * 4 : q.n_dw + (i - n4) * 4);
        for (int s = pl; s < q.splits; s += 16) {
            const float4 b = *reinterpret_cast<const float4*>(src + (size_t)s * q.slab);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    }
    sh[pl][cl] = a;
    __syncthreads();
    if (pl == 0 && i < n4 + nb4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float4 b = sh[k][cl];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (i < n4) {
            *reinterpret_cast<float4*>(q.dW + i * 4) = a;
        } else {
            const long n = (i - n4) * 4;
            const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (n + k < q.N) q.dbias[n + k] = v[k];
        }
    }
}

static int wgrad_check(const cris_wgrad_params& p, const char* fn) {
#define WG_ARG(cond, msg)                            \
    do {                                             \
        if (!(cond)) {                               \
            cris_set_error("%s: %s", fn, msg);       \
            return -1;                               \
        }                                            \
    } while (0)
    WG_ARG(p.dY && p.X && p.dW, "null operand");
    WG_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.splits > 0, "empty problem");
    WG_ARG((p.C & 7) == 0 && (p.ldx & 7) == 0 && (p.x_coff & 7) == 0, "X channels/ld/offset must be multiples of 8");
    WG_ARG((p.ldy & 7) == 0 && (p.y_coff & 7) == 0 && (p.N_ld & 7) == 0 && p.N_ld >= p.N, "dY ld/offset/N_ld");
    WG_ARG(p.K == p.KH * p.KW * p.C && p.M == p.Bn * p.OH * p.OW, "geometry");
    WG_ARG(p.ldw >= p.K && p.ldw <= (p.K + WG_T - 1) / WG_T * WG_T, "ldw must lie in [K, K rounded up to 128]");
    WG_ARG((uintptr_t)p.dY % 16 == 0 && (uintptr_t)p.X % 16 == 0, "operands must be 16-byte aligned");
    WG_ARG((size_t)p.M * p.ldy * 2 < (1UL << 31) && (size_t)p.Bn * p.H * p.W * p.ldx * 2 < (1UL << 31),
           "operand extent must stay below 2 GiB (32-bit buffer offsets)");
    WG_ARG(p.splits == 1 || (p.ws && ((long)p.N * p.ldw) % 4 == 0 && (uintptr_t)p.ws % 16 == 0 && (uintptr_t)p.dW % 16 == 0),
           "a split reduction needs the workspace (cris_wgrad_ws_floats) and a 16-byte aligned gradient with N*ldw % 4 == 0");
#undef WG_ARG
    return 0;
}

// splits actually launched for a request: every split gets a non-empty pixel range
static int wgrad_effective_splits(int M, int splits) {
    const int rows_per = wg_rows_per_split(M, splits < 1 ? 1 : splits);
    return (M + rows_per - 1) / rows_per;
}
// output tile the launchers use for a problem (p.tile == 0; 128 / 256 there force one): the 8-wave 256x256 kernel for the long,
// wide reductions only.  Measured (profiles/r03_ab_experiments.md): +16-19% on the three K = 4608, M >= 21632 shapes of the
// benchmark (766 vs 643 TFLOP/s), slower than the 4-wave kernel on the mid-size ones (fewer, larger blocks on 256 CUs).
// CRIS_WGRAD8=0 switches the 8-wave kernel off.
static int wgrad_tile_size(const cris_wgrad_params& p) {
    static const int on = cris_env_int("CRIS_WGRAD8", 1);
    static const int min_n = cris_env_int("CRIS_WGRAD8_MIN_N", 192), min_k = cris_env_int("CRIS_WGRAD8_MIN_K", 4096);
    static const int min_m = cris_env_int("CRIS_WGRAD8_MIN_M", 16384);
    if (p.tile == WG_T || p.tile == WG8_T) return p.tile;
    return (on && p.N >= min_n && p.K >= min_k && p.M >= min_m) ? WG8_T : WG_T;
}
extern "C" int cris_conv_wgrad_tile(const cris_wgrad_params* p) { return wgrad_tile_size(*p); }
static int wgrad_blocks(const cris_wgrad_params& p, int tile) { return cris_cdiv(p.K, tile) * cris_cdiv(p.N, tile) * p.splits; }

static int wgrad_lds_ready() {
    static const int rc = (int)hipFuncSetAttribute((const void*)conv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS) |
                          (int)hipFuncSetAttribute((const void*)conv_wgrad_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS) |
                          (int)hipFuncSetAttribute((const void*)conv_wgrad8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WG8_LDS) |
                          (int)hipFuncSetAttribute((const void*)conv_wgrad8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WG8_LDS) |
                          (int)hipFuncSetAttribute((const void*)conv_wgrad8_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG8_LDS);
    if (rc != 0) cris_set_error("cris_conv_wgrad: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (%d)", rc);
    return rc;
}

extern "C" long cris_wgrad_ws_floats(int M, int N, int ldw, int splits) {
    const int s = wgrad_effective_splits(M, splits);
    return s > 1 ? (long)s * wg_slab_floats(N, ldw) : 0;
}

extern "C" int cris_wgrad_reduce(const cris_wgrad_params* pp, void* stream) {
    cris_wgrad_params p = *pp;
    p.splits = wgrad_effective_splits(p.M, p.splits);
    if (p.splits == 1) return 0;
    if (wgrad_check(p, __func__)) return -1;
    const long n_dw = (long)p.N * p.ldw;
    const long cols4 = n_dw / 4 + (p.dbias ? (p.N + 3) / 4 : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cris_cdiv(cols4, 16)), dim3(256), 0, (hipStream_t)stream, p.ws, p.splits,
                       wg_slab_floats(p.N, p.ldw), p.dW, n_dw, p.dbias, p.N);
    CRIS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cris_conv_wgrad(const cris_wgrad_params* pp, void* stream) {
    cris_wgrad_params p = *pp;
    p.splits = wgrad_effective_splits(p.M, p.splits);
    if (wgrad_check(p, __func__)) return -1;
    if (wgrad_lds_ready() != 0) return -1;
    static const int w8_mi = cris_env_int("CRIS_WGRAD8_MI", 1);
    if (wgrad_tile_size(p) == WG8_T) {
        if (w8_mi) hipLaunchKernelGGL(conv_wgrad8_kernel<true>, dim3(wgrad_blocks(p, WG8_T)), dim3(512), WG8_LDS, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(conv_wgrad8_kernel<false>, dim3(wgrad_blocks(p, WG8_T)), dim3(512), WG8_LDS, (hipStream_t)stream, p);
    }
    else hipLaunchKernelGGL(conv_wgrad_kernel, dim3(wgrad_blocks(p, WG_T)), dim3(256), WG_LDS, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return (p.splits > 1 && !p.defer_reduce) ? cris_wgrad_reduce(&p, stream) : 0;
}

extern "C" int cris_wgrad_reduce_group(const cris_wgrad_group* gp, void* stream) {
    CRIS_CHECK_ARG(gp && gp->n > 0 && gp->n <= CRIS_WGRAD_GROUP_MAX, "1 .. CRIS_WGRAD_GROUP_MAX problems per launch");
    wgrad_reduce_table t;
    t.n = 0;
    t.pad_ = 0;
    int start = 0;
    for (int i = 0; i < gp->n; ++i) {
        cris_wgrad_params p = gp->prob[i];
        p.splits = wgrad_effective_splits(p.M, p.splits);          // as cris_conv_wgrad launched it
        if (p.splits <= 1) continue;
        if (wgrad_check(p, __func__)) return -1;
        CRIS_CHECK_ARG(p.ws != nullptr, "split reduction without a workspace");
        wgrad_reduce_job& q = t.job[t.n++];
        q.ws = p.ws; q.dW = p.dW; q.dbias = p.dbias;
        q.slab = wg_slab_floats(p.N, p.ldw);
        q.n_dw = (long)p.N * p.ldw;
        q.splits = p.splits; q.N = p.N; q.block_start = start; q.pad_ = 0;
        const long cols4 = q.n_dw / 4 + (p.dbias ? (p.N + 3) / 4 : 0);
        start += cris_cdiv(cols4, 16);
    }
    if (t.n == 0) return 0;
    hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3(start), dim3(256), 0, (hipStream_t)stream, t);
    CRIS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cris_conv_wgrad_group(const cris_wgrad_group* gp, void* stream) {
    CRIS_CHECK_ARG(gp && gp->n > 0 && gp->n <= CRIS_WGRAD_GROUP_MAX, "1 .. CRIS_WGRAD_GROUP_MAX problems per launch");
    cris_wgrad_group g = *gp;
    int start = 0;
    const int tile = wgrad_tile_size(g.prob[0]);
    for (int i = 0; i < g.n; ++i) {
        g.prob[i].splits = wgrad_effective_splits(g.prob[i].M, g.prob[i].splits);
        if (wgrad_check(g.prob[i], __func__)) return -1;
        CRIS_CHECK_ARG(wgrad_tile_size(g.prob[i]) == tile, "the problems of one group must share the output tile (cris_conv_wgrad_tile)");
        g.block_start[i] = start;
        start += wgrad_blocks(g.prob[i], tile);
    }
    for (int i = g.n; i <= CRIS_WGRAD_GROUP_MAX; ++i) g.block_start[i] = start;
    if (wgrad_lds_ready() != 0) return -1;
    if (tile == WG8_T) hipLaunchKernelGGL(conv_wgrad8_group_kernel, dim3(start), dim3(512), WG8_LDS, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(conv_wgrad_group_kernel, dim3(start), dim3(256), WG_LDS, (hipStream_t)stream, g);
    CRIS_LAUNCH_CHECK();
    for (int i = 0; i < g.n; ++i)
        if (g.prob[i].splits > 1 && cris_wgrad_reduce(&g.prob[i], stream) != 0) return -1;
    return 0;
}
