// Shared device helpers for libcris_hip.so (gfx950 / CDNA4 only - no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef unsigned short bf16_t;                                   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;       // MFMA 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4;         // MFMA 16x16x16 A/B fragment (2 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;         // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(16))) float f32x16;       // MFMA 32x32 C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // 16-byte raw buffer load

#define CRIS_WAVE 64

// raw buffer access / LDS-DMA helpers shared by the GEMM-shaped kernels
typedef __attribute__((address_space(3))) void lds_void_t;
#define CRIS_BUF_FLAGS 0x00020000          // V# dword 3 for raw (stride 0) buffers on gfx9 / CDNA
#define CRIS_OOB 0x80000000u               // byte offset beyond every descriptor used here (extents are < 2 GiB): reads as 0
// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
#define CRIS_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 0xF) | ((((N) >> 4) & 3) << 14) | (0x7 << 4) | (0xF << 8))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {                // round to nearest even
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 8 bf16 <-> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
    f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// Cache policy of the large activation / gradient stores (GEMM epilogues, BatchNorm / LayerNorm apply kernels).  0 (default): plain
// stores - dirty lines stay in the XCD's L2 until the end-of-kernel write-back.  1: agent-scope write-through (sc1): the data
// leaves the L2 while the kernel still runs.  2: non-temporal.  A/B switch (-DCRIS_STORE_POLICY=..., tools/build_variants.sh);
// results are identical.
#ifndef CRIS_STORE_POLICY
#define CRIS_STORE_POLICY 0
#endif
#define CRIS_STORE_AUX (CRIS_STORE_POLICY == 1 ? 16 : CRIS_STORE_POLICY == 2 ? 2 : 0)      // buffer-store aux: bit 4 = sc1, bit 1 = nt
__device__ __forceinline__ void cris_st16(void* p, const uint4& v) {
#if CRIS_STORE_POLICY == 1
    const u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(w) : "memory");
#elif CRIS_STORE_POLICY == 2
    const u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(w) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}

// Counter-based dropout decision (restated in oracle/dropout_hash.py - keep the two in sync).
__device__ __forceinline__ uint32_t cris_fmix32(uint32_t v) {
    v ^= v >> 16; v *= 0x85EBCA6Bu; v ^= v >> 13; v *= 0xC2B2AE35u; v ^= v >> 16;
    return v;
}
__device__ __forceinline__ uint32_t cris_drop_key(uint32_t seed, uint32_t stream) { return seed ^ (stream * 0x9E3779B9u); }
#define CRIS_DROP_MUL 0x9E3779B1u
// keep decision from the pre-mixed word h = idx * CRIS_DROP_MUL + key.  For a run of indices idx0 + d the multiply is hoisted:
// h = (idx0 * CRIS_DROP_MUL + key) + d * CRIS_DROP_MUL (mod 2^32) - one add per decision instead of a quarter-rate 32-bit
// multiply-add (the attention kernels draw 30 M decisions per site and step)
__device__ __forceinline__ bool cris_keep_h(uint32_t h, uint32_t thresh) { return cris_fmix32(h) >= thresh; }
__device__ __forceinline__ bool cris_keep(uint32_t key, uint32_t idx, uint32_t thresh) {
    return cris_keep_h(idx * CRIS_DROP_MUL + key, thresh);
}

// n / d for 0 <= n < 2^24, 0 < d (rd = 1.0f / d): a float multiply and one correction step instead of the ~40-instruction integer
// division sequence.  The prologue of a tile kernel ran four to eight of those per lane (pixel -> (image, row, column) of every
// DMA row, tile index, tap of the first K-step): 1.26 us of the 5.1 us a 64x64 block of the M 5408 / N 512 / K 512 problem lives
// (phase stamps, profiles/r05_gemm4_phases.md).  Every pixel / tile count of the supported problems is far below 2^24.
__device__ __forceinline__ int cris_fast_div(int n, int d, float rd) {
    int q = (int)((float)n * rd);
    const int r = n - q * d;
    q += r >= d ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

// n / d through the reciprocal when n is known to fit 24 bits (`small`), the plain division otherwise
// (cris_fast_div is exact while the QUOTIENT stays below 2^22 - tests/test_fast_div_model.py; below 2^24 only a divisor of 3 can
// exceed that: divisors 1 and 2 have exact reciprocals)
__device__ __forceinline__ int cris_div24(int n, int d, bool small) {
    return (small && d != 3) ? cris_fast_div(n, d, __builtin_amdgcn_rcpf((float)d)) : n / d;
}

// (b, y, x, cv) of a flat index over [b][Y][X][CV]: three reciprocal divisions when the index fits 24 bits (every feature map of
// the networks up to batch 32), 64-bit divisions otherwise.  The pooling / resampling kernels spent five 64-bit divisions by
// run-time divisors per 16-byte vector (~100 instructions each) - more than their loads and arithmetic together.
struct cris_idx4 { int cv, x, y, b; };
__device__ __forceinline__ cris_idx4 cris_split4(long idx, int CV, int X, int Y, bool small) {
    cris_idx4 r;
    if (small) {
        const int i = (int)idx;
        const int m = cris_div24(i, CV, true);
        r.cv = i - m * CV;
        const int q = cris_div24(m, X, true);
        r.x = m - q * X;
        r.b = cris_div24(q, Y, true);
        r.y = q - r.b * Y;
    } else {
        r.cv = (int)(idx % CV);
        const long m = idx / CV;
        r.x = (int)(m % X);
        r.y = (int)((m / X) % Y);
        r.b = (int)(m / ((long)X * Y));
    }
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- host side -------------------------------------------------------------------------------
void cris_set_error(const char* fmt, ...);
#define CRIS_CHECK_ARG(cond, msg)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            cris_set_error("%s: %s", __func__, msg);                \
            return -1;                                              \
        }                                                           \
    } while (0)
#define CRIS_LAUNCH_CHECK()                                                          \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            cris_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return (int)e_;                                                          \
        }                                                                            \
    } while (0)

#include <stdlib.h>
// tuning knob read once from the environment (launch geometry only - never changes results)
static inline int cris_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
static inline int cris_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int cris_grid_1d(long work_items, int per_block, int cap = 8192) {
    long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
