// Low-latency ("LL") peer-mailbox words shared by the exchange kernel (p2p.hip) and the BatchNorm kernels that exchange their
// statistics INSIDE the launch that produces and consumes them (norm.hip: cris_bn_finalize_sync, cris_bn_bwd_reduce_sync).
//
// One value = one 8-byte word {float bits | generation << 32} written with a single system-scope store into every rank's
// mailbox; the reader polls the word until its upper half carries the generation it waits for - data and "flag" arrive in the
// same store, so there is no fence, no separate flag and no block-wide step between writing and reading: a thread exchanges
// the values it owns by itself (RCCL's LL protocol, applied per channel).  Layout of the LL region (behind the flag-protocol
// region of p2p.hip, 256-byte aligned), in words:
//   ll [2 parities][slots][world][max_floats]          word (parity, slot, src, i) of rank r's mailbox = value i sent by rank src
// The generation is a device counter of its own (cris_step_advance's exchange_gen) that advances once per step and is never
// rewound - the optimizer step count is, when a checkpoint is loaded into a live trainer, and a generation used twice would
// accept the words of its first use without waiting.
// Reuse: a word is rewritten two generations (steps) later at the earliest; a rank can only reach step t+2's exchange `slot`
// after every rank has taken part in step t+1's - i.e. after every rank's step-t kernel of that slot has completed (kernels of
// one stream run in order) - so nobody still reads generation t when generation t+2 is written.
#pragma once
#include "common.h"
#include "../../../include/cris_hip.h"

// A peer that never arrives raises an error instead of hanging the GPU.  The default limit is WALL TIME (round 6; it was a poll
// count worth 5 - 30 s depending on the memory path, and four ranks time-slicing one GPU lost a peer to the start-up skew of the
// first step - code objects load lazily, seconds per process - once in seven runs of tests/test_p2p_gpu.py): P2P_TIMEOUT_TICKS of
// the constant 100 MHz counter (wall_clock64) = 120 s.  A positive spin_limit in the link / parameter block is still a poll
// count (the start-up self-test and the missing-peer test use fractions of a second).
#define P2P_SPIN_LIMIT (1L << 25)
#define P2P_TIMEOUT_TICKS (120LL * 100000000LL)
// true when the wait that began at `t0` (wall_clock64 ticks) with `spins` polls so far has to give up
__device__ __forceinline__ bool p2p_wait_expired(long spin_limit, long spins, long long t0) {
    if (spin_limit > 0) return spins > spin_limit;
    return (spins & 1023) == 0 && (long long)wall_clock64() - t0 > P2P_TIMEOUT_TICKS;
}

static inline __host__ __device__ size_t p2p_flag_region_bytes(int world, int slots, int max_floats) {
    return ((size_t)2 * slots * world * max_floats + (size_t)2 * slots * world) * 4;
}
static inline __host__ __device__ size_t p2p_ll_offset(int world, int slots, int max_floats) {
    return (p2p_flag_region_bytes(world, slots, max_floats) + 255) & ~(size_t)255;
}
static inline __host__ __device__ size_t p2p_ll_bytes(int world, int slots, int max_floats) {
    return (size_t)2 * slots * world * max_floats * 8;
}

__device__ __forceinline__ int p2p_link_gen(const cris_p2p_link& L) { return (L.gen_dev ? L.gen_dev[0] : L.gen_host) + 1; }   // never 0

__device__ __forceinline__ unsigned long long* p2p_ll_row(const cris_p2p_link& L, int box, int src, int gen) {
    char* base = reinterpret_cast<char*>(L.boxes[box]) + p2p_ll_offset(L.world, L.slots, L.max_floats);
    const size_t entry = ((size_t)(gen & 1) * L.slots + L.slot) * L.world + src;
    return reinterpret_cast<unsigned long long*>(base) + entry * L.max_floats;
}

// this rank's value `idx` of the exchange into every mailbox (the own one included: the sum then reads one place for all ranks)
__device__ __forceinline__ void p2p_ll_send(const cris_p2p_link& L, int gen, int idx, float v) {
    const unsigned long long w = ((unsigned long long)(unsigned)gen << 32) | (unsigned long long)__float_as_uint(v);
    for (int q = 0; q < L.world; ++q)
        __hip_atomic_store(p2p_ll_row(L, q, L.rank, gen) + idx, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sum over the ranks of value `idx`, added in rank order (the same order on every rank: bit-identical results everywhere);
// a peer that never arrives within the poll limit sets `bad`
__device__ __forceinline__ float p2p_ll_recv_sum(const cris_p2p_link& L, int gen, int idx, bool& bad) {
    const long limit = (long)L.spin_limit;
    // a peer already timed out in an EARLIER launch of this rank (the flag is only read here; kernels of one stream run in order):
    // do not wait again - one timeout, not one per exchange, until the host looks at the flag (the values are NaN from here on)
    if (L.err && L.err[0] != 0) {
        bad = true;
        return 0.f;
    }
    float s = 0.f;
    for (int q = 0; q < L.world; ++q) {
        const unsigned long long* src = p2p_ll_row(L, L.rank, q, gen) + idx;
        unsigned long long w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        long spins = 0;
        long long t0 = 0;
        while ((unsigned)(w >> 32) != (unsigned)gen) {
            if (spins == 0) t0 = (long long)wall_clock64();
            __builtin_amdgcn_s_sleep(2);
            if (p2p_wait_expired(limit, ++spins, t0)) {
                bad = true;
                break;
            }
            w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s += __uint_as_float((unsigned)w);
    }
    return s;
}
