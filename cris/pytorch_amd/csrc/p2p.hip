// Low-latency cross-rank sum of small fp32 vectors through peer-mapped mailboxes (the SyncBN exchange of the native trainer
// when its start-up self-test passes on every rank; CRIS_SYNCBN_P2P=0 keeps the RCCL collectives).
//
// Why: with SyncBatchNorm (reference train.py:97-98) every BatchNorm layer exchanges 2C floats forward and 2C backward -
// 142 collectives per CRIS-R50 step, each a few KB, strictly on the critical path.  Through torch.distributed / RCCL each one
// costs a host call plus a multi-kernel ring or tree; here it is ONE kernel: every rank writes its vector straight into
// every peer's mailbox over xGMI, publishes a flag, waits for the peers' flags and adds the world vectors in rank order
// (the same order on every rank, so all ranks hold bit-identical statistics and stay in lock step).
//
// Memory: the mailbox is fine-grained device memory (stores of a running kernel become visible to peers without a
// kernel boundary), exported with hipIpcGetMemHandle and mapped by every other process.  Layout in floats:
//   data  [2 parities][slots][world][max_floats]
//   flags [2 parities][slots][world]   (int; value = generation that filled the entry)
// `slot` numbers the exchanges inside one step, `gen` is the step counter (+1): an entry is rewritten one step later
// at the earliest.  Two parities (gen & 1) make that safe even with a single exchange per step: a peer can only start
// step t+2 after this rank has joined every exchange of step t+1, i.e. after it finished reading step t.
#include "common.h"
#include "../../../include/cris_hip.h"
#include "p2p_ll.h"
#include <string.h>

#define P2P_MAXV 32                      // 256 threads x 32 = 8192 floats per exchange at most

// the flag-protocol region described above, then (256-byte aligned) the LL words of p2p_ll.h
extern "C" size_t cris_p2p_mailbox_bytes(int world, int slots, int max_floats) {
    if (world <= 0 || slots <= 0 || max_floats <= 0) return 0;
    return p2p_ll_offset(world, slots, max_floats) + p2p_ll_bytes(world, slots, max_floats);
}

extern "C" int cris_p2p_alloc(size_t bytes, void** dev_ptr) {
    CRIS_CHECK_ARG(dev_ptr && bytes > 0, "bad arguments");
    void* p = nullptr;
    // uncached (MTYPE_UC) by default - what RCCL uses for its own peer-visible buffers on gfx942 / gfx950; every access of
    // the exchange kernel is a system-scope atomic anyway.  CRIS_P2P_MEM=0 selects plain fine-grained memory instead.
    static const int mem_kind = cris_env_int("CRIS_P2P_MEM", 1);
    hipError_t e = hipExtMallocWithFlags(&p, bytes, mem_kind ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        cris_set_error("%s: hipExtMallocWithFlags(%s, %zu) failed: %s", __func__, mem_kind ? "uncached" : "finegrained", bytes,
                       hipGetErrorString(e));
        return (int)e;
    }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        cris_set_error("%s: clearing the mailbox failed: %s", __func__, hipGetErrorString(e));
        (void)hipFree(p);
        return (int)e;
    }
    *dev_ptr = p;
    return 0;
}

extern "C" int cris_p2p_free(void* dev_ptr) {
    if (!dev_ptr) return 0;
    hipError_t e = hipFree(dev_ptr);
    if (e != hipSuccess) {
        cris_set_error("%s: %s", __func__, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int cris_p2p_export(void* dev_ptr, void* handle_64_bytes) {
    CRIS_CHECK_ARG(dev_ptr && handle_64_bytes, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_64_bytes), dev_ptr);
    if (e != hipSuccess) {
        cris_set_error("%s: hipIpcGetMemHandle failed: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", __func__, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int cris_p2p_import(const void* handle_64_bytes, void** peer_ptr) {
    CRIS_CHECK_ARG(handle_64_bytes && peer_ptr, "null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle_64_bytes, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        cris_set_error("%s: hipIpcOpenMemHandle failed: %s", __func__, hipGetErrorString(e));
        return (int)e;
    }
    *peer_ptr = p;
    return 0;
}

extern "C" int cris_p2p_close(void* peer_ptr) {
    if (!peer_ptr) return 0;
    hipError_t e = hipIpcCloseMemHandle(peer_ptr);
    if (e != hipSuccess) {
        cris_set_error("%s: %s", __func__, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

__device__ __forceinline__ void p2p_store_f32(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void p2p_allreduce_sum_kernel(const cris_p2p_params p) {
    const int gen = (p.gen_dev ? p.gen_dev[0] : p.gen_host) + 1;            // never 0 (the cleared mailbox)
    const int parity = gen & 1;
    const size_t data_floats = (size_t)2 * p.slots * p.world * p.max_floats;
    const size_t entry = ((size_t)parity * p.slots + p.slot) * p.world;      // [parity][slot][.]
    // 1. my vector into every mailbox (own one included: the sum below then reads one place for all ranks).  The vector is
    // read into registers first and the remote stores are issued back to back: on gfx9 stores count in vmcnt, so a
    // load -> store loop would wait for every remote store to be acknowledged before the next element.
    float v[P2P_MAXV];
#pragma unroll
    for (int k = 0; k < P2P_MAXV; ++k) {
        const int i = threadIdx.x + k * 256;
        v[k] = p.data[min(i, p.n - 1)];
    }
    for (int q = 0; q < p.world; ++q) {
        float* dst = reinterpret_cast<float*>(p.boxes[q]) + (entry + p.rank) * p.max_floats;
#pragma unroll
        for (int k = 0; k < P2P_MAXV; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < p.n) p2p_store_f32(dst + i, v[k]);
        }
    }
    __threadfence_system();                        // the data is visible system-wide before any flag is
    __syncthreads();
    // 2. publish: flag[entry + rank] = gen in every mailbox
    if ((int)threadIdx.x < p.world) {
        int* flag = reinterpret_cast<int*>(reinterpret_cast<float*>(p.boxes[threadIdx.x]) + data_floats) + entry + p.rank;
        __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 3. wait until every rank has published into MY mailbox
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if (p.err && p.err[0] != 0) {                  // a peer already timed out in an earlier launch: do not wait again
        if (threadIdx.x == 0) s_bad = 1;
    } else if ((int)threadIdx.x < p.world) {
        const int* flag = reinterpret_cast<const int*>(reinterpret_cast<float*>(p.boxes[p.rank]) + data_floats) + entry + threadIdx.x;
        long spins = 0;
        const long limit = (long)p.spin_limit;
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != gen) {
            __builtin_amdgcn_s_sleep(8);
            if (p2p_wait_expired(limit, ++spins, t0)) {
                s_bad = 1;
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    // 4. sum in rank order.  Plain loads: every thread has executed the system-scope fence above (write-back + invalidate)
    // after the acquire, and the mailbox is uncached memory, so they cannot hit stale lines of the previous generation.
    const float* mine = reinterpret_cast<const float*>(p.boxes[p.rank]) + entry * p.max_floats;
    const bool bad = s_bad != 0;
    if (bad && threadIdx.x == 0 && p.err) p.err[0] = 1;
#pragma unroll 4
    for (int k = 0; k < P2P_MAXV; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i >= p.n) break;
        float s = 0.f;
        for (int q = 0; q < p.world; ++q) s += mine[(size_t)q * p.max_floats + i];
        p.data[i] = bad ? __int_as_float(0x7fc00000) : s;                   // a missing peer must not pass silently
    }
}

extern "C" int cris_p2p_allreduce_sum(const cris_p2p_params* pp, void* stream) {
    const cris_p2p_params& p = *pp;
    CRIS_CHECK_ARG(p.data && p.boxes, "null operand");
    CRIS_CHECK_ARG(p.world >= 1 && p.world <= 64 && p.rank >= 0 && p.rank < p.world, "rank / world");
    CRIS_CHECK_ARG(p.n > 0 && p.n <= p.max_floats && p.slot >= 0 && p.slot < p.slots, "n / slot out of the mailbox geometry");
    CRIS_CHECK_ARG(p.n <= 256 * P2P_MAXV, "at most 8192 floats per exchange");
    hipLaunchKernelGGL(p2p_allreduce_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// In-place sum of a vector through the LL words (p2p_ll.h): every thread sends and sums the values it owns - no flags, no
// fences, any number of blocks.  The start-up self-test of the words the BatchNorm kernels use (norm.hip).
__global__ __launch_bounds__(256) void p2p_ll_allreduce_kernel(const cris_p2p_link link, float* data, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int gen = p2p_link_gen(link);
    p2p_ll_send(link, gen, i, data[i]);
    bool bad = false;
    const float s = p2p_ll_recv_sum(link, gen, i, bad);
    data[i] = bad ? __int_as_float(0x7fc00000) : s;
    if (bad && link.err) link.err[0] = 1;
}

extern "C" int cris_p2p_ll_allreduce_sum(const cris_p2p_link* lp, float* data, int n, void* stream) {
    CRIS_CHECK_ARG(lp && data && lp->boxes, "null operand");
    const cris_p2p_link& l = *lp;
    CRIS_CHECK_ARG(l.world >= 1 && l.world <= 64 && l.rank >= 0 && l.rank < l.world, "rank / world");
    CRIS_CHECK_ARG(n > 0 && n <= l.max_floats && l.slot >= 0 && l.slot < l.slots, "n / slot out of the mailbox geometry");
    hipLaunchKernelGGL(p2p_ll_allreduce_kernel, dim3(cris_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, l, data, n);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Gradient exchange over the peer-mapped gradient arenas (include/cris_hip.h: cris_p2p_arena_allreduce; opt-in).
//
// xGMI on an 8 x MI355X node is a full mesh of point-to-point links: a ring all-reduce is bound by ONE link per hop, a direct
// exchange uses all seven at once.  Two-shot form, in place, over 8-byte words (two floats):
//   reduce-scatter  rank r owns slice r of the range; for every word of its slice it reads the word from every rank's arena in
//                   RANK ORDER (its own through the local pointer, the peers' through their IPC mappings: system-scope loads,
//                   nothing of a remote arena may be served from this GPU's L2) and stores the sum into its own arena with a
//                   system-scope store (written through: the peers read it next);
//   all-gather      every rank copies the other ranks' reduced slices into its own arena.
// Every element is summed once, on its owner, in one fixed order: all ranks end with the same bits, and the same bits as a
// sequential sum over the ranks (tests/test_p2p_gpu.py).  Barriers = one LL word per (rank, barrier) in the peer mailboxes
// (p2p_ll.h): written by block 0, polled by one thread of every block.
//   ready    before a peer's arena is read: its backward has issued the range (kernels of one stream run in order, and a
//            kernel's end writes its stores back to memory);
//   reduced  before a reduced slice is read;
//   done     before the arena may be overwritten (the next backward) every peer has finished reading - waited for by the third
//            launch, so the exchange as a whole is complete when its last kernel is.
// ------------------------------------------------------------------------------------------------------------------------
#define ARENA_U 8                       // words in flight per thread and peer: 128 blocks x 256 threads x 8 x 8 B = 2 MB in flight per rank
                                        // (a remote read takes ~2.5 us: ~800 GB/s of requests against ~450 GB/s of links)

__device__ __forceinline__ unsigned long long arena_load_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void arena_store_sys(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// announce (block 0) and wait for every rank's word of barrier `which` of this exchange; returns false when a peer is missing
__device__ __forceinline__ bool arena_barrier(const cris_p2p_arena_params& p, int which, int* s_bad) {
    if (threadIdx.x == 0) {
        cris_p2p_link L = p.link;
        L.slot = p.link.slot + which;
        const int gen = p2p_link_gen(L);
        if (blockIdx.x == 0) p2p_ll_send(L, gen, 0, 0.f);
        bool bad = false;
        (void)p2p_ll_recv_sum(L, gen, 0, bad);
        *s_bad = bad ? 1 : 0;
        if (bad && L.err) L.err[0] = 1;
    }
    __syncthreads();
    return *s_bad == 0;
}
// words [w0, w1) of the range that rank q owns
__device__ __forceinline__ void arena_slice(long words, int world, int q, long& w0, long& w1) {
    const long chunk = (words + world - 1) / world;
    w0 = min(words, (long)q * chunk);
    w1 = min(words, w0 + chunk);
}

__global__ __launch_bounds__(256) void p2p_arena_reduce_scatter_kernel(const cris_p2p_arena_params p) {
    __shared__ int s_bad;
    const bool ok = arena_barrier(p, 0, &s_bad);
    const int W = p.link.world, r = p.link.rank;
    long w0, w1;
    arena_slice(p.n >> 1, W, r, w0, w1);
    unsigned long long* mine = reinterpret_cast<unsigned long long*>(reinterpret_cast<float*>(p.arenas[r]) + p.lo);
    const long stride = (long)gridDim.x * 256;
    if (!ok) {                                       // a missing peer must not pass silently: this rank's slice becomes NaN
        for (long i = w0 + (long)blockIdx.x * 256 + threadIdx.x; i < w1; i += stride) mine[i] = 0x7fc000007fc00000ull;
        return;
    }
    for (long i0 = w0 + (long)blockIdx.x * 256 + threadIdx.x; i0 < w1; i0 += stride * ARENA_U) {
        float ax[ARENA_U], ay[ARENA_U];
#pragma unroll
        for (int u = 0; u < ARENA_U; ++u) { ax[u] = 0.f; ay[u] = 0.f; }
        for (int q = 0; q < W; ++q) {
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const float*>(p.arenas[q]) + p.lo);
            unsigned long long w[ARENA_U];
#pragma unroll
            for (int u = 0; u < ARENA_U; ++u) {
                const long i = min(i0 + u * stride, w1 - 1);
                w[u] = q == r ? mine[i] : arena_load_sys(src + i);
            }
#pragma unroll
            for (int u = 0; u < ARENA_U; ++u) {
                ax[u] += __uint_as_float((unsigned)w[u]);
                ay[u] += __uint_as_float((unsigned)(w[u] >> 32));
            }
        }
#pragma unroll
        for (int u = 0; u < ARENA_U; ++u) {
            const long i = i0 + u * stride;
            if (i < w1) arena_store_sys(mine + i, ((unsigned long long)__float_as_uint(ay[u]) << 32) | (unsigned long long)__float_as_uint(ax[u]));
        }
    }
}

__global__ __launch_bounds__(256) void p2p_arena_all_gather_kernel(const cris_p2p_arena_params p) {
    __shared__ int s_bad;
    if (!arena_barrier(p, 1, &s_bad)) return;
    const int W = p.link.world, r = p.link.rank;
    const long words = p.n >> 1;
    unsigned long long* mine = reinterpret_cast<unsigned long long*>(reinterpret_cast<float*>(p.arenas[r]) + p.lo);
    const long stride = (long)gridDim.x * 256;
    for (int d = 1; d < W; ++d) {
        const int q = (r + d) % W;                   // every rank starts with a different peer: all links busy from the start
        long w0, w1;
        arena_slice(words, W, q, w0, w1);
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const float*>(p.arenas[q]) + p.lo);
        for (long i0 = w0 + (long)blockIdx.x * 256 + threadIdx.x; i0 < w1; i0 += stride * ARENA_U) {
            unsigned long long w[ARENA_U];
#pragma unroll
            for (int u = 0; u < ARENA_U; ++u) w[u] = arena_load_sys(src + min(i0 + u * stride, w1 - 1));
#pragma unroll
            for (int u = 0; u < ARENA_U; ++u) {
                const long i = i0 + u * stride;
                if (i < w1) mine[i] = w[u];
            }
        }
    }
}

__global__ __launch_bounds__(64) void p2p_arena_done_kernel(const cris_p2p_arena_params p) {
    __shared__ int s_bad;
    (void)arena_barrier(p, 2, &s_bad);
}

extern "C" int cris_p2p_arena_allreduce(const cris_p2p_arena_params* pp, void* stream) {
    CRIS_CHECK_ARG(pp && pp->arenas && pp->link.boxes, "null operand");
    const cris_p2p_arena_params& p = *pp;
    const cris_p2p_link& l = p.link;
    CRIS_CHECK_ARG(l.world >= 1 && l.world <= 64 && l.rank >= 0 && l.rank < l.world, "rank / world");
    CRIS_CHECK_ARG(l.slot >= 0 && l.slot + 3 <= l.slots && l.max_floats >= 1, "three barrier slots out of the mailbox geometry");
    CRIS_CHECK_ARG(p.n > 0 && (p.n & 1) == 0 && p.lo >= 0 && (p.lo & 1) == 0, "range: lo and n must be even");
    // (a world of one still runs the three launches: tools/dist1_check.py and tests/test_dist_gpu.py capture them that way)
    static const int def_blocks = cris_env_int("CRIS_P2P_ARENA_BLOCKS", 128);
    int blocks = p.blocks > 0 ? p.blocks : def_blocks;
    const long per_rank = ((p.n >> 1) + l.world - 1) / l.world;
    const int need = (int)((per_rank + 256L * ARENA_U - 1) / (256L * ARENA_U));
    if (blocks > need) blocks = need > 0 ? need : 1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(p2p_arena_reduce_scatter_kernel, dim3(blocks), dim3(256), 0, s, p);
    CRIS_LAUNCH_CHECK();
    hipLaunchKernelGGL(p2p_arena_all_gather_kernel, dim3(blocks), dim3(256), 0, s, p);
    CRIS_LAUNCH_CHECK();
    hipLaunchKernelGGL(p2p_arena_done_kernel, dim3(1), dim3(64), 0, s, p);
    CRIS_LAUNCH_CHECK();
    return 0;
}
