// LDS read cost of the wgrad operand fetch, per layout (gfx950).  A 128(m) x 128(col) bf16 tile per operand (2 x 32 KB), read
// as v_mfma_f32_32x32x16_bf16 fragments by 4 waves (wave tile 64 x 64: 2 A + 2 B fragments per 16-pixel slice):
//   V0  ds_read_b64_tr_b16, contiguous 512 B per wave instruction (no conflicts by construction; the reference)
//   V1  ds_read_b64_tr_b16 from the NATURAL row-major [m][128] image (what LDS-DMA of 256-B rows produces)
//   V2  same image with the 16-B chunk index XOR-ed by (row & 3) << 2 (applied on the global side of the DMA)
//   V3  same, chunk ^ ((row & 7) << 1)
//   V4  today's kernel: transposed [col][128 m] image, ds_read_b128 with the (chunk ^ row) & 15 swizzle
// Prints microseconds per variant for the same number of fragment fetches (V4 issues half as many instructions of twice
// the width).  Build: hipcc --offload-arch=gfx950 -O3 lds_tr_layout_bench.hip -o lds_tr_layout_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int V>
__device__ __forceinline__ int tr_addr(int row, int colel) {          // byte offset inside a 32 KB tile
    int chunk = colel >> 3;
    const int within = (colel & 7) * 2;
    if (V == 2) chunk ^= (row & 3) << 2;
    if (V == 3) chunk ^= (row & 7) << 1;
    return row * 256 + chunk * 16 + within;
}

template <int V>
__global__ __launch_bounds__(256) void bench(int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int t = l & 15, gb = (l >> 4) & 1, h = l >> 5;
    int acc = 0;
    if (V <= 3) {
        int base[2][2][2];                                            // [operand][fragment][q]
        for (int o = 0; o < 2; ++o)
            for (int f = 0; f < 2; ++f)
                for (int q = 0; q < 2; ++q) {
                    const int cb = (o == 0 ? wr : wc) * 2 + f;
                    if (V == 0) base[o][f][q] = o * 32768 + ((cb * 2 + q) * 512 + l * 8);
                    else base[o][f][q] = o * 32768 + tr_addr<V>(h * 8 + q * 4 + (t >> 2), cb * 32 + 16 * gb + (t & 3) * 4);
                }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                s16x4 v[8];
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            // V0: just stay inside the tile; V1..3: 16 rows further per slice (row & 7 unchanged)
                            const int off = V == 0 ? ks * 4096 : ks * 16 * 256;
                            v[o * 4 + f * 2 + q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + base[o][f][q] + off));
                        }
#pragma unroll
                for (int r = 0; r < 8; ++r) acc += v[r][0] ^ v[r][1] ^ v[r][2] ^ v[r][3];
            }
        }
    } else {
        const int fr = l & 31, fh = l >> 5;
        int rowb[2][2];
        for (int o = 0; o < 2; ++o)
            for (int f = 0; f < 2; ++f) rowb[o][f] = (o == 0 ? wr : wc) * 64 + f * 32 + fr;
        for (int it = 0; it < iters; ++it) {
            // keep the (loop-invariant) plain loads inside the loop: without this LICM hoists all 32 of them and the
            // variant measures nothing (the round-1 run printed 14 us for V4 for exactly that reason)
            asm volatile("" : "+v"(rowb[0][0]), "+v"(rowb[0][1]), "+v"(rowb[1][0]), "+v"(rowb[1][1]));
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                i32x4 v[4];
#pragma unroll
                for (int o = 0; o < 2; ++o)
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const int row = rowb[o][f];
                        const int offb = o * 32768 + row * 256 + ((((ks * 2 + fh) ^ row) & 15) << 4);
                        v[o * 2 + f] = *reinterpret_cast<const i32x4*>(smem + offb);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) acc += v[r][0] ^ v[r][1] ^ v[r][2] ^ v[r][3];
            }
        }
    }
    if (acc == 0x12345678) sink[0] = acc;
}

template <int V>
static float run(int iters, int* sink) {
    hipFuncSetAttribute((const void*)bench<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(bench<V>, dim3(512), dim3(256), 65536, 0, 16, sink);        // warm-up
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(bench<V>, dim3(512), dim3(256), 65536, 0, iters, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f;
}

int main() {
    int* sink;
    if (hipMalloc(&sink, 4) != hipSuccess) { printf("no device\n"); return 1; }
    const int iters = 2000;
    const float t0 = run<0>(iters, sink), t1 = run<1>(iters, sink), t2 = run<2>(iters, sink), t3 = run<3>(iters, sink),
                t4 = run<4>(iters, sink);
    printf("iters %d x 8 slices, 512 blocks x 256 threads, 64 KB LDS per block\n", iters);
    printf("V0 tr contiguous      %9.1f us\nV1 tr natural rows    %9.1f us\nV2 tr chunk^(r&3)<<2  %9.1f us\n"
           "V3 tr chunk^(r&7)<<1  %9.1f us\nV4 b128 transposed    %9.1f us\n", t0, t1, t2, t3, t4);
    printf("err %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
