// Ground truth for ds_read_b64_tr_b16 on gfx950: which LDS elements does lane l receive when every lane supplies the
// address of 4 contiguous 16-bit elements?  LDS holds lds[i] = i; case A: 16-lane group g reads the contiguous [4][16]
// block g (lane t -> elements g*64 + t*4 ..); case B: the natural row-major [m][128] tile (row stride 128 elements),
// lane t of a group -> row (t>>2), columns (t&3)*4.. of the block at (rows 4*(g>>1).., cols 16*(g&1)..).
// Build: hipcc --offload-arch=gfx950 -O2 ds_read_tr_probe.hip -o ds_read_tr_probe ; prints elem indices per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__global__ void probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, t = l & 15;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + g * 64 + t * 4));
    const int row = 4 * (g >> 1) + (t >> 2), col = 16 * (g & 1) + (t & 3) * 4;
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + row * 128 + col));
    for (int j = 0; j < 4; ++j) {
        out[l * 8 + j] = a[j];
        out[l * 8 + 4 + j] = b[j];
    }
}

int main() {
    short* d;
    short h[64 * 8];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 1; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d  A: %4d %4d %4d %4d   B(row*128+col): ", l, h[l * 8], h[l * 8 + 1], h[l * 8 + 2], h[l * 8 + 3]);
        for (int j = 0; j < 4; ++j) printf("(%d,%d) ", h[l * 8 + 4 + j] / 128, h[l * 8 + 4 + j] % 128);
        printf("\n");
    }
    return 0;
}
