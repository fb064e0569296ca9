// Error reporting + ABI self-description for libcris_hip.so
#include "common.h"
#include "../../../include/cris_hip.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = {0};

void cris_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cris_last_error(void) { return g_err; }
extern "C" int cris_abi_version(void) { return CRIS_ABI_VERSION; }

extern "C" int cris_sizeof(const char* name) {
#define S(n) if (!strcmp(name, #n)) return (int)sizeof(n)
    S(cris_conv_gemm_params);
    S(cris_wgrad_params);
    S(cris_conv_gemm_group);
    S(cris_wgrad_group);
    S(cris_pack_desc);
    S(cris_bn_apply_params);
    S(cris_bn_bwd_params);
    S(cris_ln_fwd_params);
    S(cris_ln_bwd_params);
    S(cris_sum_entry);
    S(cris_sum_group);
    S(cris_attn_params);
    S(cris_adam_desc);
    S(cris_p2p_params);
    S(cris_p2p_link);
    S(cris_p2p_arena_params);
    S(cris_zero_ranges);
    S(cris_sample_desc);
    S(cris_jpeg_info);
    S(cris_jpeg_image);
#undef S
    return -1;
}

extern "C" long cris_echo_conv_gemm(const void* vp) {
    const cris_conv_gemm_params* p = (const cris_conv_gemm_params*)vp;
    long h = 0;
    h = h * 31 + (long)(uintptr_t)p->A; h = h * 31 + (long)(uintptr_t)p->outT; h = h * 31 + p->T_sec_stride;
    h = h * 31 + p->lda; h = h * 31 + p->C; h = h * 31 + p->pad; h = h * 31 + p->ldb; h = h * 31 + p->K;
    h = h * 31 + p->act; h = h * 31 + p->out_f32; h = h * 31 + p->T_E; h = h * 31 + (long)(p->drop_p * 1000.f);
    h = h * 31 + p->drop_seed; h = h * 31 + p->drop_stream;
    return h;
}

// zero fill as an ordinary kernel (hipMemsetAsync measured ~85 us per call inside this step's graph)
__global__ void zero_bytes_kernel(unsigned char* p, size_t nvec, size_t nbytes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4* v = reinterpret_cast<uint4*>(p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) v[i] = make_uint4(0, 0, 0, 0);
    const size_t tail0 = nvec * 16;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < nbytes) p[tail0 + threadIdx.x] = 0;      // < 16 tail bytes
}

// several ranges in one launch (grid.y = range)
__global__ void zero_ranges_kernel(const cris_zero_ranges r) {
    const cris_zero_range z = r.r[blockIdx.y];
    const size_t nvec = z.nbytes / 16;
    uint4* v = reinterpret_cast<uint4*>(z.p);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) v[i] = make_uint4(0, 0, 0, 0);
    const size_t tail0 = nvec * 16;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < z.nbytes) reinterpret_cast<unsigned char*>(z.p)[tail0 + threadIdx.x] = 0;
}
extern "C" int cris_zero_many(const cris_zero_ranges* r, void* stream) {
    CRIS_CHECK_ARG(r && r->n > 0 && r->n <= CRIS_ZERO_RANGES_MAX, "1 .. CRIS_ZERO_RANGES_MAX ranges");
    size_t big = 0;
    for (int i = 0; i < r->n; ++i) {
        CRIS_CHECK_ARG(r->r[i].p && ((uintptr_t)r->r[i].p & 15) == 0, "ranges must be 16-byte aligned");
        if (r->r[i].nbytes > big) big = r->r[i].nbytes;
    }
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(cris_grid_1d((long)(big / 16 + 1), 256, 2048), r->n), dim3(256), 0, (hipStream_t)stream, *r);
    CRIS_LAUNCH_CHECK();
    return 0;
}

extern "C" int cris_zero_bytes(void* p, size_t nbytes, void* stream) {
    if (!p || nbytes == 0) return 0;
    CRIS_CHECK_ARG(((uintptr_t)p & 15) == 0, "buffer must be 16-byte aligned");
    const size_t nvec = nbytes / 16;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3(cris_grid_1d((long)(nvec ? nvec : 1), 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned char*)p, nvec, nbytes);
    CRIS_LAUNCH_CHECK();
    return 0;
}
