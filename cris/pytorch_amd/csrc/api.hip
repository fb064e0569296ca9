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
extern "C" int cris_abi_version(void) { return 1; }

extern "C" int cris_sizeof(const char* name) {
#define S(n) if (!strcmp(name, #n)) return (int)sizeof(n)
    S(cris_conv_gemm_params);
    S(cris_wgrad_params);
    S(cris_pack_desc);
    S(cris_bn_apply_params);
    S(cris_bn_bwd_params);
    S(cris_ln_fwd_params);
    S(cris_ln_bwd_params);
    S(cris_attn_params);
    S(cris_adam_desc);
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

extern "C" int cris_zero_bytes(void* p, size_t nbytes, void* stream) {
    if (!p || nbytes == 0) return 0;
    hipError_t e = hipMemsetAsync(p, 0, nbytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        cris_set_error("cris_zero_bytes: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
