// Standalone timing probe for the 8-wave GEMM tiles (csrc/gemm8.hip): compiles the kernel source as is, optionally with an
// ablation mask (-DG8_ABL=..., see gemm8.hip), and times one problem with HIP events.  Diagnostic only - results of ablated
// builds are wrong by construction.   tools/probe/build_gemm8_probes.sh builds one binary per mask.
//   gemm8_probe <variant 0|1|2> <Bn> <HW> <C> <N> <k> [reps]
#include "../../cris/pytorch_amd/csrc/gemm8.hip"
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

void cris_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

static unsigned short f2bf_host(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s variant Bn HW C N k [reps]\n", argv[0]); return 2; }
    const int variant = atoi(argv[1]), Bn = atoi(argv[2]), HW = atoi(argv[3]), C = atoi(argv[4]), N = atoi(argv[5]), k = atoi(argv[6]);
    const int reps = argc > 7 ? atoi(argv[7]) : 20;
    const int mode = argc > 8 ? atoi(argv[8]) : 0;         // 1: no output stores (statistics only), 2: no statistics
    cris_conv_gemm_params p;
    memset(&p, 0, sizeof(p));
    p.Bn = Bn; p.H = p.W = p.OH = p.OW = HW; p.C = C; p.KH = p.KW = k; p.stride = 1; p.pad = k / 2;
    p.M = Bn * HW * HW; p.N = N; p.K = k * k * C; p.lda = C; p.ldb = p.K; p.ldc = N;
    const size_t na = (size_t)p.M * C, nw = (size_t)N * p.K, no = (size_t)p.M * N;
    std::vector<unsigned short> ha(na), hw(nw);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : ha) v = f2bf_host(rnd());
    for (auto& v : hw) v = f2bf_host(rnd() * 0.05f);
    void *dA, *dW, *dO;
    float *cs, *cq;
    hipMalloc(&dA, na * 2); hipMalloc(&dW, nw * 2); hipMalloc(&dO, no * 2);
    const int rows = variant >= 2 ? 64 : 128;
    const size_t nst = (size_t)((p.M + rows - 1) / rows) * N;
    hipMalloc((void**)&cs, nst * 4); hipMalloc((void**)&cq, nst * 4);
    hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice);
    p.A = (const cris_bf16*)dA; p.Wt = (const cris_bf16*)dW; p.out = dO; p.colsum = cs; p.colsq = cq;
    if (mode == 1) p.out = nullptr;
    if (mode == 2) p.colsum = p.colsq = nullptr;
    hipStream_t st;
    hipStreamCreate(&st);
    for (int i = 0; i < 3; ++i)
        if (cris_launch_gemm8(variant, p, 1, st) != 0) return 1;
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, tot = 0.f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) cris_launch_gemm8(variant, p, 1, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        tot += ms;
    }
    std::vector<unsigned short> ho(no);
    hipMemcpy(ho.data(), dO, no * 2, hipMemcpyDeviceToHost);
    unsigned long long ck = 0;
    for (size_t i = 0; i < no; ++i) ck = ck * 1099511628211ull + ho[i];
    const double us = best * 1e3 / reps, fl = 2.0 * p.M * N * p.K;
    printf("G8PROBE mode=%d abl=%d variant=%d M=%d N=%d K=%d k=%d : %.1f us  %.0f TFLOP/s (best of 5 x %d launches; mean %.1f us) out checksum %016llx\n", mode, G8_ABL,
           variant, p.M, N, p.K, k, us, fl / us / 1e6, reps, tot / 5 * 1e3 / reps, ck);
    return 0;
}
