// Standalone probe for the 4-wave GEMM tiles (csrc/gemm.hip compiled as is with -DG4_PROBE [-DG4_ABL=mask]): HIP-event timing of one
// problem (best / median of 5 x 20 back-to-back launches, lean epilogue with BatchNorm statistics) and the per-block PHASE STAMPS
// (shader clock at entry / prologue issued / first K-step landed / K loop done / epilogue issued / stores acknowledged).
// Diagnostic only.   gemm4_probe <variant name> <Bn> <HW> <C> <N> <k> [reps]
#include "../../cris/pytorch_amd/csrc/gemm.hip"
#include <algorithm>
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
int cris_launch_gemm8(int, const cris_conv_gemm_params&, int, hipStream_t) { return -1; }                 // (gemm8.hip is not linked)
int cris_launch_gemm8_group(const cris_conv_gemm_group&, int, int, hipStream_t) { return -1; }

static unsigned short f2bf_host(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s variant Bn HW C N k [reps]\n", argv[0]); return 2; }
    int variant = -1;
    for (int v = 0; v < V_COUNT; ++v)
        if (!strcmp(argv[1], cris_conv_gemm_variant_name(v))) variant = v;
    if (variant < 0) { fprintf(stderr, "unknown variant %s\n", argv[1]); return 2; }
    const int Bn = atoi(argv[2]), HW = atoi(argv[3]), C = atoi(argv[4]), N = atoi(argv[5]), k = atoi(argv[6]);
    const int reps = argc > 7 ? atoi(argv[7]) : 20;
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
    const size_t nst = (size_t)((p.M + 31) / 32) * N;
    hipMalloc((void**)&cs, nst * 4); hipMalloc((void**)&cq, nst * 4);
    hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice);
    p.A = (const cris_bf16*)dA; p.Wt = (const cris_bf16*)dW; p.out = dO; p.colsum = cs; p.colsq = cq;
    // BatchNorm coefficients for the -DG4_AFUSE build of call r05b (A -> relu(A * scale + shift) on the operand path: the kernel
    // block that build compiled is archived as profiles/r05/call_b_afuse_probe_block.hip.txt - measured, rejected and removed from
    // csrc/gemm.hip); with G4_AFUSE_REF the plain kernel runs on the host-transformed operand: the reference of that comparison
    std::vector<float> hsc(C), hsh(C);
    for (int c = 0; c < C; ++c) { hsc[c] = 0.5f + 0.001f * (c % 97); hsh[c] = 0.1f - 0.002f * (c % 53); }
    float *dsc, *dsh;
    hipMalloc((void**)&dsc, C * 4); hipMalloc((void**)&dsh, C * 4);
    hipMemcpy(dsc, hsc.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(dsh, hsh.data(), C * 4, hipMemcpyHostToDevice);
    p.bnr_scale = dsc; p.bnr_shift = dsh;
#ifdef G4_AFUSE_REF
    {
        auto bf2f_host = [](unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; };
        for (size_t i = 0; i < na; ++i) {
            const int c = (int)(i % C);
            const float x = fmaf(bf2f_host(ha[i]), hsc[c], hsh[c]);
            ha[i] = f2bf_host(x > 0.f ? x : 0.f);
        }
        hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
    }
#endif
    static const int bm[V_COUNT] = {0, 0, 0, 128, 64, 64, 128, 256, 256, 128, 128, 64}, bn[V_COUNT] = {0, 0, 0, 64, 64, 128, 128, 256, 128, 256, 128, 64};
    const int blocks = cris_cdiv(p.M, bm[variant]) * cris_cdiv(p.N, bn[variant]);
    unsigned long long* dst = nullptr;
    hipMalloc((void**)&dst, (size_t)blocks * 8 * 8);
    hipMemset(dst, 0, (size_t)blocks * 8 * 8);
    unsigned long long* null_ptr = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g4_stamps), &null_ptr, sizeof(null_ptr));
    hipStream_t st;
    hipStreamCreate(&st);
    for (int i = 0; i < 3; ++i)
        if (cris_conv_gemm_variant(&p, variant, st) != 0) return 1;
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) cris_conv_gemm_variant(&p, variant, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms * 1e3f / reps);
    }
    std::sort(ts.begin(), ts.end());
    const double flop = 2.0 * p.M * N * p.K;
    printf("G4 %s abl=%d M%d N%d K%d k%d blocks %d : best %.2f us median %.2f us  (%.0f TFLOP/s at best)\n", argv[1], (int)G4_ABL, p.M, N, p.K, k, blocks,
           ts[0], ts[2], flop / ts[0] / 1e6);
    {   // checksum of the output (the AFUSE and AFUSE_REF builds must agree)
        std::vector<unsigned short> ho(no);
        hipMemcpy(ho.data(), dO, no * 2, hipMemcpyDeviceToHost);
        double sum = 0, asum = 0;
        for (size_t i = 0; i < no; ++i) { unsigned u = (unsigned)ho[i] << 16; float f; memcpy(&f, &u, 4); sum += f; asum += f < 0 ? -f : f; }
        printf("  output checksum: sum %.6e  abs-sum %.6e\n", sum, asum);
    }
    // ---- phase stamps of ONE launch in the middle of a back-to-back run (its neighbours keep the chip in the steady state)
    for (int i = 0; i < 4; ++i) cris_conv_gemm_variant(&p, variant, st);
    hipMemcpyToSymbolAsync(HIP_SYMBOL(g4_stamps), &dst, sizeof(dst), 0, hipMemcpyHostToDevice, st);
    cris_conv_gemm_variant(&p, variant, st);
    hipMemcpyToSymbolAsync(HIP_SYMBOL(g4_stamps), &null_ptr, sizeof(null_ptr), 0, hipMemcpyHostToDevice, st);
    for (int i = 0; i < 2; ++i) cris_conv_gemm_variant(&p, variant, st);
    hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)blocks * 8);
    hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < blocks; ++b) { tmin = std::min(tmin, h[b * 8]); tmax = std::max(tmax, h[b * 8 + 5]); }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    const char* names[6] = {"entry after first block's entry", "prologue (set-up + first DMAs issued)", "first K-step landed (wait + barrier)",
                            "K loop", "epilogue issue", "store drain (vmcnt 0)"};
    printf("  span first entry -> last store ack: %llu ticks (s_memtime; compare with the event time above for the tick length)\n", tmax - tmin);
    for (int ph = 0; ph < 6; ++ph) {
        std::vector<double> v;
        for (int b = 0; b < blocks; ++b) v.push_back(ph == 0 ? (double)(h[b * 8] - tmin) : (double)(h[b * 8 + ph] - h[b * 8 + ph - 1]));
        printf("  phase %d %-40s ticks p10 %8.0f  median %8.0f  p90 %8.0f  max %8.0f\n", ph, names[ph], pct(v, 0.1), pct(v, 0.5), pct(v, 0.9), pct(v, 1.0));
    }
    {
        std::vector<double> v, e;
        for (int b = 0; b < blocks; ++b) { v.push_back((double)(h[b * 8 + 5] - h[b * 8])); e.push_back((double)(h[b * 8 + 5] - tmin)); }
        printf("  block lifetime                                   ticks p10 %8.0f  median %8.0f  p90 %8.0f  max %8.0f ; last ack after first entry: median %8.0f\n",
               pct(v, 0.1), pct(v, 0.5), pct(v, 0.9), pct(v, 1.0), pct(e, 0.5));
        // per XCD: earliest entry and latest ack relative to the global first entry (are the XCDs' counters aligned? is one XCD late?)
        unsigned long long lo[8], hi[8];
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int x = 0; x < 8; ++x) { lo[x] = ~0ull; hi[x] = 0; }
        for (int b = 0; b < blocks; ++b) {
            const int x = (int)(h[b * 8 + 6] & 7);
            lo[x] = std::min(lo[x], h[b * 8]); hi[x] = std::max(hi[x], h[b * 8 + 5]); ++cnt[x];
        }
        printf("  per XCD (blocks: first entry .. last ack, ticks after the global first entry):");
        for (int x = 0; x < 8; ++x) if (cnt[x]) printf("  x%d %d: %lld..%lld", x, cnt[x], (long long)(lo[x] - tmin), (long long)(hi[x] - tmin));
        printf("\n");
    }
    return 0;
}
