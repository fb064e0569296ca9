// The sentence-vector path in fp32 (round 4).  The per-stage error budget (tools/error_budget.py, profiles/parity_r04.md) names the
// text side as the largest contributor to the bf16 path's loss error, and inside it the rounding of the final LayerNorm output:
// the sentence vector `state` (model/clip.py:451-456) scales every pixel of a sample in the FPN (model/layers.py:286-290) and is
// the source of the projector's per-sample 3x3 kernel (model/layers.py:71-84), so one rounding of it is a COHERENT shift of all
// the logits of a sample, where the rounding of a feature map averages out over its pixels.  Everything on that path has at most
// B (= 8) rows: LayerNorm of the end-of-text rows, `@ text_projection`, neck.txt_proj (Linear + BatchNorm1d + ReLU), proj.txt
// (Linear) - a few MFLOP.  These kernels keep it in fp32 end to end on the VALU, reading the fp32 parameters directly (no bf16
// operand copies), for a handful of launches that replace the bf16 GEMM launches of the same layers one for one.
// All reductions run in a fixed order (deterministic); rows M <= CRIS_SMALL_MAX_ROWS.
#include "common.h"
#include "../../../include/cris_hip.h"

#define SM_MAXR CRIS_SMALL_MAX_ROWS

// ---- end-of-text rows through the final LayerNorm, in fp32 ---------------------------------------------------------------
// out[b][:] = (x[b*L + eot_b][:] - mean) * rstd * gamma + beta with the row statistics cris_ln_fwd saved; eot_b = first argmax of
// the token ids (model/clip.py:451-452)
__global__ void eot_gather_ln_f32_kernel(const int64_t* tokens, const float* x, const float* mean, const float* rstd, const float* gamma,
                                         const float* beta, int L, int D, float* out, int* eot_index) {
    __shared__ int s_idx;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int best = 0;
        int64_t bv = tokens[(size_t)b * L];
        for (int l = 1; l < L; ++l) {
            const int64_t v = tokens[(size_t)b * L + l];
            if (v > bv) { bv = v; best = l; }
        }
        s_idx = best;
        eot_index[b] = best;
    }
    __syncthreads();
    const size_t row = (size_t)b * L + s_idx;
    const float mu = mean[row], rs = rstd[row];
    const float* src = x + row * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[(size_t)b * D + d] = (src[d] - mu) * rs * gamma[d] + beta[d];
}
extern "C" int cris_eot_gather_ln_f32(const int64_t* tokens, const float* x, const float* mean, const float* rstd, const float* gamma,
                                      const float* beta, int Bn, int L, int D, float* out, int* eot_index, void* stream) {
    CRIS_CHECK_ARG(tokens && x && mean && rstd && gamma && beta && out && eot_index && Bn > 0, "bad args");
    hipLaunchKernelGGL(eot_gather_ln_f32_kernel, dim3(Bn), dim3(256), 0, (hipStream_t)stream, tokens, x, mean, rstd, gamma, beta, L, D, out,
                       eot_index);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// the gradient of those rows added into the bf16 gradient of the LayerNorm output (which also carries the word features' gradient)
__global__ void eot_scatter_add_f32_kernel(const int* eot_index, const float* g, int L, int D, bf16_t* dx) {
    const int b = blockIdx.x;
    bf16_t* dst = dx + ((size_t)b * L + eot_index[b]) * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dst[d] = f2bf(bf2f(dst[d]) + g[(size_t)b * D + d]);
}
extern "C" int cris_eot_scatter_add_f32(const int* eot_index, const float* drows, int Bn, int L, int D, cris_bf16* dx, void* stream) {
    CRIS_CHECK_ARG(eot_index && drows && dx && Bn > 0, "bad args");
    hipLaunchKernelGGL(eot_scatter_add_f32_kernel, dim3(Bn), dim3(256), 0, (hipStream_t)stream, eot_index, drows, L, D, dx);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ---- linear layers with at most SM_MAXR rows -------------------------------------------------------------------------------
// (a) contraction along the CONTIGUOUS dimension of the matrix: out[m][n] (+)= sum_k A[m][k] * W[n][k] (+ bias[n]).  One wave
// per output column: lanes stride k (coalesced 256-byte reads of the matrix row, the few rows of A come from the cache), M
// accumulators per lane, then a fixed-order wave reduction.  Forward of a [out][in] weight, input gradient of an [in][out] one.
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, int M, int N, int K, float* __restrict__ out, int ldo,
                                                          int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[SM_MAXR];
#pragma unroll
    for (int m = 0; m < SM_MAXR; ++m) acc[m] = 0.f;
    const float* w = W + (size_t)n * ldw;
    for (int k = lane; k < K; k += 64) {
        const float wv = w[k];
#pragma unroll
        for (int m = 0; m < SM_MAXR; ++m)
            if (m < M) acc[m] += A[(size_t)m * lda + k] * wv;
    }
#pragma unroll
    for (int m = 0; m < SM_MAXR; ++m) {
        if (m >= M) break;
        const float s = wave_sum(acc[m]);
        if (lane == 0) {
            float v = s + (bias ? bias[n] : 0.f);
            float* o = out + (size_t)m * ldo + n;
            *o = accumulate ? *o + v : v;
        }
    }
}
// (b) contraction along the STRIDED dimension: out[m][n] (+)= sum_k A[m][k] * W[k][n].  Forward of an [in][out] parameter
// (x @ text_projection), input gradient of a [out][in] one.  Block = 16 output columns x 16 k-groups: for one k the 16 columns are
// one 64-byte run of a matrix row, the 16 k-groups walk K in parallel (k = group, group + 16, ...; eight rows in flight each),
// their partial sums are added through LDS in group order (deterministic).  (A first form with one thread per column and K in
// sequence - four blocks on the chip for N 1024 - took 0.3 ms for the projector's K 2305: it sat on the critical path of backward.)
#define LC_COLS 16
#define LC_KG 16
__global__ __launch_bounds__(256) void linear_cols_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int M,
                                                          int N, int K, float* __restrict__ out, int ldo, int accumulate) {
    __shared__ float red[LC_KG][SM_MAXR][LC_COLS + 1];
    const int cl = threadIdx.x & (LC_COLS - 1), kg = threadIdx.x / LC_COLS;
    const int n = blockIdx.x * LC_COLS + cl;
    const int nc = min(n, N - 1);                                  // columns beyond N read column N-1 and store nothing
    float acc[SM_MAXR];
#pragma unroll
    for (int m = 0; m < SM_MAXR; ++m) acc[m] = 0.f;
    for (int k0 = kg; k0 < K; k0 += LC_KG * 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = W[(size_t)min(k0 + u * LC_KG, K - 1) * ldw + nc];          // eight rows in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * LC_KG;
            if (k < K) {
#pragma unroll
                for (int m = 0; m < SM_MAXR; ++m)
                    if (m < M) acc[m] += A[(size_t)m * lda + k] * wv[u];
            }
        }
    }
#pragma unroll
    for (int m = 0; m < SM_MAXR; ++m) red[kg][m][cl] = acc[m];
    __syncthreads();
    // thread (cl, kg) finishes row m = kg of its column: the k-groups' partial sums in group order
    if (kg < M && n < N) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < LC_KG; ++g) s += red[g][kg][cl];
        float* o = out + (size_t)kg * ldo + n;
        *o = accumulate ? *o + s : s;
    }
}
extern "C" int cris_linear_f32_small(const float* A, int lda, const float* W, int ldw, int w_is_kn, const float* bias, int M, int N, int K,
                                     float* out, int ldo, int accumulate, void* stream) {
    CRIS_CHECK_ARG(A && W && out && M > 0 && M <= SM_MAXR && N > 0 && K > 0, "bad args (at most CRIS_SMALL_MAX_ROWS rows)");
    CRIS_CHECK_ARG(!(w_is_kn && bias), "bias only with a [N][K] matrix");
    if (w_is_kn)
        hipLaunchKernelGGL(linear_cols_kernel, dim3(cris_cdiv(N, LC_COLS)), dim3(LC_COLS * LC_KG), 0, (hipStream_t)stream, A, lda, W, ldw, M, N, K, out,
                           ldo, accumulate);
    else
        hipLaunchKernelGGL(linear_rows_kernel, dim3(cris_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, A, lda, W, ldw, bias, M, N, K, out, ldo,
                           accumulate);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// weight gradient of either layout: G[r][c] = sum_m X1[m][r] * X2[m][c] (overwritten), optionally rowsum[r] = sum_m X1[m][r]
// ([out][in] weight: X1 = d out, X2 = input, rowsum = the bias gradient; [in][out] parameter: X1 = input, X2 = d out)
__global__ __launch_bounds__(256) void outer_sum_kernel(const float* __restrict__ X1, int ld1, const float* __restrict__ X2, int ld2, int M,
                                                        int R, int Cc, float* __restrict__ G, int ldg, float* __restrict__ rowsum) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    float a[SM_MAXR];
#pragma unroll
    for (int m = 0; m < SM_MAXR; ++m) a[m] = m < M ? X1[(size_t)m * ld1 + r] : 0.f;
    if (c < Cc) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < SM_MAXR; ++m)
            if (m < M) s += a[m] * X2[(size_t)m * ld2 + c];
        G[(size_t)r * ldg + c] = s;
    }
    if (rowsum && c == 0) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < SM_MAXR; ++m) s += a[m];
        rowsum[r] = s;
    }
}
extern "C" int cris_outer_sum_f32_small(const float* X1, int ld1, const float* X2, int ld2, int M, int R, int Cc, float* G, int ldg,
                                        float* rowsum, void* stream) {
    CRIS_CHECK_ARG(X1 && X2 && G && M > 0 && M <= SM_MAXR && R > 0 && R <= 65535 && Cc > 0, "bad args");
    hipLaunchKernelGGL(outer_sum_kernel, dim3(cris_cdiv(Cc, 256), R), dim3(256), 0, (hipStream_t)stream, X1, ld1, X2, ld2, M, R, Cc, G, ldg, rowsum);
    CRIS_LAUNCH_CHECK();
    return 0;
}

// ---- BatchNorm1d + ReLU over at most SM_MAXR rows (neck.txt_proj, model/layers.py:262-264) -------------------------------------
// statistics in the partial format of cris_bn_finalize (ONE part: column sum, M2 about the column mean) so that the coefficients -
// and with SyncBatchNorm the exchange - come from the same launch as for every other BatchNorm
__global__ void colstats_f32_small_kernel(const float* y, int ldy, int M, int C, float* psum, float* pm2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += y[(size_t)m * ldy + c];
    const float mu = s / (float)M;
    float q = 0.f;
    for (int m = 0; m < M; ++m) {
        const float d = y[(size_t)m * ldy + c] - mu;
        q += d * d;
    }
    psum[c] = s;
    pm2[c] = q;
}
extern "C" int cris_colstats_f32_small(const float* y, int ldy, int M, int C, float* psum, float* pm2, void* stream) {
    CRIS_CHECK_ARG(y && psum && pm2 && M > 0 && C > 0, "bad args");
    hipLaunchKernelGGL(colstats_f32_small_kernel, dim3(cris_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, y, ldy, M, C, psum, pm2);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// z = relu(scale * y + shift)
__global__ void bn_relu_f32_small_kernel(const float* y, int ldy, const float* scale, const float* shift, int M, int C, float* z, int ldz) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = scale[c], sh = shift[c];
    for (int m = 0; m < M; ++m) z[(size_t)m * ldz + c] = fmaxf(y[(size_t)m * ldy + c] * sc + sh, 0.f);
}
extern "C" int cris_bn_relu_f32_small(const float* y, int ldy, const float* scale, const float* shift, int M, int C, float* z, int ldz,
                                      void* stream) {
    CRIS_CHECK_ARG(y && scale && shift && z && M > 0 && C > 0, "bad args");
    hipLaunchKernelGGL(bn_relu_f32_small_kernel, dim3(cris_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, y, ldy, scale, shift, M, C, z, ldz);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// backward, first half: part[0][c] = sum_m g, part[0][C + c] = sum_m g * xhat with g = dz where scale*y + shift > 0 - ONE partial row in
// the format cris_bn_bwd_sum / cris_bn_bwd_sum_sync add up (and exchange)
__global__ void bn_relu_bwd_reduce_f32_small_kernel(const float* dz, int lddz, const float* y, int ldy, const float* scale, const float* shift,
                                                    const float* mean, const float* invstd, int M, int C, float* part) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = scale[c], sh = shift[c], mu = mean[c], iv = invstd[c];
    float s0 = 0.f, s1 = 0.f;
    for (int m = 0; m < M; ++m) {
        const float yv = y[(size_t)m * ldy + c];
        const float g = (yv * sc + sh) > 0.f ? dz[(size_t)m * lddz + c] : 0.f;
        s0 += g;
        s1 += g * ((yv - mu) * iv);
    }
    part[c] = s0;
    part[C + c] = s1;
}
extern "C" int cris_bn_relu_bwd_reduce_f32_small(const float* dz, int lddz, const float* y, int ldy, const float* scale, const float* shift,
                                                 const float* mean, const float* invstd, int M, int C, float* part, void* stream) {
    CRIS_CHECK_ARG(dz && y && scale && shift && mean && invstd && part && M > 0 && C > 0, "bad args");
    hipLaunchKernelGGL(bn_relu_bwd_reduce_f32_small_kernel, dim3(cris_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dz, lddz, y, ldy, scale,
                       shift, mean, invstd, M, C, part);
    CRIS_LAUNCH_CHECK();
    return 0;
}
// backward, second half: dy = scale * (g - sums[c] / count - xhat * sums[C + c] / count)   (sums over the GLOBAL batch)
__global__ void bn_relu_bwd_apply_f32_small_kernel(const float* dz, int lddz, const float* y, int ldy, const float* scale, const float* shift,
                                                   const float* mean, const float* invstd, const float* sums, float count, int M, int C,
                                                   float* dy, int lddy) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = scale[c], sh = shift[c], mu = mean[c], iv = invstd[c];
    const float invc = 1.0f / count;
    const float a0 = sums[c] * invc, a1 = sums[C + c];
    for (int m = 0; m < M; ++m) {
        const float yv = y[(size_t)m * ldy + c];
        const float g = (yv * sc + sh) > 0.f ? dz[(size_t)m * lddz + c] : 0.f;
        const float xh = (yv - mu) * iv;
        dy[(size_t)m * lddy + c] = sc * (g - a0 - xh * a1 * invc);
    }
}
extern "C" int cris_bn_relu_bwd_apply_f32_small(const float* dz, int lddz, const float* y, int ldy, const float* scale, const float* shift,
                                                const float* mean, const float* invstd, const float* sums, float count, int M, int C,
                                                float* dy, int lddy, void* stream) {
    CRIS_CHECK_ARG(dz && y && scale && shift && mean && invstd && sums && dy && M > 0 && C > 0 && count > 0.f, "bad args");
    hipLaunchKernelGGL(bn_relu_bwd_apply_f32_small_kernel, dim3(cris_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dz, lddz, y, ldy, scale,
                       shift, mean, invstd, sums, count, M, C, dy, lddy);
    CRIS_LAUNCH_CHECK();
    return 0;
}
