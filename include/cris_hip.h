/* libcris_hip.so - C ABI of the MI355X-native CRIS training path (gfx950 only).
 *
 * The reference (DerrickWang005/CRIS.pytorch) has NO native / FFI interface: every kernel it runs is a
 * stock torch/cuDNN/cuBLAS kernel reached through torch.nn.  Each entry point below
 * therefore replaces the torch operator(s) named in its comment, cited as reference file:line of the
 * call site.  Conventions:
 *   - plain pointers + sizes, no torch types; every buffer is owned by the caller (device memory from
 *     torch's caching allocator) and no reference is kept past the call; the library never allocates or
 *     frees device memory;
 *   - every launcher takes the hipStream_t to launch on (as void*), is re-entrant, keeps no global mutable
 *     state (forward runs on the Python thread, backward on autograd's worker thread);
 *   - return 0 on success, non-zero on error; the message is available from cris_last_error()
 *     (thread local).  No exceptions cross the ABI.
 *   - activations are NHWC / token-major bf16 (raw uint16), parameters and statistics fp32.
 */
#ifndef CRIS_HIP_H
#define CRIS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t cris_bf16;

const char* cris_last_error(void);
/* CRIS_ABI_VERSION moves whenever an exported signature or struct changes; a binding compares cris_abi_version() with the
 * value it was written against and refuses a library of another version (a stale build loaded with new argument lists would
 * mis-read them silently).  2: cris_step_advance took its fourth argument (round 5); 3, 4: round 6 (arena exchange; row strides of the weight packs) */
#define CRIS_ABI_VERSION 4
int cris_abi_version(void);
/* sizeof() of the parameter structs, so the Python mirror (ctypes) can be checked without a GPU */
int cris_sizeof(const char* struct_name);
/* Host-only self check of struct layout: returns a checksum of fields read through the C struct. */
long cris_echo_conv_gemm(const void* conv_gemm_params);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear, bf16 MFMA (v_mfma_f32_32x32x16_bf16; 16x16x32 in the M <= 144 kernel), fp32 accumulate.
 *   out[m, n] = epilogue( sum_k A_im2col[m, k] * Wt[n, k] )
 *   m = (b, oh, ow) over an NHWC input [Bn, H, W, lda] (channels a_coff .. a_coff+C),  k = tap*C + c.
 * Replaces: nn.Conv2d forward everywhere (reference model/clip.py:17-42,77,165-182; model/layers.py:10,58),
 * nn.Linear / MHA in-proj / out-proj (model/clip.py:246-251; model/layers.py:15,61,202-212), `@ text_projection`
 * (model/clip.py:451-452); with flipped/transposed weight packs also their input-gradient (dgrad).
 * Linear layers use Bn=M, H=W=OH=OW=KH=KW=1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const cris_bf16* A;      /* NHWC input */
    const cris_bf16* Wt;     /* [N][ldb], k contiguous */
    const float* bias;       /* [N] or NULL */
    const void* resid;       /* [M][ldr] (+r_coff) added after activation/dropout, bf16 or fp32; or NULL */
    void* out;               /* [M][ldc] (+c_coff) bf16 or fp32; or NULL */
    cris_bf16* outT;         /* optional head-split transposed copy, see T_* ; or NULL */
    float* colsum;           /* optional BatchNorm statistics partials [nparts][N]: per row-block column sums ...  */
    float* colsq;            /* ... and sums of squared deviations from the block mean; nparts = ceil(M / R) blocks of
                                R = cris_conv_gemm_stat_rows(p) rows (deterministic, no atomics) */
    long T_sec_stride;       /* elements between consecutive T_E-wide column sections in outT */
    int lda, a_coff;
    int Bn, H, W, C;
    int OH, OW, KH, KW, stride, pad;
    int ldb;
    int M, N, K;
    int act;                 /* 0 none, 1 relu, 2 QuickGELU x*sigmoid(1.702x) (model/clip.py:234-236); 3 relu applied AFTER the
                              * residual add (Bottleneck tail `out += identity; relu` model/clip.py:55-56 with the BatchNorm folded) */
    int ldr, r_coff, resid_f32;
    int ldc, c_coff, out_f32;
    int T_L, T_Lpad, T_E;    /* outT[sec][(b*(T_E/64)+h)*64+d][l], m = b*T_L + l, n = sec*T_E + h*64 + d */
    float drop_p;            /* dropout on (acc+bias, act) before the residual add; survivors scaled 1/(1-p) */
    uint32_t drop_thresh;    /* keep iff hash >= drop_thresh; host computes min(int(p*2^32), 2^32-1); 0 = off */
    uint32_t drop_seed, drop_stream;
    const uint32_t* drop_seed_dev; /* optional device word added to drop_seed (per-step seed of a replayed HIP graph) */
    float* ws;               /* workspace of cris_conv_gemm_ws_floats(p, variant) floats (the split-K skinny variant; else unused) */
    /* BatchNorm-BACKWARD partials instead of the forward statistics (input-gradient GEMMs; tile variants only): when bnr_y is set,
     * the output of this GEMM is dz, the gradient of z = relu(bn(y)) (model/clip.py:47-50 `relu(bn(conv(.)))` whose only consumer is
     * the convolution this GEMM differentiates), and colsum / colsq receive per row block (sum g, sum g * xhat) with
     * g = dz where scale*y + shift > 0 else 0, xhat = (y - mean) * invstd - what cris_bn_bwd_reduce would compute in a launch of
     * its own.  stat_ld: row stride of the colsum / colsq tables in floats (0 = N), so that both can live in ONE [parts][2N]
     * table (colsq = colsum + N) that cris_bn_bwd_sum adds up. */
    const cris_bf16* bnr_y;
    const float* bnr_mean; const float* bnr_invstd; const float* bnr_scale; const float* bnr_shift;
    int bnr_ldy, bnr_coff;
    int stat_ld, pad_;
} cris_conv_gemm_params;
/* Limits checked by every launcher: channels / leading dimensions / offsets multiples of 8, operand extents below 2 GiB (32-bit
 * buffer offsets), and - since the prologue's index arithmetic works by reciprocal multiplication (round 5) - fewer than 2^24 output
 * rows and 2^22 tiles (CRIS-R50 at 416x416: 346112 rows per 8 samples, i.e. up to batch 384 per launch). */
int cris_conv_gemm(const cris_conv_gemm_params* p, void* stream);
/* rows per BatchNorm-statistics partial written for this problem (depends on the tile variant chosen; host only) */
int cris_conv_gemm_stat_rows(const cris_conv_gemm_params* p);
/* The same launch with the tile variant named by the caller instead of chosen from the problem size (tests and the
 * per-shape tuning tool tools/gemm_variants.py; `variant` < 0 = automatic = cris_conv_gemm).  Variants: the 4-wave tiles
 * 128x64 / 64x64 / 64x128 / 128x128, the two skinny kernels, the 8-wave ping-pong tiles 256x256 / 256x128 / 128x256
 * (csrc/gemm8.hip; C % 64 == 0 only).  Returns -1 when the variant cannot run the problem.  Results of different variants
 * differ only by fp32 summation order inside a K-tile (none: every variant adds k in the same order) - the outputs are
 * bit-identical; the BatchNorm partials differ in their row grouping (cris_conv_gemm_variant_stat_rows). */
int cris_conv_gemm_variant(const cris_conv_gemm_params* p, int variant, void* stream);
int cris_conv_gemm_variant_stat_rows(const cris_conv_gemm_params* p, int variant);
/* floats of workspace p->ws the launch of this problem with this variant needs (0 for every variant but the split-K skinny one) */
long cris_conv_gemm_ws_floats(const cris_conv_gemm_params* p, int variant);
int cris_conv_gemm_num_variants(void);
const char* cris_conv_gemm_variant_name(int variant);
/* what cris_conv_gemm_variant(p, variant) would run: the resolved tile variant (>= 0; -1 when `variant` cannot run the problem)
 * and, in *epilogue, the epilogue instantiation (0 general, 1 lean, 2 lean + bias / ReLU); host only */
int cris_conv_gemm_plan(const cris_conv_gemm_params* p, int variant, int* epilogue);

/* Up to CRIS_GEMM_GROUP_MAX INDEPENDENT forward / input-gradient problems in ONE launch (problem table passed by value in the
 * kernel arguments; reference call sites: the q / k / v projections of nn.MultiheadAttention model/layers.py:202-207,235-243 and
 * AttentionPool2d model/clip.py:112-139, the FPN's f4_proj3/4/5 model/layers.py:300-302, a Bottleneck's conv1 + downsample
 * model/clip.py:44-53 - and their input-gradient twins).  No problem may read or accumulate into what another one writes.  All
 * problems run the tile `variant` (one of the 4-wave tiles 128x64 / 64x64 / 64x128 / 128x128 or the 8-wave 128x128 tile,
 * which needs C % 64 == 0 everywhere) and must share the epilogue instantiation (cris_conv_gemm_plan).  Results are bit-identical to
 * cris_conv_gemm_variant(prob[i], variant) one by one.  block_start is filled by the launcher. */
#define CRIS_GEMM_GROUP_MAX 12
typedef struct {
    int n;                                           /* problems in prob[] */
    int block_start[CRIS_GEMM_GROUP_MAX + 1];        /* (out) first block of each problem */
    cris_conv_gemm_params prob[CRIS_GEMM_GROUP_MAX];
} cris_conv_gemm_group;
int cris_conv_gemm_group_launch(const cris_conv_gemm_group* g, int variant, void* stream);

/* Weight gradient in the GEMM layout: dW[n][tap*C + c] = sum_m dY[m, n] * X_im2col[m, tap*C + c]  (csrc/wgrad.hip).
 * Deterministic, no atomics: splits == 1 stores the whole reduction; splits > 1 stores one partial tile per split into the
 * workspace `ws` (cris_wgrad_ws_floats floats) and sums the slabs in split order (cris_wgrad_reduce, launched by
 * cris_conv_wgrad itself).  Replaces convolution_backward(weight) / addmm backward for every Conv2d / Linear above.
 * cris_adam_step reads this layout directly; cris_unpack_grads converts it to the parameter layout [n][c][tap]. */
typedef struct {
    const cris_bf16* dY;     /* [M][ldy] (+y_coff) */
    const cris_bf16* X;      /* NHWC input of the forward conv */
    float* dW;               /* fp32 [N][ldw], k = tap*C + c contiguous; overwritten */
    int ldy, y_coff, N_ld;   /* N_ld: columns of dY that may be read (multiple of 8, >= N) */
    int ldx, x_coff;
    int Bn, H, W, C;
    int OH, OW, KH, KW, stride, pad;
    int M, N, K;
    int ldw;                 /* row stride of dW (>= K) */
    int splits;              /* requested split of the pixel range (each split covers ceil(M/splits) rows rounded up to 128;
                                the launcher drops splits that would be empty) */
    int tile;                /* output tile: 0 = the library's choice (cris_conv_wgrad_tile), 128 = 4-wave kernel, 256 = 8-wave */
    int defer_reduce;        /* 1: cris_conv_wgrad leaves the split slabs in `ws`; the caller adds them up later with
                                cris_wgrad_reduce or, for several problems in one launch, cris_wgrad_reduce_group */
    float* dbias;            /* optional [N]: = column sums of dY (bias gradient); or NULL */
    float* ws;               /* splits > 1: workspace of cris_wgrad_ws_floats(M, N, ldw, splits) floats, 16-byte aligned */
} cris_wgrad_params;
int cris_conv_wgrad(const cris_wgrad_params* p, void* stream);
/* output tile (128: 4-wave kernel, 256: 8-wave kernel) the launchers use for this problem; host only.  A caller that sizes the
 * pixel split from the tile count asks this first; the problems of one cris_conv_wgrad_group launch must share the tile. */
int cris_conv_wgrad_tile(const cris_wgrad_params* p);
long cris_wgrad_ws_floats(int M, int N, int ldw, int splits);
int cris_wgrad_reduce(const cris_wgrad_params* p, void* stream);

/* Up to CRIS_WGRAD_GROUP_MAX weight-gradient problems in ONE launch (problem table passed by value in the kernel
 * arguments): the mid-size layers have too few 128x128 output tiles to fill the chip alone; the engine queues them per
 * gradient-arena stage and launches them together, longest pixel reductions first.  block_start is filled by the launcher. */
#define CRIS_WGRAD_GROUP_MAX 24
typedef struct {
    int n;                                           /* problems in prob[] */
    int block_start[CRIS_WGRAD_GROUP_MAX + 1];       /* (out) first block of each problem */
    cris_wgrad_params prob[CRIS_WGRAD_GROUP_MAX];
} cris_wgrad_group;
int cris_conv_wgrad_group(const cris_wgrad_group* g, void* stream);
/* the split reductions (cris_wgrad_reduce) of up to CRIS_WGRAD_GROUP_MAX problems launched with defer_reduce, in ONE launch;
 * problems with splits == 1 are skipped (the native trainer's default since round 4: ops.WgradQueue). */
int cris_wgrad_reduce_group(const cris_wgrad_group* g, void* stream);

/* Batched weight packing (fp32 parameter layout -> bf16 GEMM layouts), one launch for a table of tensors.
 *   F layout: Wf[n][tap][Cpad]        (forward, k = tap*Cpad + c, zero padded)
 *   D layout: Wd[c][taps-1-tap][Npad] (dgrad: conv of dY with flipped taps; for linear = transpose) */
typedef struct {
    const float* src;        /* [N][Cin][taps]  (or [Cin][N] when src_transposed, taps == 1) */
    cris_bf16* dstF;         /* or NULL */
    cris_bf16* dstD;         /* or NULL */
    const float* row_scale;  /* or NULL; [N]: output row n is multiplied by row_scale[n] before rounding - an inference-time
                              * BatchNorm folded into its convolution (gamma / sqrt(running_var + eps), cris_bn_eval_coeffs) */
    int N, Cin, taps, Cpad, Npad, src_transposed;
    int block_start;         /* first block of this tensor in the launch grid (prefix sum) */
    int ldF;                 /* row stride of dstF in elements, 0 = dense (taps * Cpad); the GEMMs take it as ldb.  Round 6, opt-in
                              * (CRIS_PACK_SKEW=1): a row stride of an ODD number of 128-byte lines (taps * Cpad + 64 when that is a
                              * multiple of 128 elements) spreads a resident panel's rows over all L2 channels: the K = 4608 convolutions
                              * 5 - 10 % faster standalone, nothing in the step (profiles/r06/stride_skew_probe.log, r06_ab_experiments.md) */
    int ldD;                 /* the same for dstD, 0 = dense (taps * Npad) */
    int pad_;
} cris_pack_desc;
int cris_pack_weights(const cris_pack_desc* dev_table, int n_desc, int total_blocks, void* stream);
/* number of blocks a (host-side) descriptor needs: used to build block_start prefix sums on the host */
int cris_pack_blocks(const cris_pack_desc* host_desc);
int cris_pack_block_elems(void);


/* ------------------------------------------------------------------------------------------------
 * BatchNorm (training statistics), replaces native_batch_norm / its backward
 * (reference: every nn.BatchNorm2d/1d, model/clip.py:18-41,78,173-183; model/layers.py:11,16,262).
 * Statistics arrive as per-row-block partials (column sum, M2 about the block mean) from the conv GEMM
 * epilogue or cris_colstats_bf16 and are merged exactly: mean = sum_i S_i / n, M2 = sum_i [M2_i + n_i (S_i / n_i - mean)^2] (the
 * decomposition of the sum of squares about the common mean; two passes over the list, no E[x^2] - mean^2 cancellation).
 * SyncBN: call once with `merged` (local sum / M2 / mean out), exchange (cris_bn_sync_pack + ONE all-reduce +
 * cris_bn_sync_unpack), then call again with `global_stats`.
 * Lists of up to 512 parts take one launch (16 or, for more than 128 parts, 64 merge lanes per channel); longer ones are first
 * merged into 64 slices by a second launch: psum / pm2 must have room for cris_bn_partials_rows(nparts) rows of C floats each
 * (the slices are written behind the nparts partial rows).  nparts must be ceil(count_local / rows_per_part).
 * ---------------------------------------------------------------------------------------------- */
int cris_bn_partials_rows(int nparts);
int cris_bn_finalize(const float* psum, const float* pm2, int nparts, int rows_per_part, float count_local, float count,
                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, int C, float* scale, float* shift, float* mean, float* invstd, float* merged,
                     const float* global_stats, void* stream);
/* single-exchange SyncBN: pack the local (sum | M2) [2C] into moments about `ref` (identical on every rank, e.g. the
 * running mean), all-reduce the 2C floats once, unpack into (global sum | M2 about the global mean) for
 * cris_bn_finalize(global_stats=...) */
int cris_bn_sync_pack(float* merged, const float* mean_local, const float* ref, float n_local, int C, void* stream);
int cris_bn_sync_unpack(float* merged, const float* ref, float count_global, int C, void* stream);
int cris_colstats_bf16(const cris_bf16* x, int ldx, int coff, int M, int C, int rows_per_part, float* psum, float* pm2,
                       void* stream);
/* eval mode: scale/shift from the running statistics */
int cris_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, void* stream);

typedef struct {
    const cris_bf16* y;  int ldy, y_coff;           /* conv output */
    const float* scale;  const float* shift;         /* [C] */
    const cris_bf16* y2; int ldy2, y2_coff;          /* optional second BN branch (downsample) or NULL */
    const float* scale2; const float* shift2;
    const cris_bf16* ident; int ldi, i_coff;         /* optional identity added before the ReLU, or NULL */
    const float* mul;                                /* optional [Bn][C] multiplier applied after the ReLU */
    cris_bf16* z;        int ldz, z_coff;            /* output */
    int Bn, H, W, C;                                 /* input geometry, M = Bn*H*W rows */
    int relu;
    int pool;                                        /* 1: 2x2/s2 average pool after the ReLU (H,W even) */
} cris_bn_apply_params;
int cris_bn_apply(const cris_bn_apply_params* p, void* stream);

typedef struct {
    const cris_bf16* dz; int lddz, dz_coff;          /* grad wrt output z (pooled resolution when pool) */
    const cris_bf16* z;  int ldz, z_coff;            /* forward output: ReLU mask when ident/y2 are used (else recomputed) */
    const cris_bf16* y;  int ldy, y_coff;
    const float* scale; const float* shift; const float* mean; const float* invstd;
    const cris_bf16* y2; int ldy2, y2_coff;          /* second branch or NULL */
    const float* mean2; const float* invstd2; const float* scale2;
    const float* mul;                                /* [Bn][C] or NULL (forward multiplier) */
    float* sums;                                     /* [4*C]: sum g, sum g*xhat, (branch 2) sum g, sum g*xhat2 (+=) */
    float* part;                                     /* reduce workspace, cris_bn_bwd_ws_floats(p) floats: one partial row of
                                                        sums per row block (<= 64); summed in block order (no atomics) */
    float* dmul;                                     /* [Bn][C] grad of mul or NULL */
    cris_bf16* dy;  int lddy, dy_coff;               /* grad wrt y */
    cris_bf16* dy2; int lddy2, dy2_coff;             /* grad wrt y2 or NULL */
    cris_bf16* dident; int lddi, di_coff;            /* grad wrt identity (= masked dz) or NULL */
    int dident_accum;                                /* 1: dident += */
    int Bn, H, W, C;
    int relu, pool;
    float count;                                     /* rows entering the statistics (global count under SyncBN) */
} cris_bn_bwd_params;
int cris_bn_bwd_reduce(const cris_bn_bwd_params* p, void* stream);
/* the second half of cris_bn_bwd_reduce alone: p->part already holds `nparts` partial rows [2C] (written by the epilogue of the
 * input-gradient GEMM that produced dz: cris_conv_gemm_params.bnr_y); adds them up in row order into p->sums (+=) */
int cris_bn_bwd_sum(const cris_bn_bwd_params* p, int nparts, void* stream);
long cris_bn_bwd_ws_floats(const cris_bn_bwd_params* p);
int cris_bn_bwd_apply(const cris_bn_bwd_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (replaces native_layer_norm fwd/bwd; reference model/clip.py:226-231,247,252,381;
 * model/layers.py:103,199-200,211,214-216) with the surrounding elementwise work fused.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const void* x; int x_f32; int ldx;               /* input rows */
    const float* gamma; const float* beta;
    const float* pos; int pos_rows;                  /* optional [pos_rows][C] table, row index = row % pos_rows */
    const float* resid;                              /* optional fp32 [rows][C]: out_f32 = resid + dropout(LN(x)) */
    cris_bf16* y;                                    /* optional bf16 LN(x) */
    cris_bf16* ypos;                                 /* optional bf16 LN(x) + pos */
    float* out_f32;                                  /* optional fp32 (see resid) */
    float* mean; float* rstd;                        /* [rows] saved for backward */
    int rows, C;
    int in_relu;                                     /* apply ReLU to x first (bf16 input holds pre-activation) */
    float in_drop_p; uint32_t in_thresh, in_seed, in_stream;     /* dropout on the input (FFN: LN(dropout(relu(h)))) */
    float out_drop_p; uint32_t out_thresh, out_seed, out_stream; /* dropout on LN(x) before the residual add */
    float eps;
    const uint32_t* seed_dev;                        /* optional device word added to in_seed / out_seed */
} cris_ln_fwd_params;
int cris_ln_fwd(const cris_ln_fwd_params* p, void* stream);

typedef struct {
    const void* x; int x_f32; int ldx;
    const float* gamma;
    const float* mean; const float* rstd;
    const cris_bf16* dy;                             /* optional grad wrt y */
    const cris_bf16* dypos;                          /* optional grad wrt ypos (added to dy) */
    const float* dout_f32;                           /* optional grad wrt out_f32 (passes the output dropout) */
    float* part;                                     /* (out) [cris_ln_bwd_parts(rows)][2C]: per-block (dgamma | dbeta) partial rows;
                                                        cris_sum_tables adds them in block order into the gradients */
    void* dx; int dx_f32; int dx_accum;              /* grad wrt x: bf16 or fp32; accum: += (fp32 only) */
    int rows, C;
    int in_relu;
    float in_drop_p; uint32_t in_thresh, in_seed, in_stream;
    float out_drop_p; uint32_t out_thresh, out_seed, out_stream;
    const uint32_t* seed_dev;
} cris_ln_bwd_params;
int cris_ln_bwd(const cris_ln_bwd_params* p, void* stream);
int cris_ln_bwd_parts(int rows);                     /* rows of the partials table for `rows` LayerNorm rows */

/* Ordered column sums of up to CRIS_SUM_GROUP_MAX partial tables in one launch (table passed by value):
 * out[c] = sum_p part[p*ld + c], p = 0 .. nparts-1 in order - the deterministic replacement of atomic accumulation for
 * the LayerNorm parameter gradients (the engine flushes one group per gradient-arena stage).  block_start: launcher. */
#define CRIS_SUM_GROUP_MAX 64
typedef struct { const float* part; float* out; int nparts, ncol, ld, pad_; } cris_sum_entry;
typedef struct {
    int n;
    int block_start[CRIS_SUM_GROUP_MAX + 1];
    cris_sum_entry e[CRIS_SUM_GROUP_MAX];
} cris_sum_group;
int cris_sum_tables(const cris_sum_group* g, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-head attention, head dim 64 (replaces bmm/baddbmm/_softmax/bernoulli_/bmm of the
 * need_weights=True torch path, reference model/layers.py:235,240-243, and the SDPA calls of
 * model/clip.py:119-139,259-260).  q is scaled by `scale` before QK^T; optional causal mask,
 * key-padding mask (int64 token ids == 0 are padding: model/segmenter.py:37) and dropout on the
 * probabilities.  Operands are read straight from L2 in MFMA fragment order: Q/K/V token-major
 * [B*L][ld] (head h at column h*64) plus head-split transposed copies Kt/Vt/Qt/dOt
 * [(b*H+h)*64+d][Lpad] written by the producing GEMM's epilogue (zero padded beyond L).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const cris_bf16* Q;  int ldq;                    /* [B*Lq][ldq] */
    const cris_bf16* K;  int ldk;                    /* [B*Lk][ldk] */
    const cris_bf16* V;  int ldv;
    const cris_bf16* Vt; const cris_bf16* Kt; const cris_bf16* Qt; int Lk_pad, Lq_pad;
    const int64_t* key_tokens;                       /* [B][Lk] or NULL; token == 0 -> key masked */
    cris_bf16* O;  int ldo;                          /* [B*Lq][ldo] */
    float* lse;                                      /* [B*H][Lq] log-sum-exp of scaled scores */
    /* backward */
    const cris_bf16* dO; int lddo; const cris_bf16* dOt;   /* dOt [(b*H+h)*64+d][Lq_pad] */
    float* delta;                                    /* [B*H][Lq] rowsum(dO*O), written by bwd_dq */
    cris_bf16* dQ; int lddq;
    cris_bf16* dK; int lddk;
    cris_bf16* dV; int lddv;
    int B, Hn, Lq, Lk;
    int causal;
    float scale;
    float drop_p; uint32_t drop_thresh, drop_seed, drop_stream;
    const uint32_t* drop_seed_dev;                   /* optional device word added to drop_seed */
} cris_attn_params;
int cris_attn_fwd(const cris_attn_params* p, void* stream);
int cris_attn_bwd_dq(const cris_attn_params* p, void* stream);
int cris_attn_bwd_dkv(const cris_attn_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / data movement kernels
 * ---------------------------------------------------------------------------------------------- */
/* stem: fp32 NCHW image -> bf16 im2col rows [B*OH*OW][32] for the 3x3/s2/p1 conv (k = ci*9 + kh*3 + kw, 27..31 zero)
 * (reference model/clip.py:165-170,215: x.type(dtype) + conv1) */
int cris_stem_im2col(const float* img, int Bn, int H, int W, cris_bf16* out, void* stream);
/* 2x2/s2 average pool, NHWC bf16 (avg_pool2d: model/clip.py:23,35,184; model/layers.py:297) */
int cris_avgpool2_fwd(const cris_bf16* x, int ldx, int xcoff, int Bn, int H, int W, int C, cris_bf16* y, int ldy,
                      int ycoff, void* stream);
int cris_avgpool2_bwd(const cris_bf16* dy, int lddy, int dycoff, int Bn, int H, int W, int C, cris_bf16* dx, int lddx,
                      int dxcoff, int accum, void* stream);
/* x2 bilinear upsample, align_corners=False (F.interpolate / nn.Upsample: model/layers.py:54,56,293,304) */
int cris_upsample2_fwd(const cris_bf16* x, int ldx, int xcoff, int Bn, int H, int W, int C, cris_bf16* y, int ldy,
                       int ycoff, void* stream);
int cris_upsample2_bwd(const cris_bf16* dy, int lddy, int dycoff, int Bn, int H, int W, int C, cris_bf16* dx,
                       int lddx, int dxcoff, int accum, void* stream);
/* CoordConv coordinate channels (model/layers.py:30-39): x at column coff, y at coff+1, zeros up to coff+nfill */
int cris_fill_coords(cris_bf16* x, int ldx, int coff, int nfill, int Bn, int H, int W, void* stream);
/* generic strided bf16 ops: y (=|+=) a [+ b] ;  and y = a + f32 table[row % trows] */
int cris_add_bf16(const cris_bf16* a, int lda, int acoff, const cris_bf16* b, int ldb, int bcoff, cris_bf16* y,
                  int ldy, int ycoff, int M, int C, void* stream);
int cris_add_rowtable(const cris_bf16* a, int lda, const float* table, int trows, cris_bf16* y, int ldy, int M,
                      int C, void* stream);
int cris_cast_f32_bf16(const float* x, cris_bf16* y, long n, void* stream);
int cris_cast_bf16_f32(const cris_bf16* x, float* y, long n, int accum, void* stream);
/* y = bf16(dropout(x)) over a flat fp32 tensor (gradient of nn.Dropout on the residual branches, model/layers.py:217-219) */
int cris_cast_f32_bf16_drop(const float* x, cris_bf16* y, long n, float drop_p, uint32_t drop_thresh, uint32_t seed,
                            uint32_t stream_id, const uint32_t* seed_dev, void* stream);
/* device-side per-step state of a replayed HIP graph: seed[0] = step[0] * 7919 + 17 ; step[0] += 1 ; exchange_gen[0] += 1
 * (exchange_gen may be NULL: the generation counter of the peer-mailbox exchanges, never rewound - see csrc/p2p_ll.h) */
int cris_step_advance(int32_t* step, uint32_t* seed, int32_t* exchange_gen, void* stream);
int cris_axpy_f32(float* dst, const float* src, float alpha, long n, void* stream);
/* QuickGELU x*sigmoid(1.702x) on a stored bf16 pre-activation (model/clip.py:234-236) */
int cris_quickgelu_fwd(const cris_bf16* x, cris_bf16* y, long n, void* stream);
int cris_quickgelu_bwd(const cris_bf16* x, const cris_bf16* dy, cris_bf16* dx, long n, void* stream);
/* token embedding + positional embedding (model/clip.py:440-443) and its backward (embedding_dense_backward) */
int cris_embed_fwd(const int64_t* tokens, const float* table, const float* pos, int Bn, int L, int D, float* out,
                   void* stream);
/* row_live (optional, [vocabulary] bytes, sticky): set to 1 for every token of the batch - the rows of the table that have ever
 * received a gradient (see cris_adam_desc.row_live) */
int cris_embed_bwd(const int64_t* tokens, const float* dx, int Bn, int L, int D, float* dtable, float* dpos, unsigned char* row_live,
                   void* stream);
/* rows x[b*L + argmax_l tokens[b,:]] -> bf16 (EOT feature select, model/clip.py:451-452; first max wins) */
int cris_eot_gather(const int64_t* tokens, const cris_bf16* x, int Bn, int L, int D, cris_bf16* out, int* eot_index,
                    void* stream);
int cris_eot_scatter_add(const int* eot_index, const cris_bf16* dstate_rows, int Bn, int L, int D, cris_bf16* dx,
                         void* stream);
/* posr[t][c] = sum_j R[t][j] * pos[1+j][c] (bicubic resize of the attnpool positional embedding as a
 * constant linear map, model/clip.py:80-108) and its transpose for the gradient */
int cris_posresize_fwd(const float* R, const float* pos, int T, int G, int C, float* posr, void* stream);
int cris_posresize_bwd(const float* R, const float* dposr, int T, int G, int C, float* dpos, void* stream);
/* dposr[t][c] = sum_b dx[b*T+t][c] */
int cris_batch_rowsum(const cris_bf16* dx, int ldx, int Bn, int T, int C, float* out, void* stream);
/* text-to-pixel dynamic conv (grouped conv2d, groups = B: model/layers.py:71-84), fp32 logits out */
int cris_dynconv_fwd(const cris_bf16* x, int Bn, int H, int W, int C, const float* wb, int ldwb, float* pred,
                     void* stream);
/* dx, and dwb[b][i] += (per-sample kernel + bias gradient); ws: cris_dynconv_bwd_ws_floats floats (per-block partial rows,
 * summed in block order - no atomics) */
int cris_dynconv_bwd(const cris_bf16* x, const float* dpred, int Bn, int H, int W, int C, const float* wb, int ldwb,
                     cris_bf16* dx, float* dwb, float* ws, void* stream);
long cris_dynconv_bwd_ws_floats(int Bn, int H, int W, int ldwb);
/* nearest mask resize + BCE-with-logits mean (model/segmenter.py:56-59) and gradient.
 * loss[0] = sum(loss_i)/n (per-block partials in ws [cris_bce_ws_floats()], added in block order: no atomics) ;
 * grad = (sigmoid(x) - t)/n * (*gscale or 1) */
int cris_mask_resize_nearest(const float* mask, int Bn, int IH, int IW, int OH, int OW, float* out, void* stream);
int cris_bce_fwd(const float* logits, const float* target, long n, float* loss, float* ws, void* stream);
int cris_bce_ws_floats(void);
int cris_bce_bwd(const float* logits, const float* target, long n, const float* gscale, float* dlogits, void* stream);
/* trainMetricGPU (utils/misc.py:114-129): out[0] = 100*mean IoU, out[1] = 100*mean(IoU > pr_iou) */
int cris_train_metric(const float* logits, const float* target, int Bn, int HW, float thr, float pr_iou, float* out,
                      void* stream);
/* ---- evaluation post-processing (reference engine/engine.py:100-123 validate, :171-188 inference; csrc/evalpost.hip) ----
 * out[b] = F.interpolate(sigmoid(logits[b]), (H, W), mode='bicubic', align_corners=True)   (engine.py:101-106) */
int cris_sigmoid_bicubic_up(const float* logits, int Bn, int h, int w, int H, int W, float* out, void* stream);
/* dst = cv2.warpAffine(src, mat, (w_out, h_out), flags=cv2.INTER_CUBIC, borderValue=border) (engine.py:114-116); mat: HOST
 * pointer to the 2x3 double matrix the reference passes (param['inverse']); OpenCV's fixed-point algorithm restated */
int cris_warp_affine_cubic(const float* src, int H, int W, const double* mat, int w_out, int h_out, float border, float* dst,
                           void* stream);
/* counts[0] += #(pred > thr & mask), counts[1] += #(pred > thr | mask)  (engine.py:117-122); counts: 2 device ints */
int cris_threshold_iou(const float* pred, const float* mask, long n, float thr, int* counts, void* stream);
/* ---- input preprocessing (reference utils/dataset.py:146-168 RefDataset.__getitem__, :207-221 convert; csrc/inputpipe.hip) ----
 * One launch per batch: img_out[b] = ((cv2.warpAffine(rgb_u8, mat, (S_w, S_h), INTER_CUBIC, borderValue=border_rgb).float()
 * / 255 - mean) / std) as [3][S_h][S_w] (through lut_img [3][256]), mask_out[b] = cv2.warpAffine(mask_u8, mat, ...,
 * INTER_LINEAR, borderValue=0) / 255 (through lut_mask [256]).  OpenCV's 8-bit fixed-point algorithm restated (unpinned:
 * cv2 is not available offline).  samples_dev: DEVICE array of n descriptors; img [H][W][3] / mask [H][W] device bytes (either
 * may be NULL); inv = destination->source matrix (cris_invert_affine of the matrix getTransformMat returns, :190-205). */
typedef struct {
    const unsigned char* img;
    const unsigned char* mask;
    int H, W;
    double inv[6];
} cris_sample_desc;
typedef struct { unsigned char v[4]; } cris_u8x4;
int cris_preprocess_batch(const cris_sample_desc* samples_dev, int n, int S_h, int S_w, const short* tab_linear, const short* tab_cubic,
                          const float* lut_img, const float* lut_mask, const unsigned char* border_rgb /* host, 3 bytes */,
                          float* img_out, float* mask_out, void* stream);
/* host: inv = cv::invertAffineTransform(mat) (2x3 doubles) */
int cris_invert_affine(const double* mat, double* inv);
/* host: the 16-bit remap weight tables of cv::initInterTab2D(fixpt): tab_linear [1024][4], tab_cubic [1024][16] (copy to the device) */
int cris_remap_tables_u8(short* tab_linear, short* tab_cubic);
/* zero fill of any 16-byte aligned buffer */
int cris_zero_bytes(void* p, size_t nbytes, void* stream);
/* zero several byte ranges in one launch (the parts of the gradient arena that are accumulated into or only partly written:
 * BatchNorm sums, embedding rows; everything a kernel overwrites completely every step is left alone) */
#define CRIS_ZERO_RANGES_MAX 16
typedef struct { void* p; size_t nbytes; } cris_zero_range;
typedef struct { int n; int pad_; cris_zero_range r[CRIS_ZERO_RANGES_MAX]; } cris_zero_ranges;
int cris_zero_many(const cris_zero_ranges* r, void* stream);

/* fused multi-tensor Adam (torch.optim.Adam semantics, train.py:105-107): table of {p,g,m,v,n}.
 * p/m/v are in the parameter layout; g is in the parameter layout when taps == 0, else in the GEMM layout
 * [n][tap][cpad] written by cris_conv_wgrad (element (n, c, tap) of the parameter reads g[(n*taps + tap)*cpad + c]).
 * dstF / dstD (optional): the bf16 GEMM-operand copies of a weight (layouts of cris_pack_desc); the update then also writes
 * them from the new values (tile by tile through LDS) - no separate packing pass re-reads the fp32 weights.  A table holds
 * either the 9-tap (3x3) packed tensors or everything else (pack_taps of cris_adam_step: 9 / 1).
 * step_dev (optional, device int32): 1-based step count read on the device - the bias corrections are then
 * 1 - beta^step (in double) and bias_corr1/2 are ignored (lets a captured HIP graph be replayed step after step). */
typedef struct {
    float* p; const float* g; float* m; float* v;
    long n;
    float lr; float pad_;
    int block_start; int taps;
    int cin; int cpad;
    cris_bf16* dstF; cris_bf16* dstD;
    int N; int npad;                 /* packed tensors: rows, padded rows of the D layout (cin / cpad above) */
    int transposed; int ldF;         /* packed: parameter stored [cin][N] (one tap); ldF / ldD: row strides of dstF / dstD in elements */
    const unsigned char* row_live;   /* optional (plain tensors, weight_decay == 0): byte r != 0 <=> row r (row_len elements) has ever
                                        had a non-zero gradient.  Rows that never had one are skipped: with g = m = v = 0 the
                                        Adam update is the identity (p - step * 0 / (0 + eps) = p), so the result is bit-identical
                                        to the dense update - the token embedding is 17% of CRIS-R50's parameters and a step
                                        touches at most B*L of its 49408 rows. */
    int row_len; int ldD;            /* (ldF / ldD: 0 = dense, as in cris_pack_desc) */
} cris_adam_desc;
int cris_adam_step(const cris_adam_desc* dev_table, int n_desc, int total_blocks, float beta1, float beta2, float eps,
                   float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, const int32_t* step_dev,
                   int pack_taps, void* stream);
/* The same update under torch.amp.GradScaler, decided on the device (what `scaler.step(optimizer)` does for an optimizer with
 * `_step_supports_amp_scaling`; engine/engine.py:56): loss_scale_dev (optional) = the scaler's scale, the gradients are divided
 * by it inside the update; skip_dev (optional) = the scaler's found_inf, != 0 skips the whole update (parameters, moments and
 * operand copies untouched).  cris_counter_advance_unless keeps a device step count that does not advance on a skipped step. */
int cris_adam_step_amp(const cris_adam_desc* dev_table, int n_desc, int total_blocks, float beta1, float beta2, float eps,
                       float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, const int32_t* step_dev,
                       const float* loss_scale_dev, const float* skip_dev, int pack_taps, void* stream);
int cris_counter_advance_unless(int32_t* counter, const float* skip, void* stream);
int cris_adam_blocks(const cris_adam_desc* d);       /* blocks one descriptor occupies (host: block_start prefix sums) */
int cris_adam_block_elems(void);
/* dst (param layout, desc.p) <- src (GEMM layout, desc.g) for a table of tensors; block_start as for cris_adam_step */
int cris_unpack_grads(const cris_adam_desc* dev_table, int n_desc, int total_blocks, void* stream);

/* ---- The sentence-vector path in fp32 (csrc/smallf32.hip) ----------------------------------------------------------------
 * At most CRIS_SMALL_MAX_ROWS (= the per-GPU batch) rows: LayerNorm of the end-of-text rows (model/clip.py:449-452), `@
 * text_projection` (:456), neck.txt_proj = Linear + BatchNorm1d + ReLU (model/layers.py:262-264,286), proj.txt = Linear
 * (model/layers.py:58,71).  The bf16 path's loss error is dominated by the rounding of exactly these values (one rounding of the
 * sentence vector shifts every logit of a sample; profiles/parity_r04.md), and they are a few MFLOP: fp32 on the VALU, reading the
 * fp32 parameters directly.  Matrices are row-major with a leading dimension in floats.
 *   cris_eot_gather_ln_f32     out[b] = LayerNorm(x[b, argmax tokens[b]]) from the saved row statistics; eot_index out
 *   cris_eot_scatter_add_f32   dx[b, eot_b] += d out[b]   (dx: the bf16 gradient of the LayerNorm output)
 *   cris_linear_f32_small      out[M][N] (+)= A[M][K] W^T (+ bias), W = [N][K] (w_is_kn 0) or [K][N] (w_is_kn 1: x @ P, and the
 *                              input gradient of a [N][K] layer: dA = dOut W)
 *   cris_outer_sum_f32_small   G[R][C] = sum_m X1[m][r] X2[m][c] (weight gradient of either layout), rowsum[r] = sum_m X1[m][r] (bias)
 *   cris_colstats_f32_small    one statistics part (sum, M2) per column for cris_bn_finalize(_sync)
 *   cris_bn_relu_f32_small     z = relu(scale y + shift);  _bwd_reduce: ONE partial row [2C] for cris_bn_bwd_sum(_sync);
 *                              _bwd_apply: dy from the summed (global) sums */
#define CRIS_SMALL_MAX_ROWS 16
int cris_eot_gather_ln_f32(const int64_t* tokens, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, int Bn, int L, int D, float* out, int* eot_index, void* stream);
int cris_eot_scatter_add_f32(const int* eot_index, const float* drows, int Bn, int L, int D, cris_bf16* dx, void* stream);
int cris_linear_f32_small(const float* A, int lda, const float* W, int ldw, int w_is_kn, const float* bias, int M, int N, int K,
                          float* out, int ldo, int accumulate, void* stream);
int cris_outer_sum_f32_small(const float* X1, int ld1, const float* X2, int ld2, int M, int R, int C, float* G, int ldg, float* rowsum,
                             void* stream);
int cris_colstats_f32_small(const float* y, int ldy, int M, int C, float* psum, float* pm2, void* stream);
int cris_bn_relu_f32_small(const float* y, int ldy, const float* scale, const float* shift, int M, int C, float* z, int ldz, void* stream);
int cris_bn_relu_bwd_reduce_f32_small(const float* dz, int lddz, const float* y, int ldy, const float* scale, const float* shift,
                                      const float* mean, const float* invstd, int M, int C, float* part, void* stream);
int cris_bn_relu_bwd_apply_f32_small(const float* dz, int lddz, const float* y, int ldy, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, const float* sums, float count, int M, int C, float* dy,
                                     int lddy, void* stream);

/* ---- Low-latency cross-rank sum of small fp32 vectors (EXPERIMENTAL, CRIS_SYNCBN_P2P=1) -------------------------------
 * Replaces the per-layer all_gather / all_reduce of nn.SyncBatchNorm (train.py:97-98; 142 small collectives per CRIS-R50
 * step) by ONE kernel per exchange: every rank owns a fine-grained device mailbox that all peers map through HIP IPC; the
 * kernel writes this rank's vector into every mailbox, publishes a generation flag, waits for the peers' flags and adds
 * the world vectors in rank order (bit-identical on all ranks).  Mailbox layout in csrc/p2p.hip. */
size_t cris_p2p_mailbox_bytes(int world, int slots, int max_floats);
int cris_p2p_alloc(size_t bytes, void** dev_ptr);                     /* fine-grained device memory, zero-filled */
int cris_p2p_free(void* dev_ptr);
int cris_p2p_export(void* dev_ptr, void* handle_64_bytes);            /* IPC handle of a mailbox (64 bytes) */
int cris_p2p_import(const void* handle_64_bytes, void** peer_ptr);    /* map another process's mailbox */
int cris_p2p_close(void* peer_ptr);
typedef struct {
    float* data;              /* [n] in: this rank's vector; out: the sum over ranks */
    void* const* boxes;       /* DEVICE array [world]: every rank's mailbox as mapped in this process (own one at [rank]) */
    const int* gen_dev;       /* device step counter (graph / command-list replay) or NULL -> gen_host */
    int* err;                 /* device int set to 1 if a peer never arrived (the output is then NaN); or NULL */
    int n, rank, world;
    int slot, slots;          /* which exchange of the step this is; exchanges per step the mailbox was sized for */
    int max_floats;           /* vector capacity the mailbox was sized for */
    int gen_host;
    int spin_limit;          /* polls before a missing peer is reported (err[0] = 1, NaN result); 0 = the default (~seconds) */
} cris_p2p_params;
int cris_p2p_allreduce_sum(const cris_p2p_params* p, void* stream);

/* The same mailboxes reached from INSIDE other kernels ("LL" words: value and generation in one 8-byte store, csrc/p2p_ll.h; the
 * LL region lies behind the flag-protocol region and is included in cris_p2p_mailbox_bytes).  A link names one exchange of the
 * step; the kernels that take one exchange the values they own themselves:
 *   cris_bn_finalize_sync      SyncBatchNorm forward in ONE launch: merge the local partials, moments about the running mean
 *                              to every peer, rank-order sum, scale / shift + running statistics (replaces cris_bn_finalize +
 *                              cris_bn_sync_pack + exchange + cris_bn_sync_unpack + cris_bn_finalize; same arithmetic)
 *   cris_bn_bwd_reduce_sync    SyncBatchNorm backward: the reduce launch, then ONE launch that adds up the partial rows, adds the
 *                              local sums into `local_sums` (this rank's d beta / d gamma) and leaves the sums over all ranks in
 *                              p->sums for cris_bn_bwd_apply (replaces sum + axpy + exchange)
 *   cris_p2p_ll_allreduce_sum  plain in-place sum of a vector (start-up self-test of the LL words)
 * Reference semantics: torch.nn.SyncBatchNorm as installed by train.py:97-98. */
typedef struct {
    void* const* boxes;       /* DEVICE array [world]: every rank's mailbox as mapped in this process (own one at [rank]) */
    int* err;                 /* device int set to 1 if a peer never arrived (results are then NaN); or NULL */
    const int* gen_dev;       /* device step counter (graph / command-list replay) or NULL -> gen_host */
    int gen_host;
    int rank, world;          /* world <= 0: no exchange (the kernels behave like their plain forms); 1: the exchange with itself */
    int slot, slots;          /* which exchange of the step this is; exchanges per step the mailbox was sized for */
    int max_floats;           /* vector capacity the mailbox was sized for */
    int spin_limit;           /* polls before a missing peer is reported; 0 = the default (~seconds) */
} cris_p2p_link;
int cris_p2p_ll_allreduce_sum(const cris_p2p_link* link, float* data, int n, void* stream);
int cris_bn_finalize_sync(const float* psum, const float* pm2, int nparts, int rows_per_part, float count_local, float count,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, int C, float* scale, float* shift, float* mean, float* invstd, const cris_p2p_link* link,
                          void* stream);
int cris_bn_bwd_reduce_sync(const cris_bn_bwd_params* p, float* local_sums, const cris_p2p_link* link, void* stream);
/* nparts > 0: the partial rows are already in p->part (cris_bn_bwd_sum's case), only the summation + exchange launch runs */
int cris_bn_bwd_sum_sync(const cris_bn_bwd_params* p, int nparts, float* local_sums, const cris_p2p_link* link, void* stream);

/* ---- Gradient exchange over the peer-mapped gradient arenas (csrc/p2p.hip; opt-in: CRIS_GRAD_EXCHANGE=p2p) ---------------
 * What DistributedDataParallel's bucketed all-reduce does for the reference (train.py:100-102), as SURVEY.md section 5 / 8e asks
 * for it on a fully connected xGMI node: a direct reduce-scatter + all-gather in which every rank talks to all seven peers at once,
 * instead of RCCL's schedule.  Every rank's gradient arena (one flat fp32 tensor) is mapped into every other process (HIP-IPC);
 * for one range [lo, lo + n) of the arena, in place:
 *   launch 1  barrier "my gradients of this range are final" - then rank r sums slice r of the range over the ranks IN RANK ORDER
 *             (reads of the peers' arenas over the links, bit-identical results on every rank, no atomics) into its own arena;
 *   launch 2  barrier "my slice is reduced" - then every rank copies the other ranks' reduced slices into its own arena;
 *   launch 3  barrier "I have finished reading" - returns when every peer has: the arena may be written again.
 * The barriers are LL words of the peer mailboxes (cris_p2p_link; three consecutive slots per exchange, generation = the step).
 * All three launches go to `stream` (the communicator's side stream: the exchange overlaps the rest of backward) and can be
 * captured into a HIP graph.  A peer that never arrives: bounded poll, link.err set, this rank's slice poisoned with NaN. */
typedef struct {
    void* const* arenas;      /* DEVICE array [world]: base address of every rank's arena as mapped in THIS process (own: its own) */
    long lo, n;               /* range in floats; both even (8-byte words travel) */
    cris_p2p_link link;       /* link.slot = the first of this exchange's three barrier slots */
    int blocks;               /* blocks per launch, 0 = default */
} cris_p2p_arena_params;
int cris_p2p_arena_allreduce(const cris_p2p_arena_params* p, void* stream);

/* ---- Data-parallel exchanges on library-owned RCCL communicators (csrc/comm.hip) ---------------------------------------
 * What the reference gets from `dist.init_process_group("nccl")` + `DistributedDataParallel` + `SyncBatchNorm`
 * (train.py:80-102; SURVEY.md 8b "comm entry points"), without torch.distributed on the data path.  One cris_comm per
 * process (= per GPU, the device current at cris_comm_init).  RCCL is resolved at run time (the librccl.so.1 already in the
 * process, else the system one; CRIS_RCCL_LIB overrides): the library does not link against it.
 *   bootstrap:  rank 0 calls cris_comm_unique_id and ships the CRIS_COMM_ID_BYTES bytes to the other ranks by any host
 *               channel (a c10d Store, a file, MPI); every rank then calls cris_comm_init (collective).
 *   SyncBN:     cris_comm_syncbn_exchange - in-place fp32 sum over ranks on the CALLER's stream (replaces the all_gather of
 *               (mean, invstd, count) / the all_reduce of (sum_dy, sum_dy_xmu) in torch.nn.SyncBatchNorm).
 *   gradients:  cris_comm_allreduce_bucket - in-place fp32 sum of one range of the gradient arena on the communicator's OWN
 *               side stream and OWN RCCL communicator, after everything `ready_stream` has queued so far (replaces DDP's
 *               bucketed all-reduce; the caller chooses few large ranges: xGMI is per-link bound); cris_comm_wait makes
 *               `stream` wait for every range issued since the last wait (before the optimizer).  Averaging (1/world) is
 *               left to the optimizer's grad_scale.
 *   start-up:   cris_comm_broadcast - rank root's bytes to all ranks (DDP broadcasts parameters and buffers when it wraps).
 * All calls enqueue on streams and return; they can be captured into a HIP graph together with the step's kernels. */
#define CRIS_COMM_ID_BYTES 256                 /* two ncclUniqueId: the statistics and the gradient communicator */
typedef struct cris_comm cris_comm;
const char* cris_comm_rccl_path(void);         /* which RCCL was resolved (NULL + cris_last_error() if none) */
int cris_comm_unique_id(void* id_bytes);
int cris_comm_init(int rank, int world, const void* id_bytes, cris_comm** out);
int cris_comm_destroy(cris_comm* c);
int cris_comm_rank(const cris_comm* c);
int cris_comm_world(const cris_comm* c);
int cris_comm_syncbn_exchange(cris_comm* c, float* stats, size_t n, void* stream);
int cris_comm_allreduce_bucket(cris_comm* c, float* buf, size_t n, void* ready_stream);
int cris_comm_wait(cris_comm* c, void* stream);
int cris_comm_broadcast(cris_comm* c, void* buf, size_t nbytes, int root, void* stream);

/* ---- Baseline JPEG decoding (csrc/jpeg.hip) -------------------------------------------------------------
 * Replaces `cv2.imdecode(np.frombuffer(ref['img'], np.uint8), cv2.IMREAD_COLOR)` + `cv2.cvtColor(.., COLOR_BGR2RGB)` of the
 * reference's loader (utils/dataset.py:127-129) for the files its LMDB records hold (tools/folder2lmdb.py:50-56 stores the raw
 * JPEG bytes): 8-bit Huffman-coded JPEG, sequential (one interleaved scan or one scan per component) or progressive
 * (spectral selection + successive approximation), gray or YCbCr 4:4:4 / 4:2:2 / 4:2:0, restart intervals.  Hybrid split, as the bit-serial part does not parallelise inside an image:
 *   host   cris_jpeg_read_header, cris_jpeg_decode_coefficients(_batch): marker parsing + Huffman decoding (T.81 Annex F) into
 *          quantised coefficient blocks int16 [component][block row][block col][64] (natural order), one image per thread;
 *   device cris_jpeg_reconstruct: dequantisation + libjpeg's islow inverse DCT -> sample planes, then fancy chroma upsampling +
 *          YCbCr -> RGB -> uint8 [H][W][3], a whole ragged batch in two launches; the result is what cris_preprocess_batch reads.
 * Bit-exact with libjpeg(-turbo) at its default settings (JDCT_ISLOW, do_fancy_upsampling) - pinned against Pillow's
 * libjpeg-turbo through oracle/jpeg_baseline.py.  Anything else (arithmetic coding, lossless, 12-bit, CMYK, other samplings)
 * is refused with an error: the caller keeps such a file on its CPU decoder.  The EXIF orientation is the host mirror's job
 * (cris/pytorch_amd/jpegdec.py turns the decoded image like cv2.imdecode does). */
typedef struct {
    int width, height, ncomp;            /* ncomp 1 (gray) or 3 (YCbCr) */
    int hmax, vmax;                      /* luma sampling factors: (1,1) 4:4:4, (2,1) 4:2:2, (2,2) 4:2:0 */
    int mcus_x, mcus_y, restart_interval;
    int comp_h[3], comp_v[3];
    int blocks_w[3], blocks_h[3];        /* block grid of each component (whole MCUs) */
    int down_w[3], down_h[3];            /* real samples of each component (libjpeg downsampled_width / height) */
    unsigned short quant[3][64];         /* quantisation table of each component, natural order */
    long coef_offset[3];                 /* int16 index of each component's blocks inside the image's coefficient buffer */
    long coef_count;                     /* int16 elements of that buffer */
    long plane_offset[3];                /* byte index of each component's [blocks_h*8][blocks_w*8] sample plane in the scratch */
    long plane_bytes;
    long scan_offset;                    /* byte offset of the entropy-coded segment in the file */
    int total_blocks;
    int multiscan;                       /* 0: one interleaved sequential scan; 1: sequential, one scan per component; 2: progressive (SOF2) */
} cris_jpeg_info;
int cris_jpeg_read_header(const unsigned char* data, size_t nbytes, cris_jpeg_info* info);                         /* host */
int cris_jpeg_decode_coefficients(const unsigned char* data, size_t nbytes, const cris_jpeg_info* info, short* coef); /* host */
/* host: images [0, n) on up to n_threads threads; returns 0 or the first failing image's error (cris_last_error names it) */
int cris_jpeg_decode_coefficients_batch(int n, const unsigned char* const* data, const size_t* nbytes, const cris_jpeg_info* infos,
                                        short* const* coefs, int n_threads);
typedef struct {
    const short* coef;                   /* DEVICE: info.coef_count int16 */
    unsigned char* planes;               /* DEVICE scratch: info.plane_bytes */
    unsigned char* rgb;                  /* DEVICE out: [height][width][3] */
    cris_jpeg_info info;
} cris_jpeg_image;
/* dev_table: DEVICE array of n images; max_blocks / max_pixels: the largest info.total_blocks / width*height in the batch */
int cris_jpeg_reconstruct(const cris_jpeg_image* dev_table, int n, int max_blocks, long max_pixels, void* stream);

/* ---- PNG masks (csrc/png.hip; host only) -----------------------------------------------------------------------------------
 * Replaces `cv2.imdecode(np.frombuffer(ref['mask'], np.uint8), cv2.IMREAD_GRAYSCALE)` (utils/dataset.py:148-149) for the
 * files tools/data_process.py:115-117 writes (`cv2.imwrite(.., mask * 255)`: 8-bit grayscale, non-interlaced): chunk walk,
 * zlib + DEFLATE, row filters.  Lossless, so "parity" = equality with any PNG decoder (pinned against Pillow's).  Other PNG
 * flavours (colour, palette, 1/2/4/16-bit, interlaced) are refused. */
int cris_png_gray8_size(const unsigned char* data, size_t nbytes, int* width, int* height);
int cris_png_decode_gray8(const unsigned char* data, size_t nbytes, unsigned char* out, int width, int height);   /* out: HOST [height][width] */

#ifdef __cplusplus
}
#endif
#endif
