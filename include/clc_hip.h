/* clc_hip.h — C ABI of libclc_hip.so, the MI355X (gfx950) engine behind the CLC hot path.
 *
 * The reference (ydchen0806/CLC) has no FFI of its own: its callers import Python
 * nn.Modules (`from models import TCM, CLC`, /root/reference/train_CLC.py:25,
 * eval_CLC.py:4) whose arithmetic is executed by cuDNN / cuBLAS / ATen and by the
 * CompressAI C++ extensions.  This header is therefore the boundary a maintainer would
 * bind INSTEAD of those native back-ends (INTEGRATION.md shows the ctypes stub); each
 * entry point cites the reference call site whose native work it replaces.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  Device pointers unless marked HOST.
 *  - All activations are NHWC fp32 ("pixel-major"): element (n,h,w,c) of a tensor lives
 *    at base[((n*H+h)*W+w)*ld + c]; `ld` (floats) lets a call read or write a channel
 *    slice of a wider buffer, so torch.cat / split / chunk never materialise.
 *  - Conv weights are [Cout][KH][KW][Cin] (PyTorch OIHW tensor in channels_last memory format).
 *  - Every function enqueues work on `stream` and returns immediately: no allocation,
 *    no synchronisation, no host read-back -> capturable in a hipGraph.  The caller owns
 *    every buffer including workspaces (sizes from the *_workspace_bytes helpers).
 *  - Return value: >= 0 = ok (clc_conv2d / clc_conv2d_wgrad return the id BM*1000+BN of the
 *    tile variant they launched, 1 = small-Cin kernel), negative = error; clc_last_error()
 *    gives the message (thread-local).  No C++ exception crosses the ABI.
 *  - Results are run-to-run deterministic (no floating-point atomics anywhere) and do not
 *    depend on the batch size for a given image (per-output accumulation order is fixed),
 *    which the encoder/decoder agreement of compress()/decompress() relies on.
 */
#ifndef CLC_HIP_H
#define CLC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* clc_stream_t; /* hipStream_t */

const char* clc_last_error(void);
int clc_version(void);
/* A/B switch between kernel variants that compute the same bits (key 0: K-loop timing of the LDS-DMA convolution, 1 or 2;
 * key 1: stream-K filter gradients, 0 or 1); returns the previous value.  Benchmark tooling only. */
int clc_set_tuning(int key, int value);
int clc_get_tuning(int key);
/* The codec container's kernel-configuration tag (clc_amd/codec.py): which generation of context-model summation orders this build,
 * in its CURRENT tuning state, runs.  1..127 = default tuning of that generation; 128..255 = a hash of the order-affecting keys
 * when any is off its default.  A stream decodes only under the tag it was encoded with (the slice loop is autoregressive through
 * the arithmetic decoder: CLC_run.py:738-814 has the same property across devices and records nothing). */
int clc_kernel_config_tag(void);
/* ... and the full 32-bit hash of the same state (generation + order-affecting keys): containers written under a NON-default state (tag >= 128,
 * 7 hash bits) carry it too, so that two such states cannot be mistaken for each other (clc_amd/codec.py, header version 2). */
unsigned clc_kernel_config_hash(void);

/* ---- activation / epilogue codes ---------------------------------------------------- */
enum { CLC_ACT_NONE = 0, CLC_ACT_LRELU = 1, CLC_ACT_RELU = 2, CLC_ACT_GELU = 3, CLC_ACT_HALFTANH = 4 /* 0.5*tanh(v), LRP head */,
       CLC_ACT_SIGMOID = 5 /* forward only (CLM modulation) */,
       CLC_ACT_SAVED_DERIV = 6 /* backward only: the saved tensor IS act'(pre-activation) (see pre_deriv) */ };
/* input prologue applied to the gathered activations */
enum { CLC_IN_NONE = 0, CLC_IN_SQUARE = 1 };
/* norm modes of the epilogue: out = mul * rsqrt(v) (GDN) or mul * sqrt(v) (inverse GDN) */
enum { CLC_NORM_NONE = 0, CLC_NORM_GDN = 1, CLC_NORM_IGDN = 2,
       CLC_NORM_MUL2 = 3 /* out = 2 * mul * v: the chain-rule factor d(x^2)/dx of GDN's backward, fused into the 1x1 data-gradient conv */ };

/* ---- convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 ------------------------- *
 * Replaces cuDNN conv fwd / bwd-data / bwd-weight behind every nn.Conv2d and nn.Linear
 * of the path: /root/reference/models/CLC_run.py:120,123,185-187,207-208,232-233,
 * 273-277,335-354,371-396,412-481 and the CompressAI layers of SURVEY.md A.1.
 *
 * y[n,oh,ow,co] = epi( sum_{kh,kw,ci} in_op(x[n, oh*s-pad+kh, ow*s-pad+kw, ci]) * w[co,kh,kw,ci] )
 * epi(v): v += bias[co]; (y_pre <- v); v = norm(v, mul); v = act(v); v += res_scale*res
 *         (res_first = 1: the residual is added BEFORE the activation: v = act(v + res_scale*res), y_pre includes it)
 * transposed = 1 computes the data gradient of that conv instead: "x" is dY
 * [N,H,W,Cin=Cout_fwd], "y" is dX [N,OH,OW,Cout=Cin_fwd] and w must be the transposed
 * filter [Cin_fwd][KH][KW][Cout_fwd] (clc_filter_transpose).
 * shuffle = 1 fuses PixelShuffle(2) into the store: y has 2*OH x 2*OW pixels of Cout/4 ch.
 */
typedef struct {
  const float* x; int N, H, W, Cin, ldx;
  const float* w;            /* [Cout][ks][ks][Cin] */
  const float* bias;         /* [Cout] or NULL */
  float* y; int OH, OW, Cout, ldy;
  int ks, stride, pad;
  int transposed;
  int in_op;
  int act;
  int norm; const float* mul; int ldm;
  const float* res; int ldr; float res_scale;
  float* y_pre; int ldp;     /* optional pre-activation copy (training) */
  int shuffle;
  int res_first;
  /* optional fused activation backward on the gathered operand: x <- x * act'(xs) with xs the saved
   * pre-activation (xs_pre = 1) or activation output of the layer whose gradient x is (same geometry as x) */
  const float* xs; int ldxs; int xs_act; int xs_pre;
  /* optional second filter set: images [N/2, N) of the batch are convolved with w2 / bias2 (same geometry as w / bias),
   * images [0, N/2) with w / bias.  Two same-shaped layers that run side by side on different data (the mean- and the
   * scale-parameter nets of a slice, CLC_run.py:560-566) become ONE launch with twice the rows — the 16x16-map layers
   * are latency-bound, so the second one is nearly free.  NULL = ordinary convolution; N must be even. */
  const float* w2; const float* bias2;
  /* pre_deriv = 1: y_pre receives act'(v) instead of v, so the backward of an expensive activation (GELU: erf + exp) is
   * one multiply in the gradient kernels' loaders (xs_act / dys_act = CLC_ACT_SAVED_DERIV) — no elementwise pass */
  int pre_deriv;
  /* optional gate on the residual term: v += res_scale * res * act'(res_gate) (res_gate_pre as xs_pre).  Lets a data-gradient
   * kernel add the gradient of a residual branch that passed through an activation (ResidualUnit: relu(conv(..) + x)) without a
   * separate activation-backward pass */
  const float* res_gate; int ldg; int res_gate_act; int res_gate_pre;
  /* optional gate on the whole result: y *= act'(out_gate) — a data-gradient kernel hands the PRODUCING layer its gradient
   * already multiplied by that layer's activation derivative (out_gate = the producer's activated output, i.e. this layer's
   * forward input, for LeakyReLU / ReLU; the stored derivative for GELU), so the producer's own gradient kernels need no
   * operand prologue (and run on the LDS-DMA path). */
  const float* out_gate; int ldog; int out_gate_act; int out_gate_pre;
  /* optional third and fourth filter set (both or neither, with w2): quarter k of the batch, images [k N/4, (k+1) N/4), is convolved
   * with set k of (w, w2, w3, w4).  The ResidualUnits of an attention block's two branches — conv_a on the block input, conv_b on the
   * Swin output (CLC_run.py:235-244) — of the mean- AND the scale-parameter net are four same-shaped layers on different data: one
   * launch instead of four.  N must be a multiple of 4. */
  const float* w3; const float* bias3; const float* w4; const float* bias4;
  /* optional scratch (clc_conv2d_workspace_bytes(d) bytes, caller-owned, 16-B aligned): lets a DATA-GRADIENT launch whose grid would
   * leave most CUs with one 4-wave workgroup (3x3 layers on 32x32 / 16x16 maps with few output channels) split K over 2-4
   * workgroups per tile — partial tiles to the scratch, a fixed-order finish launch adds them and applies the epilogue.  Without it
   * (NULL) the launch is not split.  Forward launches are not split unless batch_variant_ok is set: an image's bits must not depend on
   * the batch size in the codec. */
  void* workspace; size_t workspace_bytes;
  /* nonzero: this launch's result may depend (in summation order only) on the batch size — set by training forward passes, never by the
   * codec path.  Lets the long-K forward layers of the slice-parameter nets (3x3, 384..704 -> 224 on 16x16 maps) run on 128x128 LDS tiles
   * with a K split sized to the grid (needs `workspace`). */
  int batch_variant_ok;
  /* optional: the same filter in the fragment order of conv_halo3x3_kernel (clc_filter_pack_halo; for transposed = 1: of the transposed
   * filter).  With it, 3x3 / stride-1 / pad-1 layers with 128 (64) input channels and a multiple of 128 (64) output channels on maps whose
   * height is a multiple of 8 and width of 16 run on the halo-resident kernel (csrc/conv_halo.hip) — same bits as the tiled kernels.  NULL: tiled. */
  const float* w_packed;
  /* optional: the filter's Winograd F(2x2, 3x3) transform in fragment order (clc_filter_wino; for transposed = 1: of the transposed filter, taps
   * flipped).  With it, 3x3 / stride-1 / pad-1 layers with 64 k input and 64 k output channels on maps whose height is a multiple of 8 and
   * width of 16 run on conv_wino_kernel / conv_wino64_kernel (csrc/conv_wino.hip): 2.25x fewer multiplications, ANOTHER summation order than the direct kernels
   * (fp32 error of a few ulp of the operands; measured 4..9e-7 of the largest output against fp64, the direct kernels 1.3e-6).  Takes precedence
   * over w_packed.  For TRAINING launches (forward of a recorded pass, data gradients): a launch whose result feeds the entropy coder or a parity
   * measurement leaves it NULL (direct kernels) — the codec's kernel_config tag does not cover this kernel. */
  const float* w_wino;
} clc_conv_desc;

int clc_conv2d(const clc_conv_desc* d, clc_stream_t stream);
/* [N][3][3][K] filter rows, K = 128 or 64, N a multiple of K (the forward filter [Cout][kh][kw][Cin = K], or the transposed filter
 * [Cin][kh][kw][Cout = K] of a K -> K layer) -> `out` (same number of floats) in the fragment order clc_conv_desc.w_packed expects. */
int clc_filter_pack_halo(const float* w, float* out, int N, int K, clc_stream_t stream);
/* ... for every such filter of a model in ONE launch: device table of entries; block_begin = running sum of ceil(N * 9 * K / 4 / 256) (blocks of 256
 * 16-B elements) over the preceding entries, total_blocks = the grand total */
typedef struct { const float* w; float* out; int N, K, block_begin; } clc_halo_pack_entry;
int clc_filter_pack_halo_batched(const clc_halo_pack_entry* table_dev, int n_entries, int total_blocks, clc_stream_t stream);
/* [N][3][3][K] filter rows (N, K multiples of 64) -> `out` (ceil(N / 128) * 128 * 16 * K floats): U = G g G^T per (row, channel) in the fragment order
 * clc_conv_desc.w_wino expects; flip = 1 reverses the taps (data gradients: pass the transposed filter).  The batched form: device table,
 * block_begin = running sum of ceil(N * K / 4 / 256). */
int clc_filter_wino(const float* w, float* out, int N, int K, int flip, clc_stream_t stream);
typedef struct { const float* w; float* out; int N, K, flip, block_begin; } clc_wino_entry;
int clc_filter_wino_batched(const clc_wino_entry* table_dev, int n_entries, int total_blocks, clc_stream_t stream);
/* scratch bytes with which clc_conv2d would split this launch's K range (0: it would not split) */
size_t clc_conv2d_workspace_bytes(const clc_conv_desc* d);

/* Weight gradient: dw[co,kh,kw,ci] = sum_{n,oh,ow} dy[n,oh,ow,co] * in_op(x[n,oh*s-pad+kh,ow*s-pad+kw,ci]).
 * Deterministic split-K: partial slabs in `workspace`, summed in a fixed order.
 * dbias (optional) = column sums of dy. accumulate=1 adds into dw/dbias. */
typedef struct {
  const float* x; int N, H, W, Cin, ldx;
  const float* dy; int OH, OW, Cout, lddy;
  float* dw; float* dbias;
  int ks, stride, pad;
  int in_op;
  int accumulate;
  void* workspace; size_t workspace_bytes;
  /* optional fused activation backward on dy: dy <- dy * act'(dys) (same geometry as dy) */
  const float* dys; int lddys; int dys_act; int dys_pre;
} clc_wgrad_desc;

size_t clc_conv2d_wgrad_workspace_bytes(const clc_wgrad_desc* d);
int clc_conv2d_wgrad(const clc_wgrad_desc* d, clc_stream_t stream);
/* `count` independent filter-gradient problems (each with its own workspace): same results, bit for bit, as `count`
 * calls of clc_conv2d_wgrad in order, but problems of one tile shape share a grid and all slab sets share one
 * fixed-order reduce launch (the 16x16-map layers of the slice loop are launch-bound one by one).  Problems that write
 * the same dw / dbias (a filter applied twice, e.g. the reference encoder over several reference frames) are kept in
 * separate launches, in order. */
int clc_conv2d_wgrad_batched(const clc_wgrad_desc* descs, int count, clc_stream_t stream);
/* Stream-K form of the grouped launch: ONE grid of <= 512 workgroups per tile shape walks the K-units of all problems laid end
 * to end; a workgroup accumulates in registers while it stays inside an output tile and leaves at most two partial tile
 * images in `group_workspace` (clc_conv2d_wgrad_group_workspace_bytes(), caller-owned, reusable across calls on one stream),
 * which a fix-up launch adds in workgroup order.  Same results as clc_conv2d_wgrad_batched up to summation order; run-to-run
 * reproducible (the ranges depend only on the group's shapes). */
size_t clc_conv2d_wgrad_group_workspace_bytes(void);
/* per-problem workspace a problem needs INSIDE clc_conv2d_wgrad_batched_sk: 0 for every plan whose partial tiles live in the
 * group workspace (d->workspace may then be NULL), clc_conv2d_wgrad_workspace_bytes(d) for the small-Cin split path. */
size_t clc_conv2d_wgrad_sk_workspace_bytes(const clc_wgrad_desc* d);
/* kernel family a problem is planned on: 1 = small-Cin VALU kernel, 64900 + TW = all-taps 3x3 kernel (TW = 32 | 16 | 8),
 * BM * 1000 + BN = tap-per-workgroup kernel (same ids clc_conv2d_wgrad returns).  Benchmark tooling: FLOP accounting per kernel. */
int clc_conv2d_wgrad_variant(const clc_wgrad_desc* d);
int clc_conv2d_wgrad_batched_sk(const clc_wgrad_desc* descs, int count, void* group_workspace, size_t group_workspace_bytes,
                                clc_stream_t stream);

/* [Cout][T][Cin] -> [Cin][T][Cout]  (T = ks*ks) */
int clc_filter_transpose(const float* w, float* wt, int Cout, int T, int Cin, clc_stream_t stream);
/* all filters of a model in ONE launch: device table of entries; tile_begin = running sum of
 * T*ceil(Cout/32)*ceil(Cin/32) over the preceding entries, total_tiles = the grand total */
typedef struct { const float* w; float* wt; int Cout, T, Cin, tile_begin; } clc_transpose_entry;
int clc_filter_transpose_batched(const clc_transpose_entry* table_dev, int n_entries, int total_tiles, clc_stream_t stream);

/* ---- elementwise / normalisation ------------------------------------------------------ */
/* dz = dy * act'(saved); saved = pre-activation (use_pre=1) or the activation output */
int clc_act_bwd(const float* dy, int lddy, const float* saved, int lds, int use_pre, int act, float* dz, int lddz,
                long rows, int C, clc_stream_t stream);
/* column sums: out[c] (+)= sum_r x[r*ld + c]; workspace >= clc_colsum_workspace_bytes */
size_t clc_colsum_workspace_bytes(long rows, int C);
int clc_colsum(const float* x, int ld, long rows, int C, float* out, int accumulate, void* ws, size_t ws_bytes,
               clc_stream_t stream);

/* Deferred parameter-gradient reductions: out[i] (+)= sum_{b < nblocks} partial[b*n + i], i < n; columns i < split go to
 * out0[i], the rest to out1[i - split] (LayerNorm: n = 2C, split = C, out0 = dgamma, out1 = dbeta; relative-position bias:
 * split = n).  clc_layernorm_bwd with dgamma = dbeta = NULL and clc_winattn_bwd with drelbias = NULL leave their partial
 * rows in the workspace (nblocks = workspace_bytes / (n * 4)); one call here sums up to 64 of them per launch, in a fixed
 * order.  `entries` is a HOST array (copied into the kernel arguments). */
typedef struct { const float* partial; int nblocks; int n; float* out0; float* out1; int split; int accumulate; } clc_reduce_entry;
int clc_partial_reduce_batched(const clc_reduce_entry* entries, int count, clc_stream_t stream);

/* LayerNorm over the channel dim of [rows, C] tokens (nn.LayerNorm, CLC_run.py:180,183), eps = 1e-5.
 * fwd saves mean/rstd per row when non-NULL. */
int clc_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* mean,
                      float* rstd, long rows, int C, clc_stream_t stream);
/* dx (+ dx_add when non-NULL: the gradient of the residual branch x + f(LN(x)), CLC_run.py:190-191, folded into the same
 * pass); dgamma/dbeta partial sums go through ws (deterministic two-stage) */
size_t clc_layernorm_bwd_workspace_bytes(long rows, int C);
/* number of partial rows ([2][C] each) the backward leaves in the workspace (paired: the first half of them belongs to the
 * first module, the second half to the second) */
int clc_layernorm_bwd_blocks(long rows, int paired);
int clc_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean,
                      const float* rstd, float* dx, int lddx, const float* dx_add, int ld_add, float* dgamma, float* dbeta,
                      int accumulate, long rows, int C, void* ws, size_t ws_bytes, clc_stream_t stream);

/* Paired modules (the mean- and scale-parameter nets of a slice run as ONE stacked batch, see clc_conv_desc.w2): rows
 * [0, half_rows) use gamma / beta, rows [half_rows, rows) use gamma2 / beta2.  C must be 64, 128 or 256. */
int clc_layernorm_fwd_pair(const float* x, int ldx, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                           long half_rows, float* y, int ldy, float* mean, float* rstd, long rows, int C, clc_stream_t stream);
int clc_layernorm_bwd_pair(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* gamma2, long half_rows,
                           const float* mean, const float* rstd, float* dx, int lddx, const float* dx_add, int ld_add,
                           float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, int accumulate, long rows, int C, void* ws,
                           size_t ws_bytes, clc_stream_t stream);

/* GDN backward pieces (CompressAI GDN inside ResidualBlockWithStride / ResidualBlockUpsample):
 * given dy, x, norm v = beta + gamma.x^2 :  dx_direct = dy * f(v);  dv = dy * x * f'(v)
 * with f = rsqrt (inverse=0) or sqrt (inverse=1).  */
int clc_gdn_bwd_elem(const float* dy, const float* x, const float* v, float* dx_direct, float* dv, long n, int inverse,
                     clc_stream_t stream);
/* dx = dx_direct + 2*x*t  (t = gamma^T . dv from clc_conv2d) */
int clc_gdn_bwd_combine(const float* dx_direct, const float* x, const float* t, float* dx, long n, clc_stream_t stream);
/* CompressAI NonNegativeParametrizer (GDN beta / gamma): y = max(x, bound)^2 - pedestal for gamma [C,C] and
 * beta [C] in one launch (gamma_eff also written transposed for the data-gradient conv); backward with the LowerBound
 * gradient rule (pass where x >= bound or the gradient pushes x up), optionally accumulating into dgamma / dbeta. */
int clc_gdn_reparam_fwd(const float* gamma, const float* beta, int C, float gamma_bound, float beta_bound, float pedestal,
                        float* gamma_eff, float* gamma_eff_t, float* beta_eff, clc_stream_t stream);
/* The forward re-parametrisation of MANY GDN modules in one launch: `table_dev` = n_entries device-resident entries, entry e owning the
 * blocks [first_block, first_block + ceil((C*C + C) / 256)) of a grid of total_blocks workgroups of 256 threads. */
typedef struct {
  const float* gamma; const float* beta;
  float* gamma_eff; float* gamma_eff_t; float* beta_eff;
  int C, first_block;
  float gamma_bound, beta_bound, pedestal;
} clc_gdn_entry;
int clc_gdn_reparam_fwd_batched(const clc_gdn_entry* table_dev, int n_entries, int total_blocks, clc_stream_t stream);
int clc_gdn_reparam_bwd(const float* gamma, const float* beta, int C, float gamma_bound, float beta_bound, const float* dgamma_eff,
                        const float* dbeta_eff, float* dgamma, float* dbeta, int accumulate, clc_stream_t stream);
/* backward of PixelShuffle(2) fused with the activation backward of the conv that was stored shuffled:
 * dz[n,h,w,4q+r] = dy[n,2h+(r>>1),2w+(r&1),q] * act'(saved[same]); dy/saved [N,C/4,2H,2W] pixel-major, dz [N,C,H,W] dense.
 * saved may be NULL (no activation). */
int clc_unshuffle_act_bwd(const float* dy, int lddy, const float* saved, int lds, int use_pre, int act, float* dz, int N, int H,
                          int W, int C, clc_stream_t stream);

/* out = a * sigmoid(b) + idn  (SWAtten gate, CLC_run.py:241-242) and its backward */
int clc_gate_fwd(const float* a, const float* b, const float* idn, float* out, long n, clc_stream_t stream);
int clc_gate_bwd(const float* dout, const float* a, const float* b, float* da, float* db, long n, clc_stream_t stream);

/* generic fused adds: out = alpha*a + beta*b */
/* dst = ((src0 + src1) + src2) + ...: up to 12 pixel-major [rows][C] sources (HOST arrays of device pointers / leading dimensions), fixed
 * order — the gradient of a tensor with several consumers in one launch (ops.fanout) instead of autograd's chain of pairwise adds. */
int clc_sum_n(const float* const* srcs, const int* lds, int n, float* dst, int lddst, long rows, int C, clc_stream_t stream);
int clc_axpby(const float* a, float alpha, const float* b, float beta, float* out, long n, clc_stream_t stream);
/* RGB-head filters (3x3 / stride 2 [co][3][3][cin] and its 1x1 skip [co][cin], 9 cin <= 32) as the 32-column matrices of the patch-row
 * formulation (clc_im2col_small): w1[o][k] = w3[o][k] (k < 9 cin), ws[o][4 cin + c] = w1x1[o][c], zero elsewhere; and the reverse, ADDING the
 * two matrices' gradients into the parameters' gradient buffers. */
int clc_stem_pack(const float* w3, const float* w1x1, float* w1, float* ws, int co, int cin, clc_stream_t stream);
int clc_stem_unpack_add(const float* dw1, const float* dws, float* g3, float* g1x1, int co, int cin, clc_stream_t stream);
/* strided copy of a channel slice: dst[r*ldd + c] = src[r*lds + c] (c < C) */
int clc_copy2d(const float* src, int lds, float* dst, int ldd, long rows, int C, clc_stream_t stream);
/* Patch rows of a few-channel image (the RGB heads, Cin = 3: g_a.0 / ref_encoder.encoder.0 of CLC_run.py:274,335):
 * col[m][(kh*ks+kw)*C + c] = x[n, oh*s-pad+kh, ow*s-pad+kw, c], zeros outside the image and in columns ks*ks*C..ldc-1.
 * The 3x3/s2 conv AND the 1x1/s2 skip conv of ResidualBlockWithStride (its input is the centre tap) then run as
 * 1x1 convolutions over `col` on the MFMA kernel. */
int clc_im2col_small(const float* x, int ldx, int N, int H, int W, int C, int ks, int stride, int pad, float* col, int ldc,
                     int OH, int OW, clc_stream_t stream);

/* ---- window attention (WMSA core, CLC_run.py:142-164) --------------------------------- *
 * qkv: [B,H,W,3C] tokens, channel = which*C + head*hd + c ; out: [B,H,W,C].
 * One workgroup per (window, head): softmax(q k^T * hd^-1/2 + relbias[head] (+ shift mask)) v.
 * shift=1 -> cyclic shift by ws/2 folded into the addressing (torch.roll never materialises).
 * relbias: [heads][2ws-1][2ws-1] (the parameter itself; the index gather is done in-kernel). */
/* lse (optional, training): [B*H*W][heads] log-sum-exp of every softmax row, consumed by the backward */
int clc_winattn_fwd(const float* qkv, int ldq, const float* relbias, float* out, int ldo, float* lse, int B, int H, int W,
                    int C, int heads, int ws, int shift, clc_stream_t stream);
/* backward: rebuilds probabilities from qkv + lse, uses the forward output for D = dO.O;
 * drelbias partials via ws (deterministic) */
size_t clc_winattn_bwd_workspace_bytes(int B, int H, int W, int heads, int ws);
int clc_winattn_bwd(const float* dout, int lddo, const float* qkv, int ldq, const float* relbias, const float* out,
                    int ldo, const float* lse, float* dqkv, int lddq, float* drelbias, int accumulate, int B, int H, int W,
                    int C, int heads, int ws, int shift, void* wsb, size_t ws_bytes, clc_stream_t stream);
/* Paired modules: images [B/2, B) use relbias2 (clc_conv_desc.w2 explains the pairing); B even.  The backward's partial
 * rows: the first half of clc_winattn_bwd_blocks(..., 1) belongs to the first module. */
int clc_winattn_bwd_blocks(int B, int H, int W, int heads, int ws, int paired);
int clc_winattn_fwd_pair(const float* qkv, int ldq, const float* relbias, const float* relbias2, float* out, int ldo, float* lse,
                         int B, int H, int W, int C, int heads, int ws, int shift, clc_stream_t stream);
int clc_winattn_bwd_pair(const float* dout, int lddo, const float* qkv, int ldq, const float* relbias, const float* relbias2,
                         const float* out, int ldo, const float* lse, float* dqkv, int lddq, float* drelbias, float* drelbias2,
                         int accumulate, int B, int H, int W, int C, int heads, int ws, int shift, void* wsb, size_t ws_bytes,
                         clc_stream_t stream);

/* ---- entropy-model kernels ------------------------------------------------------------ *
 * GaussianConditional likelihood + rate (CLC_run.py:569-571, train_CLC.py:48-51)
 *   mode 0 (train): y_in = y + noise ; mode 1 (eval): y_in = round(y-mu)+mu
 *   sigma = max(scale, 0.11); lik = max(Phi((.5-|y_in-mu|)/sigma) - Phi((-.5-|..|)/sigma), 1e-9)
 * Writes lik, y_hat = round(y-mu)+mu (STE value) and adds sum(log2 lik) of this call into
 * bits_partial[blockIdx] (two-stage deterministic reduction; finish with clc_sum_partials). */
int clc_gauss_lik_fwd(const float* y, int ldy, const float* mu, int ldmu, const float* scale, int ldsc,
                      const float* noise, int ldn, float* lik, int ldl, float* y_hat, int ldh, long rows, int C,
                      int mode, float* bits_partial, int n_partials, clc_stream_t stream);
/* grads wrt y, mu, scale given dlik (gradient wrt lik) — includes both LowerBound rules.  dy_add (optional, leading dimension ldadd):
 * added to dy in the same pass — the straight-through gradient arriving through y_hat = ste_round(y - mu) + mu (CLC_run.py:571). */
int clc_gauss_lik_bwd(const float* dlik, int lddl, const float* y, int ldy, const float* mu, int ldmu,
                      const float* scale, int ldsc, const float* noise, int ldn, float* dy, int lddy, float* dmu,
                      int lddmu, float* dscale, int lddsc, long rows, int C, int mode, const float* dy_add, int ldadd,
                      clc_stream_t stream);

/* number of workgroup partials clc_gauss_lik_fwd writes for (rows, C) */
int clc_gauss_lik_partials(long rows, int C);

/* EntropyBottleneck factorised density (CLC_run.py:526-530). z/lik: [rows, C] NHWC slices;
 * matrices/biases/factors: HOST arrays of 5/5/4 device pointers to _matrix{k} [C,f(k+1),f(k)],
 * _bias{k} [C,f(k+1),1], _factor{k} [C,f(k+1),1]; quantiles [C,1,3].
 * mode 0 (train): v = z + noise; mode 1 (eval): v = round(z - median) + median. z_hat = round(z-med)+med. */
int clc_eb_lik_fwd(const float* z, int ldz, const float* noise, int ldn, const float* quantiles,
                   const float* const* matrices, const float* const* biases, const float* const* factors, float* lik,
                   int ldl, float* z_hat, int ldh, long rows, int C, int mode, clc_stream_t stream);
/* parameter gradients are written (not accumulated); dz may be NULL */
int clc_eb_lik_bwd(const float* dlik, int lddl, const float* z, int ldz, const float* noise, int ldn,
                   const float* quantiles, const float* const* matrices, const float* const* biases,
                   const float* const* factors, float* const* dmatrices, float* const* dbiases, float* const* dfactors,
                   float* dz, int lddz, long rows, int C, int mode, clc_stream_t stream);
/* aux loss (EntropyBottleneck.loss(), train_CLC.py:181): loss_partial[c] = sum_q |logits(quantiles[c,q]) - target[q]|,
 * dquantiles[c,q] = d loss / d quantiles (parameters are detached, as in the reference) */
int clc_eb_aux(const float* quantiles, const float* const* matrices, const float* const* biases,
               const float* const* factors, const float* target, float* loss_partial, float* dquantiles, int C,
               clc_stream_t stream);

/* quantize("symbols") + build_indexes (CLC_run.py:689-690): INT path, bit-exact */
int clc_quantize_build_indexes(const float* y, int ldy, const float* mu, int ldmu, const float* scale, int ldsc,
                               const float* scale_table, int n_scales, int32_t* symbols, int32_t* indexes,
                               float* y_hat, int ldh, long rows, int C, clc_stream_t stream);

/* rate / distortion reductions and their gradients (train_CLC.py:43-57). g_dev points at the upstream
 * gradient of the loss (a device scalar), so the backward never reads back to the host. */
int clc_log2_sum_partials(const float* x, int ld, long rows, int C, float* partials, int n_partials, clc_stream_t stream);
int clc_scaled_recip(const float* x, int ld, long rows, int C, const float* g_dev, float coef, float* out, int ldo,
                     clc_stream_t stream);                       /* out = g*coef / x   */
int clc_scaled_diff(const float* a, const float* b, long n, const float* g_dev, float coef, float* out,
                    clc_stream_t stream);                        /* out = g*coef*(a-b) */
/* sum of n floats (fixed order) -> out[0] (+)= scale * sum */
int clc_sum_partials(const float* partials, int n, float scale, float* out, int accumulate, clc_stream_t stream);
/* the scalar tail of RateDistortionLoss (train_CLC.py:43-59, MSE form) in one launch: the three fixed-order sums of the partials of
 * sum log2(lik_y), sum log2(lik_z), sum (x_hat - x)^2 and  bpp = (s_y + s_z) / neg_num_pixels, mse = sq / numel, loss = c * mse + bpp
 * (c = float(lmbda * 255^2)), each rounded as the reference's tensor expressions round; and the device scalars its backward hands to
 * clc_scaled_recip / clc_scaled_diff (g_* NULL: no gradient for that output). */
int clc_rd_combine(const float* py, int ny, const float* pz, int nz, const float* psq, int nsq, float neg_num_pixels, float numel, float c,
                   float* bpp, float* mse, float* loss, clc_stream_t stream);
int clc_rd_grad_scalars(const float* g_bpp, const float* g_mse, const float* g_loss, float neg_num_pixels, float numel, float c, float* g_logsum,
                        float* g_sq, clc_stream_t stream);
/* sum((a-b)^2) partials */
int clc_sqdiff_partials(const float* a, const float* b, long n, float* partials, int n_partials, clc_stream_t stream);

/* ---- Conditional Latent Matching ops (standalone module /root/reference/models/CLM.py) ---- *
 * All tensors NHWC fp32. Forward only (the reference module is an orphan that is never trained).
 * clc_clm_sim_colsum: colsum[b][q] = sum_p softmax_q( yt[b,p,:].yrt[b,q,:] / temperature )   (CLM.py:107-109,14-20)
 * clc_clm_scale_rows: out[r][c] = w[r] * x[r][c]                                             (CLM.py:16-22)
 * clc_clm_deform:     9-tap modulated bilinear sampling, zero outside the map               (CLM.py:35-60)
 * clc_clm_fuse:       out = sum_m softmax_m(att_m[r]) * feat_m[r][c] (* sigmoid(att_m[r]) if gate) + y[r][c]
 *                                                                                            (CLM.py:118-125, SimpleCLM :166-179) */
size_t clc_clm_sim_colsum_workspace_bytes(int B, int HW);
int clc_clm_sim_colsum(const float* yt, int ldy, const float* yrt, int ldr, int B, int HW, int C, float temperature,
                       float* colsum, void* ws, size_t ws_bytes, clc_stream_t stream);
int clc_clm_scale_rows(const float* x, int ldx, const float* w, float* out, int ldo, long rows, int C, clc_stream_t stream);
int clc_clm_deform(const float* x, int ldx, const float* offset, int ldo, const float* modulation, int ldm, float* out,
                   int ldy, int B, int H, int W, int C, clc_stream_t stream);
int clc_clm_fuse(const float* const* feats, const float* const* atts, int M, int ldf, int lda, const float* y, int ldy,
                 float* out, int ldo, long rows, int C, int gate, clc_stream_t stream);

/* ---- patch-matching side information (numeric core of /root/reference/models/Patch_Matching.py) ---- *
 * Planar NCHW fp32 (3-channel images), forward only.
 * clc_pm_prep:       rgb_transform(reduce_mean_and_std_normalize_images(x * in_scale))             (:913-934)
 * clc_pm_gauss_mask: create_gaussian_masks(img_h, img_w, ph, pw) -> [P, H-ph+1, W-pw+1]            (:779-807)
 * clc_pm_pearson:    L2_or_pearson_corr(q [P,C,ph,pw], y [C,H,W]) (* mask) -> [P, H-ph+1, W-pw+1]   (:854-910)
 * clc_pm_topk:       top-k positions per patch (values, flat indices), ties -> lowest index       (:105, :225)
 * clc_pm_gather:     SI_Wraper / SI_Finder patch gather + re-tiling; temperature < 0: plain argmax copy (k = 1) */
int clc_pm_prep(const float* x, float* out, int n_img, int H, int W, float in_scale, clc_stream_t stream);
int clc_pm_gauss_mask(float* out, int img_h, int img_w, int ph, int pw, clc_stream_t stream);
size_t clc_pm_pearson_workspace_bytes(int P, int C, int H, int W, int ph, int pw);
int clc_pm_pearson(const float* q, int P, const float* y, int C, int H, int W, int ph, int pw, const float* mask,
                   float* out, void* ws, size_t ws_bytes, clc_stream_t stream);
int clc_pm_topk(const float* corr, int P, int npos, int k, float* val, int32_t* idx, clc_stream_t stream);
int clc_pm_gather(const float* y, int C, int H, int W, int ph, int pw, const float* val, const int32_t* idx, int k,
                  float temperature, float* out, clc_stream_t stream);

/* ---- MS-SSIM distortion (pytorch_msssim.ms_ssim semantics; train_CLC.py:33-34,55-57) ---- *
 * NHWC fp32. One scale at a time: means[bc][0] = mean(cs map), means[bc][1] = mean(ssim map) over the valid 11x11
 * Gaussian (sigma 1.5) windows; bwd: dx = d(sum_bc g_means[bc][0]*mean_cs + g_means[bc][1]*mean_ssim)/dx
 * + 0.25 * dnext upsampled 2x (gradient arriving through the 2x2 average pool from the next coarser scale, or NULL).
 * clc_ssim_init() uploads the window once (call outside graph capture). */
int clc_ssim_init(void);
size_t clc_ssim_workspace_bytes(int B, int H, int W, int C);
int clc_ssim_scale_fwd(const float* x, int ldx, const float* y, int ldy, int B, int H, int W, int C, float data_range,
                       float* means, void* ws, size_t ws_bytes, clc_stream_t stream);
int clc_ssim_scale_bwd(const float* x, int ldx, const float* y, int ldy, int B, int H, int W, int C, float data_range,
                       const float* g_means, const float* dnext, float* dx, int lddx, void* ws, size_t ws_bytes,
                       clc_stream_t stream);
int clc_avgpool2(const float* x, int ldx, float* out, int B, int H, int W, int C, clc_stream_t stream);

/* ---- optimizer ------------------------------------------------------------------------ *
 * Multi-tensor AdamW + grad-norm clip + nan_to_num (train_CLC.py:164-179) over a flat
 * table of (param, grad, m, v, numel) entries resident on the device. */
typedef struct { float* p; float* g; float* m; float* v; long n; } clc_param_entry;
/* chunks_dev: n_chunks pairs (entry index, element offset); one workgroup per chunk of
 * clc_optim_chunk_elems() elements.  partials: n_chunks floats (sum (grad_scale * g)^2 per chunk).
 * grad_scale: 1 on one GPU; 1 / world after a SUMMED gradient all-reduce (the rank mean of run_ddp.sh:1-7 / DDP, folded into
 * the two passes that read the gradients anyway instead of a pass of its own over the arena). */
int clc_optim_chunk_elems(void);
int clc_grad_sqnorm_partials(const clc_param_entry* table_dev, const int32_t* chunks_dev, int n_chunks, float* partials,
                             float grad_scale, clc_stream_t stream);
/* state_dev[3] = {t, 1 - beta1^t, sqrt(1 - beta2^t)}: t += 1 and the two bias corrections of torch.optim.AdamW, evaluated in
 * double on the device (one thread) so the optimizer step stays graph-replayable. */
int clc_adam_tick(float* state_dev, double beta1, double beta2, clc_stream_t stream);
/* g <- nan_to_num(g * grad_scale * min(1, max_norm/(sqrt(*total_sqnorm_dev)+1e-6))) ; AdamW update with the step state
 * read from step_dev[3] (clc_adam_tick) and the learning rate from *lr_dev (device float) so a
 * captured launch is graph-replayable AND follows the MultiStepLR schedule (train_CLC.py:453,497). */
int clc_adamw_step(const clc_param_entry* table_dev, const int32_t* chunks_dev, int n_chunks, const float* total_sqnorm_dev,
                   float max_norm, const float* lr_dev, double beta1, double beta2, float eps, float weight_decay,
                   const float* step_dev, float grad_scale, clc_stream_t stream);
int clc_scalar_add(float* x_dev, float v, clc_stream_t stream);

/* ---- entropy coder (HOST, bit-exact) --------------------------------------------------- *
 * Replaces compressai.ans.BufferedRansEncoder/RansDecoder (CLC_run.py:658,712-713,762-763,793),
 * RansEncoder/RansDecoder.{encode,decode}_with_indexes behind EntropyBottleneck.compress/decompress
 * (CLC_run.py:643-644,749) and compressai._CXX.pmf_to_quantized_cdf (via update(), CLC_run.py:486-491). */
long clc_rans_encode_bound(long n_symbols);
long clc_rans_encode(const int32_t* symbols, const int32_t* indexes, long n, const int32_t* cdfs, int cdf_stride,
                     const int32_t* cdf_sizes, const int32_t* offsets, uint8_t* out, long out_cap); /* HOST */
typedef struct clc_rans_decoder clc_rans_decoder;
clc_rans_decoder* clc_rans_decoder_create(const uint8_t* stream, long nbytes);                    /* HOST, copies */
long clc_rans_decoder_decode(clc_rans_decoder* d, const int32_t* indexes, long n, const int32_t* cdfs, int cdf_stride,
                             const int32_t* cdf_sizes, const int32_t* offsets, int32_t* out);      /* HOST */
void clc_rans_decoder_destroy(clc_rans_decoder* d);
int clc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf_out /* n+1 */);  /* HOST */

/* ---- fused ResidualUnit (forward), round 3 ----
 * One launch for CompressAI's AttentionBlock.ResidualUnit as SWAtten uses it (/root/reference/models/CLC_run.py:222-244):
 *   y = relu(x + conv1x1(W3, relu(conv3x3(W2, relu(conv1x1(W1, x) + b1)) + b2)) + b3),  C -> C/2 -> C/2 -> C, on 16x16 maps, C = 128.
 * x [N,16,16,C] pixel-major (leading dimension ldx); t1, t2 [N,16,16,C/2] and y [N,16,16,C] dense outputs — t1 / t2 are the activated
 * outputs of the first two layers, which the ordinary gradient kernels (clc_conv2d transposed / clc_conv2d_wgrad) take as saved
 * activations, so the backward pass of the three layers is unchanged.  sets = 1, 2 or 4 filter sets on equal parts of the batch
 * (images [k N/sets, (k+1) N/sets) use set k), filters in clc_conv2d's layout ([Cout][kh][kw][Cin]).  A result depends on its own
 * image only (one workgroup per 8x4-pixel tile, fixed summation order). */
typedef struct {
  const float* x; int ldx;
  float* t1; float* t2; float* y;
  int N, H, W, C, sets;
  const float* w1[4]; const float* b1[4];
  const float* w2[4]; const float* b2[4];
  const float* w3[4]; const float* b3[4];
  const float* saved_y; const float* saved_t2; const float* saved_t1;   /* clc_residual_unit_dgrad only */
} clc_ru_desc;
int clc_residual_unit_fwd(const clc_ru_desc* d, clc_stream_t stream);
/* The unit's data gradient, one launch: with x = dy (leading dimension ldx), w1 / w2 / w3 = the TRANSPOSED filters of layers 3 / 2 / 1
 * (clc_filter_transpose: [64][128], [64][9][64], [128][64]; biases unused) and the forward pass's saved_y / saved_t2 / saved_t1 (dense),
 * writes  t1 <- g2 = d(layer-2 pre-activation),  t2 <- g1 = d(layer-1 pre-activation)  (the dy operands of those layers' filter
 * gradients; layer 3's is dy gated by saved_y, which clc_conv2d_wgrad applies itself) and  y <- dx  (including the identity branch). */
int clc_residual_unit_dgrad(const clc_ru_desc* d, clc_stream_t stream);

/* ---- fused Swin-block MLP, round 4 ----
 * `x + mlp(ln2(x))` of Block (/root/reference/models/CLC_run.py:185-187, 190-192) for the ConvTransBlocks' Swin blocks (trans_dim 64):
 *   clc_mlp_fwd   y = res + W2 . gelu(W1 . x + b1) + b2     x [M][64] (leading dimension ldx) -> y [M][64] (ldy), res optional (ldr);
 *                 w1 [256][64], w2 [64][256] (nn.Linear layout = clc_conv2d's 1x1 filter layout).  The 256-channel hidden tensor is
 *                 never written: it goes from the first GEMM's accumulator registers straight into the second GEMM's operands.
 *   clc_mlp_bwd   the block's whole DATA gradient from (x, dy) with the hidden tensor RECOMPUTED (fc1 again from the saved input):
 *                 dh [M][256] = (W2^T dy) . gelu'(h)  and  g [M][256] = gelu(h)  (dense; the dy / x operands of fc1's / fc2's
 *                 filter gradients, which stay on clc_conv2d_wgrad*), dx [M][64] (lddx) = W1^T dh.  w2t = clc_filter_transpose(w2)
 *                 ([256][64]).
 * Same summation order per output element and same epilogue expressions as the clc_conv2d launches they replace (fc1 + GELU with the
 * stored derivative, fc2 + residual; their transposed / out_gate forms backward): THE SAME BITS, so either path may serve the codec.
 * M = pixels (N * H * W of a pixel-major tensor), a multiple of 32.  Persistent workgroups, filters resident in LDS (129 KB). */
typedef struct {
  const float* x; int ldx;
  const float* w1; const float* b1;
  const float* w2; const float* b2;
  const float* res; int ldr;
  float* y; int ldy;
  long M; int Cin, Chid, Cout;          /* 64, 256, 64 */
  /* clc_mlp_bwd only */
  const float* dy; int lddy;
  const float* w2t;
  float* dx; int lddx;
  float* dh; float* g;
  /* optional, both directions: h [M][256] dense = the fc1 PRE-activation (W1 x + b1).  clc_mlp_fwd writes it when non-NULL (training,
   * "save" mode); clc_mlp_bwd then reads it instead of recomputing it (256 of a tile's 768 MFMAs for 134 MB of traffic per 8x128x128). */
  float* h;
  /* optional, both directions: the LayerNorm in front (Block.ln2, CLC_run.py:183,192).  ln_gamma / ln_beta [64] non-NULL: x is the block's RAW
   * input and the kernels compute  y = x + mlp(LN(x))  (res must be NULL — the residual is x; h must be NULL).  clc_mlp_fwd also writes LN(x)
   * to ln_out [M][64] dense when that is non-NULL (training).  clc_mlp_bwd reads it back (fc1's operand again; it is also the x operand of
   * fc1's filter gradient), needs a dense x, returns in dx the gradient of the whole expression (clc_layernorm_bwd's result with dy folded in
   * as dx_add) and leaves clc_mlp_blocks(M) partial rows [2][64] (dgamma, dbeta) in ln_ws for clc_partial_reduce_batched.  LN(x), y and dx
   * carry the bits of clc_layernorm_fwd / clc_layernorm_bwd around the plain form. */
  const float* ln_gamma; const float* ln_beta;
  float* ln_out; float* ln_ws;
} clc_mlp_desc;
int clc_mlp_fwd(const clc_mlp_desc* d, clc_stream_t stream);
int clc_mlp_bwd(const clc_mlp_desc* d, clc_stream_t stream);
int clc_mlp_blocks(long M);   /* workgroups (= partial rows in ln_ws) of a clc_mlp_bwd / clc_lnlin_bwd launch over M pixels */

/* ---- LayerNorm + Linear: ln1 and the window attention's embedding, round 4 ----
 * `self.embedding_layer(self.ln1(x))` of Block / WMSA (/root/reference/models/CLC_run.py:120, 141, 180, 191; nn.LayerNorm(64), nn.Linear(64, 192)):
 *   clc_lnlin_fwd   y [M][192] dense = W . LN(x) + b, x [M][64] (ldx: may be a channel range of a wider buffer); writes LN(x) to ln_out [M][64]
 *                   dense when non-NULL (training: the x operand of the Linear's filter gradient, which stays clc_conv2d_wgrad*).
 *   clc_lnlin_bwd   dx [M][64] (lddx) = LayerNorm gradient of (W^T dy) + dadd (the block's residual gradient, ldadd; optional), from dy [M][192]
 *                   dense, wt = clc_filter_transpose(w) ([64][192]) and the raw x; clc_mlp_blocks(M) partial rows [2][64] (dgamma, dbeta) in ln_ws
 *                   for clc_partial_reduce_batched.
 * y and dx carry the bits of clc_layernorm_fwd + clc_conv2d (1x1) / clc_conv2d (transposed) + clc_layernorm_bwd.  M a multiple of 32. */
typedef struct {
  const float* x; int ldx;
  const float* ln_gamma; const float* ln_beta;
  const float* w; const float* b;
  float* y; float* ln_out;
  long M; int Cin, Cout;                /* 64, 192 */
  /* clc_lnlin_bwd only */
  const float* dy; const float* wt;
  float* dx; int lddx;
  const float* dadd; int ldadd;
  float* ln_ws;
} clc_lnlin_desc;
int clc_lnlin_fwd(const clc_lnlin_desc* d, clc_stream_t stream);
int clc_lnlin_bwd(const clc_lnlin_desc* d, clc_stream_t stream);

/* ---- GDN / IGDN data gradient in one launch, round 4 ----
 * CompressAI GDN (y = x * (beta + gamma . x^2)^-1/2, inverse: ^+1/2; g_a / g_s, /root/reference/models/CLC_run.py:296-318, 337-353) on a dense
 * 128-channel map with >= 32 768 pixels: from dy, x and the saved norm v (clc_conv2d's y_pre) -> dv [M][128] = d(loss)/d(v) (the dy operand of
 * gamma's filter gradient) and dx [M][128] = dy * v^-+1/2 + 2 x (gamma_eff^T dv).  Replaces clc_gdn_bwd_elem + the transposed 1x1 clc_conv2d with
 * norm = CLC_NORM_MUL2 (their bits), without the dx_direct tensor.  gamma_eff_t = the [Cin][Cout] image clc_gdn_reparam_fwd writes. */
int clc_gdn_bwd_fused(const float* dy, const float* x, const float* v, const float* gamma_eff_t, float* dv, float* dx, long M, int C, int inverse,
                      clc_stream_t stream);

/* ---- reference-retrieval feature extractor: the pooling layers of torchvision's ResNet50 as the reference uses it ----
 * clc_maxpool2d        nn.MaxPool2d(ks, stride, pad) of resnet50.maxpool (/root/reference/dataloader_ref_cluster.py:41-44, dataloader_CLC.py:275),
 *                      pixel-major in / out, C and the leading dimensions multiples of 4.
 * clc_adaptive_pool2d  F.adaptive_avg_pool2d(x, (1, 1)) (resnet50.avgpool) and the spatial-pyramid levels
 *                      F.adaptive_max_pool2d(x, (L, L)), L = 1, 2, 4 (/root/reference/dataloader_CLC.py:250-256, 282-286); out = [N][C][L][L]
 *                      floats in NCHW order (the order h.view(N, -1) flattens). */
int clc_maxpool2d(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int ks, int stride, int pad, int OH, int OW, clc_stream_t stream);
int clc_adaptive_pool2d(const float* x, int ldx, float* out, int N, int H, int W, int C, int L, int is_max, clc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CLC_HIP_H */
