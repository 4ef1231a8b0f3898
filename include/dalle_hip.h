/* dalle_hip.h -- C ABI of libdalle_hip.so: the drop-in boundary for the DALL-E train-step hot path
 * on MI355X (gfx950).  SURVEY.md §8(b): the reference (EleutherAI/DALLE-mtf) has NO FFI boundary;
 * its arithmetic lives in mesh_tensorflow / tensorflow ops.  Each entry point below replaces one
 * implicit third-party op of SURVEY.md §2.2 (K-rows) at the reference call site cited.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers + explicit sizes; the caller (PyTorch host code) owns every buffer incl.
 *     workspace; nothing is allocated, freed or synchronised inside; everything is enqueued on the
 *     hipStream_t passed last (as void*), so calls are stream-ordered and re-entrant.
 *   - return 0 on success, negative dmi_status otherwise; dmi_last_error_string() is thread-local.
 *   - "bf16" = raw uint16 bfloat16 bits.  Matrices are row-major with explicit leading dimensions.
 *   - weights keep the reference layout [in, out]; the fwd GEMM consumes the
 *     [out, in] bf16 copy produced by dmi_transpose_bf16 / dmi_adam_step.
 */
#ifndef DALLE_HIP_H
#define DALLE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DMI_OK = 0,
  DMI_ERR_INVALID = -1,   /* bad size / alignment / null pointer */
  DMI_ERR_LAUNCH = -2,    /* hipGetLastError() after a launch */
  DMI_ERR_UNSUPPORTED = -3
} dmi_status;

const char* dmi_last_error_string(void);
int dmi_version(void);
/* Test hooks (every setting computes the same results): "nt4" 0/1/2 = never / auto / always use the 256x128 NT tile,
 * "nt8" / "tn8" 0/1/2 = the 256x256 8-wave NT / weight-gradient tiles, "tn_tail" 0/1 = row-split the ragged last residency of
 * unsplit weight gradients, "nt8_min_k" = smallest K at which the auto mode picks the 256x256 NT tile, "skinny" 0/1 = products with M <= 32 rows (the decode step) on the weight-streaming kernel,
 * "attn_xcd" = schedule of the persistent attention blocks (0: one serpentine over all blocks,
 * != 0: per-XCD item lists when the (batch, head) count divides by 8).  Unknown name -> -1. */
int dmi_get_option(const char* name);
int dmi_set_option(const char* name, int value);
/* Diagnostics (tools/phases.py): u64 device buffer [blocks][5 or 8] that the 256x128 NT kernel and the weight-gradient
 * kernel fill with per-block phase cycle stamps; NULL (default) disables the stamps. */
int dmi_set_debug_buffer(void* device_buffer);

/* ---- K1  embedding: mtf.gather(wte, tokens) + wpe[0..S)   src/dalle_mtf/models.py:186-219 ---- */
/* pos_dev (optional, device int32): every row takes wpe[*pos_dev] (clamped to [0, S)) instead of wpe[row % S] -- the incremental
 * decode step, whose position must not be a by-value argument of a replayed HIP graph. */
int dmi_embed_fwd(const int32_t* tokens, const uint16_t* wte, const uint16_t* wpe, uint16_t* x,
                  int64_t rows /*B*S*/, int S, int d, int vocab, const int* pos_dev, void* stream);
/* stable sort of the n token ids (clamped to [0, vocab)): sorted_tokens ascending, perm[i] = source position of sorted
 * position i.  Index plumbing for dmi_embed_bwd; one-block radix sort, deterministic.
 * workspace: dmi_sort_tokens_workspace_bytes(n). */
int64_t dmi_sort_tokens_workspace_bytes(int64_t n);
int dmi_sort_tokens(const int32_t* tokens, int32_t* sorted_tokens, int32_t* perm, int64_t n, int vocab,
                    void* workspace, void* stream);
/* backward: dwpe[s] = sum_b dx[b,s];  dwte[tok] = sum of dx over the positions holding tok (gradient of mtf.gather =
 * scatter-add), written for every id incl. zeros for absent ones.  Positions are visited in
 * token-id order (dmi_sort_tokens) so equal ids reduce in registers; deterministic, no atomics.
 * workspace: dmi_embed_bwd_workspace_bytes(B, S, d). */
int64_t dmi_embed_bwd_workspace_bytes(int B, int S, int d);
int dmi_embed_bwd(const int32_t* sorted_tokens, const int32_t* perm, const uint16_t* dx, float* dwte,
                  float* dwpe, int B, int S, int d, int vocab, void* workspace, void* stream);

/* ---- K2  LayerNorm eps=1e-5 biased variance   models.py:373-389, layers.py:30-33 ---- */
int dmi_layernorm_fwd(const uint16_t* x, const uint16_t* g, const uint16_t* b, uint16_t* y,
                      float* mean, float* rstd, int64_t rows, int d, float eps, void* stream);
/* dx_out = LNbwd(dy) (+ dres if non-null); dg/db fp32 [d] written (not accumulated).
 * workspace: dmi_layernorm_bwd_workspace_bytes(rows, d). */
int64_t dmi_layernorm_bwd_workspace_bytes(int64_t rows, int d);
int dmi_layernorm_bwd(const uint16_t* dy, const uint16_t* x, const uint16_t* g, const float* mean,
                      const float* rstd, const uint16_t* dres, uint16_t* dx, float* dg, float* db,
                      void* workspace, int64_t rows, int d, void* stream);
/* dg == db == NULL defers the reduce of the gain / bias gradients: the per-block partials stay in `workspace` (which the caller
 * then must not reuse) until dmi_layernorm_bwd_finish_batch reduces up to 16 LayerNorms of the same width in one launch (same
 * summation order: bit-identical to the immediate form).  Host arrays of n device pointers / row counts. */
int dmi_layernorm_bwd_finish_batch(const void* const* workspaces, float* const* dgs, float* const* dbs, const int64_t* rows,
                                   int n, int d, void* stream);

/* ---- K3/K5/K6/K7  dense layers: mtf einsum / mtf.layers.dense   models.py:242-244,303-311,320-321,369,393
 * C[M,N] = A[M,K] . Bt[N,K]^T  (both operands K-contiguous), bf16 in, fp32 accumulate on MFMA.
 * flags: DMI_GEMM_*;  bias bf16 [N];  residual bf16 [M,ldc] added;  relu_src bf16 [M,ldc]: C *= (relu_src>0);
 * rowscale fp32 [M]: C[m,:] *= rowscale[m].  K % 64 == 0, N % 8 == 0.  Output bf16 (or fp32 with DMI_GEMM_OUT_F32).
 * Supported flag sets: 0, BIAS, BIAS|RELU, BIAS|RESIDUAL, RESIDUAL, RELU_MASK, ROWSCALE, OUT_F32. */
#define DMI_GEMM_BIAS 1
#define DMI_GEMM_RELU 2
#define DMI_GEMM_RESIDUAL 4
#define DMI_GEMM_RELU_MASK 8
#define DMI_GEMM_OUT_F32 16
#define DMI_GEMM_ROWSCALE 32
int dmi_gemm_nt(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, void* C, int ldc,
                int M, int N, int K, int flags, const uint16_t* bias, const uint16_t* residual,
                const uint16_t* relu_src, const float* rowscale, void* stream);
/* LayerNorm (eps, biased variance: K2) fused into the product, for the incremental decode step only (M <= 32 rows, N % 16 == 0,
 * K = the normalised width <= 2048; anything else -> DMI_ERR_UNSUPPORTED): C = LN(X; gamma, beta) . Bt^T (+ bias bf16 [N])(ReLU).
 * flags: 0, BIAS, BIAS|RELU.  LN(X) is rounded to bf16 before the product, as dmi_layernorm_fwd + dmi_gemm_nt would. */
int dmi_ln_gemm_nt(const uint16_t* X, int ldx, const uint16_t* gamma, const uint16_t* beta, float eps, const uint16_t* Bt, int ldb,
                   uint16_t* C, int ldc, int M, int N, int K, int flags, const uint16_t* bias, void* stream);
/* same product with the K range split over nsplit block groups (fp32 slabs in workspace, deterministic reduction to
 * bf16 C with ldc == N, optionally times rowscale[m]): for long-K GEMMs whose tile count does not fill whole
 * residencies (the head's input gradient: K = vocabulary). */
int64_t dmi_gemm_nt_splitk_workspace_bytes(int M, int N, int nsplit);
int dmi_gemm_nt_splitk(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int M, int N, int K,
                       int nsplit, const float* rowscale, void* workspace, void* stream);
/* weight gradient: dW[I,J] (fp32, ld = J) = sum_m X[m,I] * dY[m,J]; deterministic split over m.
 * dbias (nullable): fp32 [J] = sum_m w[m] * dY[m, :] fused into the same pass (bias gradient of the dense layer), with
 * w = 1, or w = bias_weights (nullable bf16 [M], 4-byte aligned, M even: DMI_ERR_INVALID otherwise) for the fused-softmax head
 * where dY holds unnormalised dlogits.
 * workspace: dmi_gemm_tn_workspace_bytes(M, I, J).
 * Run-to-run bit-identical for every shape.  Unsplit gradients with at least as many 256-column stripes as the chip has gangs
 * of ceil(I / 128) block slots (the vocabulary projection's gradient) run as a gang stream-K on 128 x 256 tiles: a column
 * stripe is cut at most once over m and its two pieces are added onto zeroed elements with fp32 atomics -- a two-addend sum
 * has no order (option "tn_wide" = 0 keeps those shapes on the 128 x 128 tiles). */
int64_t dmi_gemm_tn_workspace_bytes(int M, int I, int J);
/* deferred (nullable, room for 2 items) + n_deferred: when the gradient is split over m, its final slab reduces (dW, dbias)
 * are not launched but described here; the caller runs the reduces of several GEMMs in ONE launch with
 * dmi_reduce_slabs_batch (<= 16 items) -- each deferring GEMM then needs its own workspace, untouched until that call.
 * out[i] = sum_{s < nsplit} slabs[s * 4 n4 + i], i < 4 n4, fixed order: bit-identical to the undeferred call. */
typedef struct dmi_reduce_item { const float* slabs; float* out; int nsplit; int64_t n4; } dmi_reduce_item;
int dmi_gemm_tn(const uint16_t* X, int ldx, const uint16_t* dY, int ldy, float* dW, float* dbias,
                const uint16_t* bias_weights, int M, int I, int J, void* workspace, dmi_reduce_item* deferred,
                int* n_deferred, void* stream);
int dmi_reduce_slabs_batch(const dmi_reduce_item* items, int n, void* stream);
/* Up to 4 weight gradients that contract over the same M rows in ONE launch (the gradients of a block's out-projection and QKV
 * kernels, src/dalle_mtf/models.py:242-244,303-311 under mtf.gradients, src/optimizers.py:34: 16 + 48 tiles fill the chip together,
 * neither does alone).  Per problem: dW[I,J] = X^T dY (+ dbias / bias_weights as dmi_gemm_tn) with its own workspace of
 * dmi_gemm_tn_workspace_bytes(M, I, J) bytes.  The slab reduces (<= 2 per problem) go to `deferred` / *n_deferred, or run here
 * when deferred == NULL.  The row-split plan is the group's, so sums are ordered differently from n single dmi_gemm_tn calls
 * (fp32, deterministic). */
typedef struct dmi_tn_problem { const uint16_t* X; int ldx; const uint16_t* dY; int ldy; float* dW; float* dbias;
                                const uint16_t* bias_weights; int I; int J; void* workspace; } dmi_tn_problem;
int dmi_gemm_tn_group(const dmi_tn_problem* probs, int n, int M, dmi_reduce_item* deferred, int* n_deferred, void* stream);
/* The group's plan for n problems of shapes I[k] x J[k] over M rows: the row-split count (>= 2) when the launch runs on 128 x 256
 * tiles -- the union of those tiles fills the block slots with a split count every problem's own workspace has slabs for: all four
 * gradients of an n_embd = 512 block, 96 tiles x 5 splits --, 0 when it stays on 128 x 128 tiles.  Callers group MORE problems per
 * launch where this is non-zero (the engine: a block's two FFN gradients join the attention pair). */
int dmi_gemm_tn_group_plan(const int* I, const int* J, int n, int M);

/* column sum (bias gradients): out[N] fp32 = sum_m Y[m, 0..N) ; workspace dmi_colsum_workspace_bytes */
int64_t dmi_colsum_workspace_bytes(int64_t M, int N);
int dmi_colsum(const uint16_t* Y, int ldy, float* out, int64_t M, int N, void* workspace, void* stream);
/* batched bf16 transpose: in [batch, R, C] -> out [batch, C, R] */
int dmi_transpose_bf16(const uint16_t* in, uint16_t* out, int batch, int R, int C, void* stream);

/* ---- K4  causal attention, UNSCALED logits, fp32 softmax   models.py:221-227,292-299 (Appendix A.2/A.3)
 * qkv [B*S, 3*H*128] bf16 = the QKV projection output, row = [q | k | v] x [H, 128] (heads-major, A.1);
 * o [B*S, H*128] bf16;  lse [B,H,S] fp32.  head dim is fixed at 128 (README.md:164); S % 8 == 0.
 * Every transposed MFMA operand is a hardware transpose read of the natural tile: no transposed copies. */
int dmi_attention_fwd(const uint16_t* qkv, uint16_t* o, float* lse, int B, int H, int S, void* stream);
/* backward.  d_o [B*S, H*128] bf16; scratch: 3*B*H*S floats (delta | interleaved (lse, delta) pairs);
 * dqkv [B*S, 3*H*128] bf16 in the qkv layout (what the QKV dgrad/wgrad GEMMs consume). */
int dmi_attention_bwd(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, float* scratch,
                      uint16_t* dqkv, int B, int H, int S, void* stream);

/* Incremental decode (the reference's unfinished is_incremental_inference path, src/dalle_mtf/models.py:246-254,281-285):
 * one query position `pos` against the key/value cache.  qkv = the forward pass's projection buffer [B*S, 3*H*128] used as the
 * cache; row b*S + pos must already hold q | k | v of the new position (the caller's QKV GEMM writes it in place with row
 * pitch S*3d).  o[b, h*128 ..] = softmax(q . K[0..pos]^T) V[0..pos], bf16 [B, H*128].  Unscaled logits, keys <= pos only.
 * Graph-replayable form: fresh != NULL = a [B, 3*H*128] staging buffer at a fixed address holding the step's q | k | v (the
 * QKV GEMM writes it instead of the cache row); the kernel copies it into cache row pos and attends.  pos_dev != NULL = the
 * position is read from device memory (one int32; `pos` is then ignored and a position outside [0, S) makes the launch a
 * no-op), so ONE captured HIP graph serves every step of the sampling loop. */
int dmi_attention_decode(uint16_t* qkv, const uint16_t* fresh, uint16_t* o, int B, int H, int S, int pos, const int* pos_dev,
                         void* stream);

/* ---- K7/K8  to_logits + cross entropy, labels = shift(tokens)   models.py:391-395,348-359,407-410
 * labels[t] = tokens[t+1], last = eos (bit-exact int path). */
int dmi_shift_labels(const int32_t* tokens, int32_t* labels, int B, int S, int eos, void* stream);
/* (a) evaluation / logits-returning path: cross entropy over materialised bf16 logits z [M, ldz] (ldz >= V, pad columns
 * must hold a large negative value).  loss_rows[M] fp32 = lse - z[label]; if dz_scale != 0: z is overwritten IN PLACE by
 * dz = (softmax(z) - onehot(label)) * dz_scale. */
int dmi_cross_entropy(uint16_t* z, int ldz, const int32_t* labels, float* loss_rows, float* lse,
                      int64_t M, int V, float dz_scale, void* stream);
/* (b) training path: the softmax is fused into the vocabulary projection and its row normaliser is deferred, so the
 * logits never make a round trip through HBM:
 *   dmi_label_logit      zl[m] = X[m,:] . Wt[label[m],:] + bias[label[m]] (fp32): the loss needs it; clears flag[0].
 *   dmi_gemm_nt_softmax  E[m,n] = bf16(exp(X[m,:] . Wt[n,:] + bias[n] - rowshift[m])) (rowshift nullable = no shift) and
 *                        rowsum_part[n/64][m] = fp32 sum of those exponentials over the 64-column group
 *                        (dmi_gemm_nt_softmax_partials(N) groups; pad columns need bias << 0 so that they contribute 0).
 *   dmi_softmax_finish   S[m] = sum of the partials; loss_rows[m] = log S[m] + rowshift[m] - label_logit[m] (= logsumexp - label logit);
 *                        rowscale[m] = dz_scale / S[m] (+ its bf16 copy); E[m,label] -= S[m], so dlogits = rowscale[m] * E[m,:];
 *                        Xs[M,K] = bf16(rowscale[m] * X[m,:]).  Consumers: dX = dmi_gemm_nt(E, W, DMI_GEMM_ROWSCALE),
 *                        dW = dmi_gemm_tn(Xs, E, bias_weights = rowscale_bf16).  With dz_scale == 0 only loss_rows is written.
 *                        Rows whose sum overflowed or vanished (a logit beyond +-87 of the shift) are redone exactly with the
 *                        row maximum as the shift (needs X, Wt, bias again); flag[0] != 0 afterwards tells that it happened. */
int dmi_label_logit(const uint16_t* X, int ldx, const uint16_t* Wt, int ldw, const uint16_t* bias,
                    const int32_t* labels, float* zl, int32_t* flag, int64_t M, int K, int V, void* stream);
int64_t dmi_gemm_nt_softmax_partials(int N);
/* Product + bias + residual with the LayerNorm of the result fused into the epilogue (reference src/dalle_mtf/models.py:303-314
 * feeding :333,373-389 -- out-projection + residual -> norm_2; :321-324 feeding the next block's :330 / to_logits' :392 -- FFN-2 +
 * residual -> norm_1 / final norm):  C[M, N] = bf16(A . Bt^T + bias + residual)  (bias, residual nullable),
 * Y[M, N] = bf16((C - mean) * rstd * gamma + beta), mean / rstd fp32 [M] over the ROUNDED row (biased variance, eps inside the
 * rsqrt: what dmi_layernorm_fwd computes from the stored C).  N = 512 only (a block owns whole rows): DMI_ERR_UNSUPPORTED otherwise. */
int dmi_gemm_nt_ln(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int ldc, int M, int N, int K,
                   const uint16_t* bias, const uint16_t* residual, const uint16_t* gamma, const uint16_t* beta, float eps,
                   uint16_t* Y, int ldy, float* mean, float* rstd, void* stream);
/* The FFN's ReLU mask as one bit per element (reference src/dalle_mtf/models.py:320-321: h = mtf.relu(dense(x)), and its backward
 * dh = (dx . W2^T) * (h > 0)):  dmi_gemm_nt_relu_bits: C = bf16(relu(A . Bt^T + bias)) and bits = (C > 0), one bit per output;
 * dmi_gemm_nt_mask_bits: C = bf16(A . Bt^T) * bit -- what dmi_gemm_nt(DMI_GEMM_RELU_MASK, relu_src = h) computes, bit for bit,
 * from M*N/8 bytes instead of the 2*M*N bytes of h.  `bits` is an opaque buffer of dmi_relu_bits_bytes(M, N) bytes (8-byte
 * aligned; the layout is private to the two calls).  N % 64 == 0, K % 128 == 0, operands below 2 GiB: DMI_ERR_UNSUPPORTED
 * otherwise.  dmi_relu_bits_auto: 1 where the library's own dispatch would run these shapes on the kernel that has the bit
 * forms (callers keep dmi_gemm_nt elsewhere). */
int64_t dmi_relu_bits_bytes(int M, int N);
int dmi_relu_bits_auto(int M, int N, int K);
int dmi_gemm_nt_relu_bits(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int ldc, int M, int N, int K,
                          const uint16_t* bias, void* bits, void* stream);
int dmi_gemm_nt_mask_bits(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, uint16_t* C, int ldc, int M, int N, int K,
                          const void* bits, void* stream);
/* An input-gradient product whose result arrives at a LayerNorm, with that LayerNorm's BACKWARD fused into the epilogue (reference:
 * the backward of src/dalle_mtf/layers.py:30-33 + models.py:387-388 behind models.py:330 / :333 -- norm_1 <- QKV, norm_2 <- FFN-1):
 *   dy = bf16(A . Bt^T) [M, N];  xh = (x - mean) * rstd;  dx[M, N] = bf16(rstd * (dy*gamma - mean_n(dy*gamma) - xh * mean_n(dy*gamma*xh)) + dres)
 * (dres nullable), i.e. what dmi_gemm_nt followed by dmi_layernorm_bwd computes, without writing / re-reading dy; the gain / bias
 * gradients leave as dmi_gemm_nt_lnbwd_parts(M) partial rows [2 N] fp32 in `part` (dgamma | dbeta), summed by
 * dmi_layernorm_bwd_finish_parts (fixed order).  N = 512 only (a block owns whole rows): DMI_ERR_UNSUPPORTED otherwise.  x, dres, dx have
 * row pitch N.  Differs from the two-kernel form only in the summation order of the reductions.
 * B2 / C2 (nullable, both or neither): the product that consumes dx chained in the same launch, C2[M, N] = bf16(dx . B2^T) with
 * B2 [N, ldb2] (K-contiguous) -- norm_2's dx is the gradient of the attention branch's output, and its next consumer is the
 * out-projection's input gradient d_o = dx . Wo^T (the backward of src/dalle_mtf/models.py:303-311): a block owns whole rows of
 * dx, i.e. the whole contraction range of its rows, so it reads them back from L2 as the A operand of a second main loop.
 * Bit-identical to dmi_gemm_nt on the stored dx. */
int dmi_gemm_nt_lnbwd_parts(int M);
/* dmi_gemm_nt_ln_auto: 1 where dmi_gemm_nt_ln / dmi_gemm_nt_lnbwd accept a K-contiguous [M, K] x [N, K]^T product (N = 512, operands
 * inside the kernel's 32-bit offsets) and no CUs are reserved for a concurrent exchange; callers gate the fused forms on it when they
 * size their buffers and keep dmi_gemm_nt + dmi_layernorm_fwd / _bwd elsewhere (same role as dmi_relu_bits_auto). */
int dmi_gemm_nt_ln_auto(int M, int N, int K);
int dmi_gemm_nt_lnbwd(const uint16_t* A, int lda, const uint16_t* Bt, int ldb, int M, int N, int K, const uint16_t* x,
                      const uint16_t* gamma, const float* mean, const float* rstd, const uint16_t* dres, uint16_t* dx,
                      float* part, const uint16_t* B2, int ldb2, uint16_t* C2, void* stream);
int dmi_layernorm_bwd_finish_parts(const float* part, int P, float* dg, float* db, int d, void* stream);
int dmi_gemm_nt_softmax(const uint16_t* X, int ldx, const uint16_t* Wt, int ldw, const uint16_t* bias,
                        const float* rowshift, uint16_t* E, int lde, float* rowsum_part, int M, int N, int K, void* stream);
int dmi_softmax_finish(const float* rowsum_part, int nparts, const float* label_logit, const float* rowshift,
                       const int32_t* labels, const uint16_t* X, int ldx,
                       const uint16_t* Wt, int ldw, const uint16_t* bias, uint16_t* E, int lde, int N,
                       float* loss_rows, float* rowscale, uint16_t* rowscale_bf16, uint16_t* Xs, int32_t* flag,
                       int64_t M, int K, int V, float dz_scale, void* stream);
/* out[0] = scale * sum(x[0..n))  deterministic single-block reduce (loss mean; grad-norm finish) */
int dmi_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream);

/* ---- K10  image-token indexing   src/model_fns.py:76-77,118-119 (bit-exact)
 * tokens_out[b, 0:T] = text[b]; tokens_out[b, T + p] = argmax_c logits[b, p, c] (first max) + text_vocab */
int dmi_assemble_tokens(const int32_t* text, const float* vae_logits, int32_t* tokens_out,
                        int B, int T, int P, int C, int text_vocab, void* stream);

/* Sampling of the next image token from head logits (the sampler around the reference's unfinished incremental-inference path,
 * src/dalle_mtf/models.py:246-254,281-285; src/model_fns.py:135-136 raises).  Per row b of z bf16 [B, ldz] (first nv columns;
 * bias bf16 [nv] optional, added in fp32): v = (z + bias) / temperature; top_k > 0 keeps v >= the k-th largest v (ties kept); the choice is a
 * categorical draw from softmax(v) over the kept entries (Gumbel-max with counter-based noise hash(seed, position, b, i):
 * reproducible, stateless); temperature <= 0: first maximum.  nv <= 8192.
 * next_tok[b] = token_offset + choice (optional: where the next decode step reads its token);
 * out[b, position - out_col0] = choice when that column lies in [0, out_ld) (optional).
 * params_dev (optional, device uint32[4] = {bits of float 1/temperature (0: greedy), top_k, seed low, seed high}) and pos_dev
 * (optional, device int32) override the by-value arguments: decode step + sampling replay as one HIP graph.  advance != 0
 * (needs pos_dev = device int32[2], [1] a zero-initialised scratch counter): the last block to finish stores position + 1. */
int dmi_sample_tokens(const uint16_t* z, int ldz, const uint16_t* bias, int B, int nv, float temperature, int top_k,
                      uint64_t seed, const uint32_t* params_dev, int pos, int32_t* pos_dev, int advance, int token_offset,
                      int32_t* next_tok, int32_t* out, int out_ld, int out_col0, void* stream);

/* "Go to full precision for the logits" (src/dalle_mtf/models.py:394-395) for a slice of the head's output:
 * out[b, i] = float(z[b, i]) + float(bias[i]), z bf16 [B, ldz] (first nv columns), bias bf16 [nv] (nullable), out fp32 [B, nv]. */
int dmi_logits_f32(const uint16_t* z, int ldz, const uint16_t* bias, float* out, int B, int nv, void* stream);

/* ---- K9  clip_by_global_norm + AdamWeightDecayOptimizer   src/optimizers.py:11-16,82-89,154-177
 * sumsq: out[0] = sum g^2 (deterministic two-stage; workspace dmi_sumsq_workspace_bytes(n)).
 * adam: mult = clip>0 ? clip/max(sqrt(*gnorm_sq),clip) : 1;  g*=mult; m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2;
 *       p -= lr*(m/(sqrt(v)+eps) + wd*p)   (NO bias correction); p_bf16 (optional) = bf16(p).
 * tf_adam=1 switches to tf.train.AdamOptimizer semantics (bias-corrected lr_t passed as lr, eps as given,
 * grad_scale multiplies g first: CrossShardOptimizer mean) -- src/model_fns_tf.py:58-61.
 * lr_dev (nullable): when given, the kernel reads the learning rate from lr_dev[0] (DEVICE memory) and ignores `lr` -- the
 * per-step schedule value (src/optimizers.py:46-76) can then change between replays of a captured HIP graph. */
int64_t dmi_sumsq_workspace_bytes(int64_t n);
int dmi_sumsq(const float* g, int64_t n, float* out, void* workspace, void* stream);
int dmi_adam_step(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, int64_t n,
                  const float* gnorm_sq, float clip, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float grad_scale, const float* lr_dev, void* stream);

/* ---- data-parallel exchange: the all-reduce mtf inserts for `layout: batch_dim:data`   src/model_fns.py:81-82,189,
 * VAE src/model_fns_tf.py:61 (CrossShardOptimizer).  One process per GPU, RCCL over xGMI, bound at run time
 * (dmi_comm_load(path or NULL) -- call it first when the process already maps a specific librccl, e.g. PyTorch's).
 * Rank 0 draws dmi_comm_unique_id (HOST buffer of dmi_comm_unique_id_bytes()), the caller ships the bytes to all ranks
 * out of band, every rank calls dmi_comm_init (blocking collective; the current HIP device is the rank's GPU).
 * dmi_allreduce_bucket: g[0..n) <- sum over ranks (fp32, in place) enqueued on `stream` -- the caller orders it after the
 * bucket's last gradient kernel with an event and joins the stream before clip + Adam. */
int dmi_comm_load(const char* librccl_path);
int dmi_comm_unique_id_bytes(void);
int dmi_comm_unique_id(void* id_out);
int dmi_comm_init(void** comm_out, int nranks, int rank, const void* unique_id);
int dmi_comm_destroy(void* comm);
int dmi_allreduce_bucket(void* comm, float* g, int64_t n, void* stream);
int dmi_comm_broadcast_f32(void* comm, float* buf, int64_t n, int root, void* stream);

/* fp32 -> bf16 cast (initial weight export) */
int dmi_cast_f32_bf16(const float* in, uint16_t* out, int64_t n, void* stream);

/* n independent [R,C] -> [C,R] bf16 transposes between two buffers in one launch: table[i] = {in_off, out_off, R, C,
 * first_tile} (int64, elements / 64x64 tiles, first_tile ascending; R, C multiples of 8); total_tiles = sum of
 * ceil(R/64)*ceil(C/64).  The per-step refresh of the forward GEMMs' [out,in] weight copies (replaces what mtf's
 * einsum lowering does implicitly for y = x W, reference src/dalle_mtf/models.py:361-371). */
int dmi_transpose_bf16_batch(const uint16_t* in_base, uint16_t* out_base, const int64_t* table, int n,
                             int64_t total_tiles, void* stream);

/* in [R_valid, C] -> out [C, R_pitch] with columns [R_valid, R_pitch) zero (K-padding of transposed conv kernels) */
int dmi_transpose_bf16_padded(const uint16_t* in, uint16_t* out, int R_valid, int R_pitch, int C, void* stream);

/* ================= discrete VAE (src/vae_tf/models.py:81-163, src/vae_tf/layers.py:4-25) =================
 * K11 convolutions = implicit-im2col GEMMs (dmi_conv_gemm_nt / dmi_conv_wgrad_tn) for 64-channel-aligned layers; the
 * tap-list dmi_im2col + dmi_gemm_nt / dmi_gemm_tn path serves the 3-channel input layer and is the tested reference.
 * activations NHWC bf16 [B*H*W, C], C % 8 == 0. */
/* Implicit-im2col convolution: dmi_im2col + dmi_gemm_nt in one kernel, no column matrix in HBM, bit-identical results.
 *   out[(b,oy,ox)][n] = sum_t sum_c x[b, oy*stride + dy[t], ox*stride + dx[t], c] * Wt[n][t*C + c]  (+ epilogue flags)
 * x NHWC bf16 [B,H,W,C] with C % 64 == 0 (zero outside the image = TF SAME padding, src/vae_tf/models.py:67-68);
 * Wt [N, ldw] (ldw >= ntaps*C), out [B*Ho*Wo, ldc]; flags/bias/residual/relu_src as dmi_gemm_nt. */
int dmi_conv_gemm_nt(const uint16_t* x, int B, int H, int W, int C, int Ho, int Wo, int stride, int ntaps,
                     const int* dy, const int* dx, const uint16_t* Wt, int ldw, uint16_t* out, int ldc, int N,
                     int flags, const uint16_t* bias, const uint16_t* residual, const uint16_t* relu_src, void* stream);

/* Implicit-im2col weight gradient: dW[(t*C + c)][n] = sum_{b,oy,ox} x[b, oy*stride+dy[t], ox*stride+dx[t], c] * dY[(b,oy,ox)][n]
 * (+ dbias[n] = column sums of dY, nullable) = dmi_im2col + dmi_gemm_tn without the column matrix; lands in the TF kernel
 * layout [kh*kw*Cin, Cout] of tf.layers.conv2d (src/vae_tf/models.py:67).  C % 64 == 0, Ho and Wo powers of two.
 * workspace >= dmi_conv_wgrad_tn_workspace_bytes(B*Ho*Wo, ntaps*C, N).  deferred / n_deferred as dmi_gemm_tn: the final slab
 * reduces are handed back for one dmi_reduce_slabs_batch launch (the workspace then belongs to this call until that launch). */
int64_t dmi_conv_wgrad_tn_workspace_bytes(int M, int K, int N);
int dmi_conv_wgrad_tn(const uint16_t* x, int B, int H, int W, int C, int Ho, int Wo, int stride, int ntaps,
                      const int* dy, const int* dx, const uint16_t* dY, int ldy, int N, float* dW, float* dbias,
                      void* workspace, dmi_reduce_item* deferred, int* n_deferred, void* stream);

/* out[(b,oy,ox)][t*C + c] = x[b, oy*stride + dy[t], ox*stride + dx[t], c] (0 outside); row pitch ldo, tail zero-filled.
 * dy/dx are HOST arrays (ntaps <= 16). */
int dmi_im2col(const uint16_t* x, uint16_t* out, int B, int H, int W, int C, int Ho, int Wo, int stride,
               int ntaps, const int* dy, const int* dx, int ldo, void* stream);
/* out[a][t*Bn + b] = in[idx[t]][a][b], row pitch ldo (tail zero)  (conv kernel [k][ci][co] -> [ci][(k',co)] for the
 * dgrad / output-parity GEMMs); idx on HOST */
int dmi_weight_gather(const uint16_t* in, uint16_t* out, int A, int Bn, int nsel, const int* idx, int ldo, void* stream);
/* n independent weight gathers between two buffers in ONE launch (the per-step refresh of every dgrad / output-parity weight
 * copy of a model): table[i] = {in_off, out_off, A, Bn, nsel, ldo, first_block, idx[16]} (23 int64 per row, DEVICE memory;
 * offsets in elements, first_block ascending, an item spans ceil(A*ldo / 2048) blocks); total_blocks = their sum. */
int dmi_weight_gather_batch(const uint16_t* in_base, uint16_t* out_base, const int64_t* table, int n, int64_t total_blocks,
                            void* stream);
/* out[b, 2t+py, 2u+px, :] = in4[py*2+px][b, t, u, :]  (assembles a stride-2 transposed convolution) */
int dmi_pixel_interleave(const uint16_t* in4, uint16_t* out, int B, int Ht, int Wt, int C, void* stream);
/* fp32 [N, Cin] <-> bf16 [N, Cp] (zero-padded channels): image in, reconstruction out */
int dmi_pad_channels(const float* in, uint16_t* out, int64_t N, int Cin, int Cp, void* stream);
int dmi_unpad_channels(const uint16_t* in, float* out, int64_t N, int Cin, int Cp, void* stream);
/* fp32 convolution of the TOKENISING encoder (the reference builds the VAE without use_bf16 inside dalle_model_fn,
 * src/model_fns.py:43-51, so image tokens come from fp32 convolutions and an fp32 x @ codebook, src/vae_tf/models.py:81-120):
 *   out[(b,oy,ox)][n] = sum_t sum_c x[b, oy*stride + dy[t], ox*stride + dx[t], c] * Wk[(t*C + c)][n] + bias[n] (ReLU) (+ residual)
 * x NHWC fp32 [B,H,W,C] (C % 8 == 0, zero outside the image = SAME padding), Wk fp32 [(ntaps*C), N] = the TF kernel layout
 * [kh,kw,Cin,Cout], bias / residual nullable, out fp32 [B*Ho*Wo, N], N % 4 == 0.  Exact-fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32).  One tap on a 1x1 image = a plain fp32 matmul (the codebook product). */
int dmi_conv2d_f32(const float* x, int B, int H, int W, int C, int Ho, int Wo, int stride, int ntaps, const int* dy,
                   const int* dx, const float* Wk, const float* bias, const float* residual, float* out, int N, int relu,
                   void* stream);
/* tf.space_to_depth / tf.depth_to_space (NHWC, block s; src/vae_tf/models.py:85-86,158-161) between the image
 * [B, Hs*s, Ws*s, C] and the stacked layout [B, Hs, Ws, Cp] with channel (dy*s + dx)*C + c, Cp >= s*s*C (pad = 0). */
int dmi_space_to_depth_f32(const float* img, float* stacked, int B, int Hs, int Ws, int C, int s, int Cp, void* stream);
int dmi_depth_to_space_f32(const float* stacked, float* img, int B, int Hs, int Ws, int C, int s, int Cp, void* stream);
/* K13 gumbel_softmax (layers.py:4-21) with INJECTED uniforms u in [1e-9, 1): y = softmax((logits - log(-log u))/T);
 * hard: y = one_hot(argmax) (first max), gradient straight-through.  y, y_soft bf16 [M,T]; index int32 [M] (nullable).
 * temperature_dev (nullable): read T from temperature_dev[0] (DEVICE memory) instead of the argument, so the annealed
 * schedule (src/model_fns_tf.py:40-45) can advance between replays of a captured HIP graph. */
int dmi_gumbel_softmax_fwd(const float* logits, const float* u, uint16_t* y, uint16_t* y_soft, int32_t* index,
                           int64_t M, int T, float temperature, int hard, const float* temperature_dev, void* stream);
int dmi_gumbel_softmax_bwd(const uint16_t* dy, const uint16_t* y_soft, uint16_t* dlogits, int64_t M, int T,
                           float temperature, const float* temperature_dev, void* stream);
/* K14 mse_loss (layers.py:24-25): loss[0] = mean((img - out)^2) over N*Cin; dout (nullable) = 2(out-img)*grad_scale/(N*Cin) */
int64_t dmi_mse_workspace_bytes(void);
int dmi_mse_loss(const float* img, const uint16_t* outp, uint16_t* dout, float* loss, int64_t N, int Cin, int Cp,
                 float grad_scale, void* workspace, void* stream);
int dmi_add_f32(float* dst, const float* src, int64_t n, void* stream);


/* ---- input format (host only): CRC-32C (Castagnoli) of a TFRecord length header / payload, as tf.data.TFRecordDataset
 * verifies them (reference src/input_fns.py:111); the caller applies TFRecord's mask.  No device work, thread-safe. */
uint32_t dmi_crc32c(const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif
