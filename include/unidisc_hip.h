/* unidisc_hip.h — C ABI of the MI355X (gfx950) kernel library behind UniDisc's denoising hot path.
 *
 * The reference (alexanderswerdlow/unidisc) is pure Python/PyTorch and has NO native interface of its
 * own; each entry point below therefore replaces a torch / third-party-kernel call
 * site of the reference, cited as file:line relative to the reference root.  The binding a maintainer adds
 * on the reference side is a ctypes stub; `unidisc_amd/_lib.py` is that stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-allocated DEVICE memory (PyTorch tensors in
 *     practice), the library never allocates, frees or synchronises;
 *   - bf16 tensors are `void*` to raw 16-bit words, row-major, 16-byte aligned, row strides in ELEMENTS;
 *   - all kernels are enqueued on `stream` and return immediately;
 *   - return value 0 = enqueued; non-zero = nothing enqueued (2: bad argument, 1: launch failure) and
 *     udm_last_error() holds a message; no exception or exit crosses the ABI;
 *   - thread-safety: entry points may be called from any host thread (forward on the main thread,
 *     backward on the autograd thread); the error string is thread-local.
 */
#ifndef UNIDISC_HIP_H
#define UNIDISC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define UDM_ABI_VERSION 3

/* GEMM epilogues */
#define UDM_EPI_NONE 0      /* C = A·Bᵀ                                                   */
#define UDM_EPI_BIAS 1      /* C = A·Bᵀ + bias[n]                                          */
#define UDM_EPI_BIAS_GELU 2 /* u = bf16(A·Bᵀ + bias); C = gelu_tanh(u); aux = bf16(gelu_tanh'(u))   (mlp.0 + GELU: the derivative is saved, not u) */
#define UDM_EPI_DGELU 3     /* C = (A·Bᵀ) ⊙ aux                              (GELU backward); a non-NULL `bias` is then an fp32 [N]
                               OUTPUT accumulating the column sums of C (the bias gradient of the upstream Linear)         */

const char* udm_last_error(void);
/* Diagnostics / A-B switches, none of them needed by a caller (process-global; values as documented in csrc/capi.hip): keys "gemm_tile", "gemm_quad",
 * "gemm_persist", "gemm_quad_asm" (0 = the C++ K loop of the one-wave-per-SIMD GEMM everywhere), "attention_tr_read", "attention_fwd64", "attention_dq64", "attention_dkv64" (0 = the 8-wave backward kernels everywhere, 2 = the generated programs without the balanced walk), "exp" (experiment bits, 0 in
 * production); diagnostic builds only: "attention_fwd64_timeline", "attention_dq64_timeline", "attention_dkv64_timeline", "gemm_quad_timeline" (device pointers for cycle stamps).  Returns 2 for an unknown key. */
int udm_debug_set(const char* key, int64_t value);
/* Diagnostics: hold `blocks` CUs (1..128; one 160 KiB-LDS block each) until *flag != 0 (pinned host or device memory) - the stand-in for a collective's channel
 * kernels when the GEMMs' behaviour under a CU reservation is measured on one GPU. */
int udm_debug_cu_hog(int64_t blocks, const int* flag, hipStream_t stream);
int udm_abi_version(void);

/* ---- GEMM family: nn.Linear forward / dgrad / wgrad under bf16 autocast -------------------------
 * replaces: models/dit.py:642 (attn_qkv), :877-887 (attn_out), :917-919 + :1016/1025 (mlp), :1091 (head),
 *           :966-967, :1084 (adaLN_modulation), :447 (sigma_map) and their autograd backward.
 * C[M,N] = A[M,K] · B[N,K]ᵀ, A/B bf16 K-contiguous; C bf16 or fp32; beta accumulates into an fp32 C.   */
int udm_gemm_nt_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int out_f32,
                     int epilogue, const float* bias, void* aux, int64_t ldaux, float beta, hipStream_t stream);
/* wgrad form read straight from row-major activations: C[M,N] (fp32) = beta*C + A[K,M]ᵀ · B[K,N]  (K % 64 == 0). */
/* dgrad without a transposed weight shadow: C[M,N] bf16 = A[M,K] B[K,N] (A = dY with K = out features contiguous, B = the forward's bf16 W [out, in]);
 * replaces the dX = dY W half of nn.Linear's backward (models/dit.py:642,877,917-919).  Shapes must pass udm_gemm_nn_ok(M, N, K) != 0:
 * N % 256 == 0, K % 64 == 0, K >= 128 and either M a multiple of 192 / 256 / 320 or enough 320-row tiles (>= 128, the last tile row may be ragged) to fill the chip. */
int udm_gemm_nn_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, hipStream_t stream);
int udm_gemm_nn_ok(int64_t M, int64_t N, int64_t K);
int udm_gemm_tn_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float beta,
                     hipStream_t stream);
/* the same wgrad for FEW output tiles over a LONG contraction (2048 x 2048 out-proj weight, K = B*L): K is split so that tiles x slices fill
 * the CUs, partial tiles go to `ws` (fp32, >= slices*M*N elements; no atomics), a reduce pass sums them into C.  Falls back to
 * udm_gemm_tn_bf16 when the workspace is too small.  C contiguous (ldc == N). */
int udm_gemm_tn_splitk_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float beta,
                            float* ws, int64_t ws_elems, hipStream_t stream);
/* two wgrads of one backward step in ONE launch (same N and K; M0, M1, N multiples of 256, K of 64): their 256 x 256 tiles share a grid - the qkv (192 tiles) and
 * out-proj (64 tiles) weight gradients of a DiT block fill the 256 CUs exactly once.  Returns 3 without doing anything when the shapes do not qualify. */
int udm_gemm_tn_pair_bf16(const void* A0, const void* B0, void* C0, int64_t M0, int64_t lda0, int64_t ldb0, int64_t ldc0, const void* A1, const void* B1, void* C1,
                          int64_t M1, int64_t lda1, int64_t ldb1, int64_t ldc1, int64_t N, int64_t K, float beta, float* ws, int64_t ws_elems, hipStream_t stream);
/* up to four wgrads of one backward step over the SAME contraction in ONE split-K launch + ONE reduce pass (the few-tile weight gradients of a small DiT block:
 * UniDisc-S's qkv / out-proj / mlp.0 / mlp.2 weights over K = B*L): C_i[M_i, N_i] (fp32, contiguous) = beta * C_i + A_i[K, M_i]^T B_i[K, N_i]; M_i, N_i multiples of
 * 256, K of 64, at most 128 tiles in total, ws >= slices * sum(M_i * N_i) floats with slices = min(CUs / tiles, K / 512, 32) >= 2.  Returns 3 without doing
 * anything when the shapes do not qualify. */
int udm_gemm_tn_multi_bf16(int nprob, const void* const* A, const void* const* B, void* const* C, const int64_t* M, const int64_t* N, const int64_t* lda,
                           const int64_t* ldb, int64_t K, float beta, float* ws, int64_t ws_elems, hipStream_t stream);
/* (few tiles over a long K: split in K through `ws` - fp32, >= slices * (M0 + M1) * N elements, slices = min(256 / tiles, K / 512, 32); NULL = never split) */
/* NT form of the same idea with a bf16 result (head dgrad on the compacted [MASK] rows: few 320 x 256 tiles over K = V): fp32 partial tiles in ws
 * (>= slices*M*N), reduce pass rounds to bf16; falls back to udm_gemm_nt_bf16 when splitting does not apply. */
int udm_gemm_nt_splitk_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float* ws,
                            int64_t ws_elems, hipStream_t stream);
/* NN form of the same idea (bf16 result, whole tiles: udm_gemm_nn_ok): the leftover tile rows of a dgrad when a collective holds CUs (below). */
int udm_gemm_nn_splitk_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float* ws,
                            int64_t ws_elems, hipStream_t stream);
/* data-parallel runs: the GEMMs plan for `cus` CUs (multiple of 8 in [8, 256]; 0 = all 256) so that RCCL's channel kernels of the gradient all-reduce
 * overlapped with backward (main.py:641-656) find free CUs: the persistent NT grid is capped at `cus` blocks, the split-K forms cut K for `cus` blocks, and
 * the host side (kernels.py) runs a single-round GEMM of more than `cus` tiles as whole tile rows that fit + a split-K launch of the leftover rows; also env
 * UDM_GEMM_CUS */
int udm_gemm_set_cus(int cus);
/* nn.Linear backward for a SMALL batch in one launch - adaLN_modulation (models/dit.py:922-925, the final layer's :1091-1097): dY fp32 [B, out] (rounded to bf16 on load:
 * the autocast backward's operand), X bf16 [B, in], W bf16 [out, in] (the forward's shadow): dW[out, in] = dYᵀ X (overwritten), db[out] += colsum(dY) (nullable), and the
 * input gradient dY W either added into dX[B, in] (fp32 atomics) or - dX_parts non-null - written as udm_small_batch_linear_bwd_blocks(out) partial tiles
 * dX_parts[tile][B][in] for the caller to sum (the conditioning vector's gradient is needed once, after the last block).  B <= 64, in <= 128 and a multiple of 8,
 * rows of X and W 16-byte aligned. */
int udm_small_batch_linear_bwd(const float* dY, int64_t lddy, const void* X, int64_t ldx, const void* W, int64_t ldw, float* dW, float* db, float* dX, int64_t lddx,
                               float* dX_parts, int64_t B, int64_t out, int64_t in, hipStream_t stream);
int udm_small_batch_linear_bwd_blocks(int64_t out);
/* out[C,R] = in[R,C]ᵀ (bf16); optional colsum[c] += Σ_r in[r,c] (bias gradient).  Feeds the wgrad GEMMs. */
int udm_transpose_bf16(const void* in, void* out, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, float* colsum, hipStream_t stream);
/* fp32 master weights -> bf16 shadow (and Kᵀ-major shadow for dgrad): the per-forward autocast weight cast. */
int udm_cast_transpose_f32_bf16(const float* in, void* out, void* out_t, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, int64_t ld_t,
                                hipStream_t stream);
/* the same for MANY matrices in one launch.  jobs: device array of njobs 64-byte records, sorted by tile0:
 *   { const float* in; bf16* out (or null); bf16* out_t (or null); int64 ld_in, ld_out, ld_t; int32 R, C; int32 tile0, tiles_c }
 * with tiles_c = ceil(C / 64), tile0 = number of 64 x 64 tiles of all earlier records; total_tiles = sum of ceil(R/64) * tiles_c. */
int udm_cast_transpose_multi_f32_bf16(const void* jobs, int64_t njobs, int64_t total_tiles, hipStream_t stream);
int udm_cast_f32_bf16(const float* x, void* y, int64_t n, float scale, hipStream_t stream); /* DDP bf16 compress hook, main.py:645 */
int udm_cast_bf16_f32(const void* x, float* y, int64_t n, float scale, hipStream_t stream); /* ... and decompress */

/* ---- norms: RMSNorm models/dit.py:77-100, LayerNorm(no bias) :383-403, modulate_fused :263-304 -----
 * y(bf16) = modulate(norm(x) * w).  norm_type 0 = RMS (eps 1e-6), 1 = LayerNorm (eps 1e-5).
 * shift/scale: bf16 [B, mod_stride] slices of the adaLN output or NULL; `modality` [M] + `any_img` device
 * flag select the reference's image-token-only modulation.  L = rows per batch element.              */
int udm_norm_fwd(const float* x, void* y, float* rstd, float* mean, const float* w, const void* shift, const void* scale, int64_t mod_stride,
                 const int64_t* modality, const int* any_img, int64_t M, int64_t d, int64_t L, int norm_type, float eps, hipStream_t stream);
int udm_norm_bwd(const void* dy, const float* x, const float* rstd, const float* mean, const float* w, const void* shift, const void* scale,
                 int64_t mod_stride, const int64_t* modality, const int* any_img, float* dx, float* dw, float* dshift, float* dscale, int64_t M,
                 int64_t d, int64_t L, int norm_type, int accumulate, float* ws, int64_t ws_elems, hipStream_t stream);
/* ws: optional fp32 scratch (>= 512*d): per-block dw partials + a reduce pass instead of 512-deep same-address atomic chains */

/* ---- residual branch: bias_dropout_add_scale models/dit.py:229-253 and the sandwich adds :993-994, :1015-1031
 * x_out = x_in + gate ⊙ dropout_p(sandwich_norm(branch; w_b)); every stage optional (NULL / p = 0).       */
int udm_residual_fwd(const float* x_in, const void* branch, float* x_out, const float* w_b, float* rstd_b, float* mean_b, const void* gate,
                     int64_t mod_stride, const int64_t* modality, int64_t M, int64_t d, int64_t L, int norm_type, float eps, float p_drop, uint64_t seed,
                     hipStream_t stream);
/* the same add with the NEXT (unmodulated) pre-norm fused: h_out = norm(x_out; w_next) as bf16 (norm2 of the block `dit.py:1008`, norm1 of the next
 * block `:985`, norm_final `:1091` when there is no adaLN) -- x_out is normalised while its row is still in registers */
int udm_residual_norm_fwd(const float* x_in, const void* branch, float* x_out, const float* w_b, float* rstd_b, float* mean_b, const void* gate,
                          int64_t mod_stride, const int64_t* modality, int64_t M, int64_t d, int64_t L, int norm_type, float eps, float p_drop,
                          uint64_t seed, const float* w_next, void* h_out, float* rstd_next, float* mean_next, hipStream_t stream);
/* ... with the next pre-norm MODULATED (adaLN-Zero, models/dit.py:263-304): h_out = norm(x_out; w_next) (1 + next_scale) + next_shift on the rows udm_norm_fwd would
 * modulate (next_shift / next_scale bf16 [B, next_mod_stride]; next_modality / next_any_img as udm_norm_fwd's modality / any_img). */
int udm_residual_norm_fwd_ada(const float* x_in, const void* branch, float* x_out, const float* w_b, float* rstd_b, float* mean_b, const void* gate, int64_t mod_stride,
                              const int64_t* modality, int64_t M, int64_t d, int64_t L, int norm_type, float eps, float p_drop, uint64_t seed, const float* w_next,
                              void* h_out, float* rstd_next, float* mean_next, const void* next_shift, const void* next_scale, int64_t next_mod_stride,
                              const int64_t* next_modality, const int* next_any_img, hipStream_t stream);
int udm_residual_bwd(const float* dx, const void* branch, void* dbranch, const float* w_b, const float* rstd_b, const float* mean_b, const void* gate,
                     int64_t mod_stride, const int64_t* modality, float* dw_b, float* dgate, int64_t M, int64_t d, int64_t L, int norm_type, float p_drop,
                     uint64_t seed, float* ws, int64_t ws_elems, hipStream_t stream);
/* udm_norm_bwd (unmodulated) immediately followed by udm_residual_bwd (no gate) on the dx it has just updated, as ONE pass per row
 * (d = 2048 / 4096): the pairs norm2 -> attention branch, norm1 -> previous block's MLP branch, final norm -> last MLP branch of the block
 * backward (models/dit.py:77-100 RMSNorm / :383-403 LayerNorm backward feeding :229-253, :993-994).  dbias (nullable): += column sums of the
 * bf16 d branch, i.e. the bias gradient of the Linear that produced the branch (mlp.2, :919).  ws: >= min(M, 1536) * 3 * d floats. */
int udm_norm_residual_bwd(const void* dy, const float* x, const float* rstd, const float* mean, const float* w, float* dx, float* dw, int accumulate,
                          const void* branch, void* dbranch, const float* w_b, const float* rstd_b, const float* mean_b, float* dw_b, float* dbias,
                          int64_t M, int64_t d, int norm_type, float p_drop, uint64_t seed, float* ws, int64_t ws_elems, hipStream_t stream);
/* The same fused pass for adaLN-Zero (time_conditioning = True; models/dit.py:263-304 modulate_fused, :229-253 bias_dropout_add_scale): the norm modulated by
 * shift / scale [B, mod_stride] bf16 (rows with modality == 1 only when `modality` is given and *any_img != 0), the residual branch optionally gated by gate
 * [B, mod_stride] (gate and dropout on rows with modality_r == 1 only when that map is given); dshift / dscale / dgate [B, mod_stride] fp32 += column sums over each
 * batch element's rows (a block owns rows of one element: one atomic per column and block).  d = 2048 / 4096, M = B * L.  Any of shift / gate may be null. */
int udm_norm_residual_bwd_ada(const void* dy, const float* x, const float* rstd, const float* mean, const float* w, float* dx, float* dw, int accumulate,
                              const void* branch, void* dbranch, const float* w_b, const float* rstd_b, const float* mean_b, float* dw_b, float* dbias,
                              const void* shift, const void* scale, float* dshift, float* dscale, const void* gate, float* dgate, int64_t mod_stride,
                              const int64_t* modality, const int* any_img, const int64_t* modality_r, int64_t M, int64_t d, int64_t L, int norm_type, float p_drop,
                              uint64_t seed, float* ws, int64_t ws_elems, hipStream_t stream); /* ws: optional fp32 scratch (>= 1024*d) for a two-phase dw_b reduction */

/* ---- QK LayerNorm (models/dit.py:569-572, 680-682) + rotary (models/standalone_rotary.py:14-31, call dit.py:723-726)
 * qkv bf16 [M,3d] -> qkr bf16 [M,2d] (normalised, rotated q | k).  cos/sin fp32 [L,D/2] or per-sample [M,D/2]. */
int udm_qknorm_rope_fwd(const void* qkv, void* qkr, const float* gq, const float* bq, const float* gk, const float* bk, float* stats, const float* cos_t,
                        const float* sin_t, int rope_per_sample, int64_t M, int64_t d, int64_t L, int64_t D, float eps, float q_scale, hipStream_t stream);
/* q_scale (1 = the reference's values): the rotated q is stored as bf16(q * q_scale), ONE rounding.  With q_scale = log2(e) / sqrt(D) the attention entry
 * points take UDM_ATTN_Q_PRESCALED and skip the per-score multiply (flash-attn scales the fp32 scores of a bf16 q: the same number of roundings, at a
 * different place); udm_qknorm_rope_bwd with the same q_scale takes the gradient wrt the stored (scaled) q. */
int udm_qknorm_rope_bwd(const void* dqkr, const void* qkv, void* dqkv, const float* gq, const float* gk, const float* stats, const float* cos_t,
                        const float* sin_t, int rope_per_sample, float* dgq, float* dbq, float* dgk, float* dbk, int64_t M, int64_t d, int64_t L, int64_t D,
                        float q_scale, float* ws, int64_t ws_elems, hipStream_t stream); /* ws: optional fp32 scratch (>= 4096*d); then dgq|dbq|dgk|dbk must be contiguous */

/* ---- attention core: flash_attn_qkvpacked_func models/dit.py:843 / SDPA :826-829 / FlexAttention doc mask :784-812
 * bidirectional softmax(QKᵀ/√D)V; element (b,l,h,:) of a tensor lives at base + (b*L+l)*stride + h*D.
 * sample_ids [B,L] (or NULL): attend iff ids equal and != -1 (model_utils.py:740-771).  lse/delta fp32 [B,H,L].
 * doc_ranges int32 [B, ceil(L/64), 8] (or NULL), from udm_attention_doc_ranges on the same sample_ids: per 64-row tile {lo, hi, idmin, idmax, exact, 0, 0, 0}
 * - the [lo, hi) span of positions that can share a sample id with the tile, so the kernels walk only those tiles; the tile's id interval (idmin = -1
 * when it holds padding), so that tile pairs inside one document skip the per-element id test (FlexAttention's BlockMask does both: empty / full /
 * partial blocks, model_utils.py:716-771); exact = the tile's one document occupies exactly [lo, hi) (such key blocks take the wave-specialised dK/dV
 * kernel).  The forward and dQ are bit-identical with and without it; dK/dV agree to summation order at head dim 128. */
int udm_attention_doc_ranges(const int64_t* sample_ids, int64_t B, int64_t L, int32_t* ranges, hipStream_t stream);
int udm_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int64_t* sample_ids, const int32_t* doc_ranges, int64_t B, int64_t H, int64_t L, int64_t D,
                      int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride, int64_t flags, hipStream_t stream);
int udm_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta, void* dq, void* dk,
                      void* dv, const int64_t* sample_ids, const int32_t* doc_ranges, int64_t B, int64_t H, int64_t L, int64_t D, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                      int64_t o_stride, int64_t do_stride, int64_t dq_stride, int64_t dk_stride, int64_t dv_stride, int64_t flags, hipStream_t stream);
/* delta: caller's fp32 scratch of 3 B H L floats (rowsum(dO O) of the dQ pass, then - with UDM_ATTN_Q_PRESCALED - the negated lse and delta the dK/dV pass
 * starts its score accumulators from). */
/* flags: UDM_ATTN_Q_PRESCALED = q holds bf16(q log2(e) / sqrt(D)) (udm_qknorm_rope_fwd with that q_scale): the kernels skip the per-score multiply, dq is
 * the gradient wrt that stored q (udm_qknorm_rope_bwd with the same q_scale), lse is unchanged.  The forward at head dim 128 without sample_ids,
 * L % 256 == 0, L >= 512, H >= 2, (B H) % 8 == 0 then runs the persistent 64-queries-per-wave kernel (csrc/attention_fwd64.hip), and the backward its two generated
 * counterparts (csrc/attention_dq64.hip: dQ + delta + the planes; csrc/attention_dkv64.hip: dK / dV; 16-byte aligned row strides). */
#define UDM_ATTN_Q_PRESCALED 1

/* ---- embeddings: EmbeddingLayer models/dit.py:1036-1043 (+modality embedding :1402-1411) ------------- */
int udm_embedding_fwd(const int64_t* ids, const float* E, const int64_t* modality, const float* Em, float* x, int64_t M, int64_t d, int64_t V,
                      hipStream_t stream);
int udm_embedding_bwd(const int64_t* ids, const int64_t* modality, const float* dx, float* dE, float* dEm, int64_t M, int64_t d, int64_t V, int64_t hot_id,
                      hipStream_t stream);

/* ---- SUBS cross-entropy: Diffusion._subs_parameterization model.py:621-658 + gather :967 -------------- */
int udm_subs_ce_fwd(const void* logits, int64_t ld, const int64_t* x0, const int64_t* xt, const int64_t* modality, float* log_p, float* lse, int64_t M,
                    int64_t V, int64_t Vt, int64_t mask_id, int restrict_modality, hipStream_t stream);
/* udm_subs_ce_bwd, narrow_txt_rows >= 0 (needs restrict_modality; the head runs per modality on a compacted row list whose first narrow_txt_rows rows are the text
 * group): only the columns the row's group of head GEMMs reads are written - text group [0, ceil64(Vt)), image group [floor8(Vt), ld) - the rest of the buffer is left
 * as it is.  -1 = write whole rows. */
int udm_subs_ce_bwd(void* logits, int64_t ld, const int64_t* x0, const int64_t* xt, const int64_t* modality, const float* lse, const float* g, int64_t M,
                    int64_t V, int64_t Vt, int64_t mask_id, int restrict_modality, int64_t narrow_txt_rows, hipStream_t stream);
int udm_subs_logprobs(const void* logits, int64_t ld, const int64_t* xt, const int64_t* modality, void* out, int64_t ld_out, int out_f32, int64_t M, int64_t V,
                      int64_t Vt, int64_t mask_id, int restrict_modality, hipStream_t stream);

/* ---- loss arithmetic behind the per-token log-probabilities: Diffusion.compute_loss model.py:1010-1160 (schedule weights applied per row, masked mean or the
 * modality-weighted text / image sum with the optional text-loss cap); nlls = -log_p w_std on attended tokens, coef = d loss / d log_p,
 * scalars = {loss, txt_loss, img_loss, txt_frac, img_frac, valid_frac, txt_count, img_count}.  attention_mask: bool [B, L]; modality_mask: bool [B, L, 2] or NULL */
int udm_diffusion_loss(const float* log_p, const float* w_loss, const float* w_std, const void* attention_mask, const void* modality_mask, float* nlls, float* coef,
                       float* scalars, int64_t B, int64_t L, int weighted, int full_mask, float text_w, float img_w, float ratio, hipStream_t stream);

/* ---- adaLN-Zero helpers: TimestepEmbedder models/dit.py:415-449, F.silu :1379 ------------------------- */
int udm_timestep_embedding(const float* sigma, void* out, int64_t B, int64_t dim, hipStream_t stream);
int udm_silu_fwd(const void* x, void* y, int64_t n, hipStream_t stream);
int udm_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, hipStream_t stream);

/* ---- sampler step: one reverse-diffusion update of [MASK] rows straight from bf16 logits, no [rows, V] probabilities.
 * p = exp(SUBS log-probs) (`_ddpm_forward` model_eval.py:1761-1834 no-CFG branch); q_i = p_i (t - s), q_mask = s (`_ddpm_caching_update`
 * :2073-2106); out = argmax_i q_i / (1e-10 - log(u_i + 1e-10)) (`_sample_categorical` model_utils.py:95-97; first index on ties).
 * logits: [M, ld] rows of masked positions only; t / s: per-row move chance now / next; u: explicit uniforms [M, ldu] (parity runs replay the
 * reference's rand stream) or NULL -> Philox(seed).  greedy != 0: argmax of the log-probs (`noise_removal` :2437-2444), t / s / u ignored. */
int udm_ddpm_sample_rows(const void* logits, int64_t ld, const int64_t* modality, const float* t, const float* s, const float* u, int64_t ldu,
                         uint64_t seed, int64_t* out, int64_t M, int64_t V, int64_t Vt, int64_t mask_id, int restrict_modality, int greedy,
                         hipStream_t stream);

/* Same update with classifier-free guidance fused in (`_ddpm_forward` CFG branch model_eval.py:1763-1817): z = (1 + w[row]) logits - w[row] logits_uncond in
 * fp32, then SUBS and the draw as above.  logits_uncond / w NULL: identical to udm_ddpm_sample_rows. */
int udm_ddpm_sample_rows_cfg(const void* logits, const void* logits_uncond, const float* w, int64_t ld, const int64_t* modality, const float* t, const float* s,
                             const float* u, int64_t ldu, uint64_t seed, int64_t* out, int64_t M, int64_t V, int64_t Vt, int64_t mask_id,
                             int restrict_modality, int greedy, hipStream_t stream);

/* `maskgit` predictor, per [MASK] row (`_maskgit_update` model_eval.py:3069-3074): x ~ Categorical(exp(SUBS log-probs)) - the exponential race
 * argmax p_i / (1e-10 - log(u_i + 1e-10)), explicit uniforms or Philox(seed) - or x = given[row] (replay of a recorded draw), and out_logp[row] =
 * log p(x).  Guidance (logits_uncond, w) as in udm_ddpm_sample_rows_cfg. */
int udm_categorical_sample_rows(const void* logits, const void* logits_uncond, const float* w, int64_t ld, const int64_t* modality, const float* u, int64_t ldu,
                                uint64_t seed, const int64_t* given, int64_t* out, float* out_logp, int64_t M, int64_t V, int64_t Vt, int64_t mask_id,
                                int restrict_modality, hipStream_t stream);

/* ---- token data path: joint-sequence assembly, `Diffusion.update_batch` token-dataset branch model.py:183-212 over the dataset
 * schema of models/datasets/image_datasets.py:263-281.  txt [n, Lt] int32, txt_mask [n, Lt] bool bytes (nullable: all valid), img [n, Li] int16;
 * idx [B] rows to gather (nullable: rows 0..B-1).  Writes input_ids int64 [B, Lt+Li] (image ids shifted by Vt), attention_mask bool bytes, modality
 * int64 (0 text / 1 image).  The caller guarantees 0 <= idx[b] < n. */
int udm_assemble_joint_tokens(const int32_t* txt, const uint8_t* txt_mask, const int16_t* img, const int64_t* idx, int64_t B, int64_t Lt, int64_t Li,
                              int64_t Vt, int64_t* ids, uint8_t* mask, int64_t* modality, hipStream_t stream);

/* ---- interleaved / packed rows: the per-position layout work of a packed batch as kernels (static shapes, no host reads)
 * udm_interleaved_rope: models/dit.py:1421-1444 with add_img_data_to_blocks / add_txt_data_to_blocks (:122-191).  modality / sid int64 [B, L]; img_cos / img_sin the 2-D
 *   tables of the supported image block sizes concatenated in the order of `sizes` (a HOST array of nsizes <= 8 ints), [sum sizes, half] fp32; txt_cos / txt_sin
 *   [txt_rows, half].  Writes cos, sin fp32 [B, L, half] and count_idx int64 [B, L] (row of img_count_embedding to add, -1: none).  scratch: B * 5 * L ints.
 * udm_interleaved_block_lottery: model.py:483-522 after its draws - candidate block i of the batch (row-major order) reads r[i] (r fp32 [n_r]; ranks >= n_r never hit).
 *   Writes accum bool bytes [B, L] (positions of the blocks masked as a whole), rows_hit bool bytes [B], n_cand int64 [1] (number of candidate blocks).
 *   scratch: B * 4 * L ints, row_cands: B ints.
 * udm_rowgroup_sum_f32: out[g, :] += sum of the rows r of x [M, d] with group[r] == g (0 <= g < G <= 32; other rows are skipped): the image-count embedding's gradient. */
int udm_interleaved_rope(const int64_t* modality, const int64_t* sid, const float* img_cos, const float* img_sin, const int32_t* sizes, int64_t nsizes, const float* txt_cos,
                         const float* txt_sin, int64_t txt_rows, int64_t B, int64_t L, int64_t half, float* cos, float* sin, int64_t* count_idx, int32_t* scratch,
                         hipStream_t stream);
int udm_interleaved_block_lottery(const int64_t* modality, const int64_t* sid, const float* r, int64_t n_r, float mask_prob, int64_t B, int64_t L, uint8_t* accum,
                                  uint8_t* rows_hit, int64_t* n_cand, int32_t* scratch, int32_t* row_cands, hipStream_t stream);
int udm_rowgroup_sum_f32(const float* x, const int64_t* group, float* out, int64_t M, int64_t d, int64_t G, hipStream_t stream);
/* q_xt model.py:424-587 for multimodal (non-interleaved) batches after its random draws: move = r_move < move_chance[b], whole-modality masking (a row drawn for both
 * modalities masks neither), xt = move ? mask_id : x; outputs bool [B, L] move_indices and bool [B] rows (text masked, image masked, either).  r_txt / r_img NULL: no draw. */
int udm_qxt_absorbing(const int64_t* x, const float* r_move, const float* move_chance, const float* r_txt, const float* r_img, float thr_txt, float thr_img,
                      const void* modality_mask, int64_t B, int64_t L, int64_t mask_id, int64_t* xt, void* move_indices, void* row_txt, void* row_img, void* row_ignore,
                      hipStream_t stream);
/* _sample_t model.py:589-619 + LogLinearNoise models/noise_schedule.py:128-157 from the uniform draws u [n]: t, sigma, dsigma, move chance (fp32 [n] each), every
 * statement rounded like the reference's tensor statements.  sampling_eps_complement = fp32(1 - sampling_eps), noise_eps_complement = fp32(1 - eps). */
int udm_sample_t_noise(const float* u, int64_t n, int antithetic, float sampling_eps_complement, float sampling_eps, float noise_eps_complement, float* t, float* sigma,
                       float* dsigma, float* move_chance, hipStream_t stream);

/* ---- optimizer step: torch.optim.AdamW(fused=True) `model_setup.py:385-424` + accelerator.clip_grad_norm_ `model.py:1516-1520`.
 * fp32 masters and moments; `step` is the 1-based update count (bias corrections are computed from it); `grad_norm_sq` (nullable) is a DEVICE
 * scalar holding the sum of squares of ALL gradients (udm_sumsq_f32 over the engine's flat gradient buffer): g is scaled by
 * min(1, max_grad_norm / (sqrt(*grad_norm_sq) + 1e-6)) without a host round trip.  The _shadow form updates a row-major [R, C] GEMM weight and
 * writes its bf16 copy [R, ld16] and transposed bf16 copy [C, ldt] in the same pass (replaces the per-forward autocast weight cast). */
int udm_sumsq_f32(const float* x, int64_t n, float* out, float* ws, int64_t ws_elems, hipStream_t stream); /* ws: >= 1024 floats */
int udm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   int64_t step, const float* grad_norm_sq, float max_grad_norm, hipStream_t stream);
int udm_adamw_step_shadow(float* p, const float* g, float* m, float* v, int64_t R, int64_t C, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int64_t step, const float* grad_norm_sq, float max_grad_norm, void* w16, int64_t ld16, void* w16t,
                          int64_t ldt, hipStream_t stream);
/* The same updates with the parameter EMA of models/ema.py:44-53 (`ExponentialMovingAverage.update`, called after optimizer.step at model.py:1541-1545)
 * folded into the pass: ema <- ema - (1 - ema_decay) (ema - p_new).  `ema_decay` is the decay OF THIS UPDATE (the caller applies the reference's
 * warm-up min(decay, (1 + n) / (10 + n))).  ema NULL: identical to the plain forms. */
int udm_adamw_step_ema(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int64_t step, const float* grad_norm_sq, float max_grad_norm, float* ema, float ema_decay, hipStream_t stream);
/* the same update for MANY flat tensors in one launch: `jobs` = device array of njobs records {float* p; const float* g; float* m; float* v; float* ema (nullable);
 * int64 n; int64 chunk0} with chunk0 = sum over the earlier jobs of ceil(n / 1024) and nchunks the total; every pointer 16-byte aligned. */
int udm_adamw_step_multi(const void* jobs, int64_t njobs, int64_t nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                         const float* grad_norm_sq, float max_grad_norm, float ema_decay, hipStream_t stream);
/* ... and for MANY 2-D GEMM weights with their bf16 shadows: `jobs` = device array of records {float* p; const float* g; float* m; float* v; float* ema (nullable);
 * bf16* w16 (nullable); bf16* w16t (nullable); int64 ld16; int64 ldt; int32 R; int32 C; int32 tile0; int32 tiles_c} with tiles_c = ceil(C / 64) and tile0 = sum over
 * the earlier jobs of ceil(R / 64) * tiles_c; ntiles the total. */
int udm_adamw_step_shadow_multi(const void* jobs, int64_t njobs, int64_t ntiles, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                const float* grad_norm_sq, float max_grad_norm, float ema_decay, hipStream_t stream);
int udm_adamw_step_shadow_ema(float* p, const float* g, float* m, float* v, int64_t R, int64_t C, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int64_t step, const float* grad_norm_sq, float max_grad_norm, void* w16, int64_t ld16, void* w16t,
                              int64_t ldt, float* ema, float ema_decay, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIDISC_HIP_H */
