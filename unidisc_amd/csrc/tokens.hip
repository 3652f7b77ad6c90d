// Token data path: joint-sequence assembly of a batch from the token dataset fields.
//
// Reference: `Diffusion.update_batch`, token-dataset branch (model.py:183-212) - per sample
//   input_ids      = cat(txt_input_ids, img_input_ids + text_vocab_size)      int64 [Lt + Li]
//   attention_mask = cat(txt_attention_mask, ones(Li))                        bool
//   modality       = 0 on the Lt text positions, 1 on the Li image positions  int64
// over the dataset schema of models/datasets/image_datasets.py:263-281 (txt_input_ids int32, txt_attention_mask bool, img_input_ids int16).
// Here the dataset shards may stay RESIDENT in HBM (288 GB holds ~10^8 samples of the 1.4 B configuration): the host only draws the row
// indices of a batch, and this kernel gathers + widens + shifts in one pass.  idx == NULL: rows 0..B-1 (a host-staged batch).
// HBM-bound byte work: 2.7 KiB read and 17 KiB written per sample at Lt = 128, Li = 1024; one block per (sample, 256-position chunk).
#include "common.h"
#include "../../include/unidisc_hip.h"

namespace {
__global__ __launch_bounds__(256) void assemble_joint_kernel(const int32_t* __restrict__ txt, const uint8_t* __restrict__ txt_mask, const int16_t* __restrict__ img,
                                                             const int64_t* __restrict__ idx, int Lt, int Li, int64_t Vt, int64_t* __restrict__ ids,
                                                             uint8_t* __restrict__ mask, int64_t* __restrict__ modality) {
  const int b = blockIdx.y, L = Lt + Li;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const int64_t row = idx ? idx[b] : b;
  int64_t id;
  uint8_t m;
  if (l < Lt) {
    id = (int64_t)txt[row * Lt + l];
    m = txt_mask ? (txt_mask[row * Lt + l] != 0) : 1;
  } else {
    id = (int64_t)img[row * Li + (l - Lt)] + Vt;
    m = 1;
  }
  const int64_t o = (int64_t)b * L + l;
  ids[o] = id;
  mask[o] = m;
  modality[o] = l < Lt ? 0 : 1;
}
}  // namespace

extern "C" int udm_assemble_joint_tokens(const int32_t* txt, const uint8_t* txt_mask, const int16_t* img, const int64_t* idx, int64_t B, int64_t Lt,
                                         int64_t Li, int64_t Vt, int64_t* ids, uint8_t* mask, int64_t* modality, hipStream_t stream) {
  UDM_CHECK_ARG(ids && mask && modality && B >= 0 && Lt >= 0 && Li >= 0 && Lt + Li > 0, "udm_assemble_joint_tokens: bad arguments");
  UDM_CHECK_ARG((Lt == 0 || txt) && (Li == 0 || img), "udm_assemble_joint_tokens: a field with non-zero length has a null pointer");
  UDM_CHECK_ARG(B <= 65535 && Lt + Li < (1ll << 31), "udm_assemble_joint_tokens: batch of %lld rows / length %lld not supported", (long long)B, (long long)(Lt + Li));
  if (B == 0) return 0;
  const int L = (int)(Lt + Li);
  hipLaunchKernelGGL(assemble_joint_kernel, dim3((L + 255) / 256, (unsigned)B), dim3(256), 0, stream, txt, txt_mask, img, idx, (int)Lt, (int)Li, Vt, ids, mask, modality);
  UDM_CHECK_LAUNCH("udm_assemble_joint_tokens");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Absorbing-state forward process for multimodal (non-interleaved) batches, q_xt model.py:424-587, AFTER its random draws (which stay torch.rand calls in the
// reference's order): move = r_move < move_chance[b]; whole-modality masking - a row drawn for text AND image masks neither; a row drawn for one modality
// moves exactly that modality's positions; xt = move ? [MASK] : x.  Thirteen tensor statements on [B, L] / [B, 1] in the reference, comparisons and selects
// only: the fused form is bit-identical by construction.  thr_* are the thresholds already rounded to fp32 (what `tensor < python_float` compares with).
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void qxt_absorbing_kernel(const int64_t* __restrict__ x, const float* __restrict__ r_move, const float* __restrict__ move_chance,
                                                            const float* __restrict__ r_txt, const float* __restrict__ r_img, float thr_txt, float thr_img,
                                                            const uint8_t* __restrict__ mm, int L, int64_t mask_id, int64_t* __restrict__ xt,
                                                            uint8_t* __restrict__ move, uint8_t* __restrict__ row_txt, uint8_t* __restrict__ row_img,
                                                            uint8_t* __restrict__ row_ignore) {
  const int b = blockIdx.y;
  const int l = blockIdx.x * 256 + threadIdx.x;
  bool mt = false, mi = false;
  if (r_txt) mt = r_txt[b] < thr_txt;
  if (r_img) mi = r_img[b] < thr_img;
  const bool both = mt && mi;
  if (both) { mt = false; mi = false; }
  if (blockIdx.x == 0 && threadIdx.x == 0 && row_txt) { row_txt[b] = mt; row_img[b] = mi; row_ignore[b] = mt || mi; }
  if (l >= L) return;
  const long o = (long)b * L + l;
  bool mv = r_move[o] < move_chance[b];
  if (mt) mv = mm[2 * o] != 0;
  if (mi) mv = mm[2 * o + 1] != 0;
  move[o] = mv;
  xt[o] = mv ? mask_id : x[o];
}
}  // namespace

extern "C" int udm_qxt_absorbing(const int64_t* x, const float* r_move, const float* move_chance, const float* r_txt, const float* r_img, float thr_txt,
                                 float thr_img, const void* modality_mask, int64_t B, int64_t L, int64_t mask_id, int64_t* xt, void* move_indices, void* row_txt,
                                 void* row_img, void* row_ignore, hipStream_t stream) {
  UDM_CHECK_ARG(x && r_move && move_chance && xt && move_indices && B > 0 && L > 0, "udm_qxt_absorbing: bad arguments");
  UDM_CHECK_ARG((!r_txt && !r_img) || (modality_mask && row_txt && row_img && row_ignore), "udm_qxt_absorbing: whole-modality masking needs the modality mask and the row outputs");
  UDM_CHECK_ARG(B <= 65535, "udm_qxt_absorbing: batch of %lld rows not supported", (long long)B);
  hipLaunchKernelGGL(qxt_absorbing_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B), dim3(256), 0, stream, x, r_move, move_chance, r_txt, r_img, thr_txt, thr_img,
                     (const uint8_t*)modality_mask, (int)L, mask_id, xt, (uint8_t*)move_indices, (uint8_t*)row_txt, (uint8_t*)row_img, (uint8_t*)row_ignore);
  UDM_CHECK_LAUNCH("udm_qxt_absorbing");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Timestep sampling and the log-linear noise schedule for one batch (model.py:589-619, models/noise_schedule.py:128-157): from the uniform draws u [n] to
// t, sigma = -log1p(-(1 - eps) t), dsigma = (1 - eps) / (1 - (1 - eps) t) and the move chance 1 - exp(-sigma) - seventeen launches on n-element tensors in the
// reference.  The statements are mirrored operation by operation, each rounded to fp32 (no contraction), INCLUDING torch's habit of dividing by a host scalar as a
// multiplication with its fp32 reciprocal and of `scalar / tensor` as reciprocal-then-multiply: t feeds bit-exact mask comparisons.
// ---------------------------------------------------------------------------------------------
namespace {
#pragma clang fp contract(off)
__global__ void sample_t_noise_kernel(const float* __restrict__ u, int n, int antithetic, float inv_n, float one_minus_seps, float seps, float neg_one_minus_eps,
                                      float one_minus_eps, float* __restrict__ t_out, float* __restrict__ sigma, float* __restrict__ dsigma,
                                      float* __restrict__ move_chance) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float e = u[i];
  if (antithetic) {
    const float off = (float)i * inv_n;        // torch.arange(n) / n
    e = e * inv_n + off;                        // _eps_t / n + offset   (two roundings: contraction is off)
    e = fmodf(e, 1.0f);                         // % 1 (operands are non-negative)
  }
  const float t = one_minus_seps * e + seps;    // (1 - sampling_eps) * _eps_t + sampling_eps
  t_out[i] = t;
  const float sg = -log1pf(neg_one_minus_eps * t);
  sigma[i] = sg;
  const float den = 1.0f - one_minus_eps * t;
  dsigma[i] = (1.0f / den) * one_minus_eps;     // scalar / tensor = reciprocal(tensor) * scalar
  move_chance[i] = 1.0f - expf(-sg);
}
}  // namespace

extern "C" int udm_sample_t_noise(const float* u, int64_t n, int antithetic, float sampling_eps_complement, float sampling_eps, float noise_eps_complement,
                                  float* t, float* sigma, float* dsigma, float* move_chance, hipStream_t stream) {
  UDM_CHECK_ARG(u && t && sigma && dsigma && move_chance && n > 0, "udm_sample_t_noise: bad arguments");
  const float inv_n = 1.0f / (float)n;
  hipLaunchKernelGGL(sample_t_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, u, (int)n, antithetic, inv_n, sampling_eps_complement, sampling_eps,
                     -noise_eps_complement, noise_eps_complement, t, sigma, dsigma, move_chance);
  UDM_CHECK_LAUNCH("udm_sample_t_noise");
  return 0;
}
