// SUBS-parameterised cross-entropy over the joint text+image vocabulary, on materialised bf16 logits.
//
// Replaces Diffusion._subs_parameterization + the fp32 cast + gather of the reference (model.py:621-658,
// :924-925, :967): ~10 full [M,V] passes and 3-4 full-size temporaries there; here one read of the
// logits row per MASKED token (unmasked tokens have log p = 0 by construction and are never read), fp32
// online log-sum-exp over the ids that are valid for the token's modality, and the gather of logit[x0].
// The backward overwrites the logits buffer in place with d logits = g * (onehot - softmax_valid).
#include "common.h"
#include "../../include/unidisc_hip.h"

namespace {
using namespace udm;
constexpr float NEG = -1000000.0f;  // reference neg_infinity (model_setup.py:269): finite on purpose

struct CeArgs {
  bf16_t* logits;      // [M, ld] (ld >= V, multiple of 8); backward writes in place
  long ld;
  const int64_t* x0;
  const int64_t* xt;
  const int64_t* modality;  // nullable
  float* log_p;        // [M]
  float* lse;          // [M] natural-log LSE over valid ids (masked rows)
  const float* g;      // [M] upstream d loss / d log_p (backward)
  int M, V, Vt, mask_id, restrict_modality;
  int narrow_txt_rows = -1;   // backward, head per modality (dit.py split_head): rows [0, n) belong to the text group, the rest to the image group; -1 = off
};

__device__ __forceinline__ void valid_range(const CeArgs& a, long row, int& lo, int& hi) {
  // valid ids = [lo, hi) minus mask_id.  force_argmax_valid_indices (model.py:627-635): text rows keep text ids,
  // image rows keep image ids.
  lo = 0; hi = a.V;
  if (a.restrict_modality) {
    const bool img = a.modality && a.modality[row] == 1;
    if (img) lo = a.Vt; else hi = a.Vt;
  }
}

__device__ __forceinline__ void block_reduce_ms(float& m, float& s, float* sm, float* ss) {
  // combine (max, sum-exp) pairs over the block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
  }
  if (lane == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  float M = sm[0], S = ss[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float m2 = sm[w], s2 = ss[w];
    const float mn = fmaxf(M, m2);
    S = (M == -INFINITY ? 0.f : S * __expf(M - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    M = mn;
  }
  m = M; s = S;
}

__global__ __launch_bounds__(256) void subs_ce_fwd_kernel(CeArgs a) {
  __shared__ float sm[4], ss[4];
  const long row = blockIdx.x;
  const int tid = threadIdx.x;
  const long x0 = a.x0[row], xt = a.xt[row];
  if (xt != a.mask_id) {  // unmasked token: one-hot log-prob (model.py:646-656)
    if (tid == 0) { a.log_p[row] = (x0 == xt) ? 0.f : NEG; a.lse[row] = 0.f; }
    return;
  }
  int lo, hi;
  valid_range(a, row, lo, hi);
  const bf16_t* z = a.logits + row * a.ld;
  float m = -INFINITY, s = 0.f;
  const int c_begin = (lo / 8) * 8;
  for (int c = c_begin + tid * 8; c < hi; c += 256 * 8) {
    uint4 u = *reinterpret_cast<const uint4*>(z + c);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    float v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    float cm = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int id = c + k;
      const bool ok = id >= lo && id < hi && id != a.mask_id;
      v[k] = ok ? v[k] : -INFINITY;
      cm = fmaxf(cm, v[k]);
    }
    if (cm > m) { s = (m == -INFINITY) ? 0.f : s * __expf(m - cm); m = cm; }
    if (m != -INFINITY) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s += __expf(v[k] - m);
    }
  }
  block_reduce_ms(m, s, sm, ss);
  if (tid == 0) {
    const float lse = m + __logf(s);
    const bool x0_ok = x0 >= lo && x0 < hi && x0 != a.mask_id && x0 >= 0 && x0 < a.V;
    const float zx = x0_ok ? bf2f(z[x0]) : NEG;
    a.lse[row] = lse;
    a.log_p[row] = zx - lse;
  }
}

__global__ __launch_bounds__(256) void subs_ce_bwd_kernel(CeArgs a) {
  const long row = blockIdx.x;
  const int tid = threadIdx.x;
  bf16_t* z = a.logits + row * a.ld;
  const long xt = a.xt[row];
  const float g = a.g[row];
  // head per modality (dit.py split_head): only the columns the row's GROUP of GEMMs reads are written - rows of the text group [0, ceil64(Vt)), rows of the
  // image group [floor8(Vt), ld) - instead of the whole row; the rest of the buffer is never read.  The group is the row's position in the compacted list, not
  // its modality: padding rows (unmasked, any modality) sit in either group.
  int w_lo = 0, w_hi = (int)a.ld;
  if (a.narrow_txt_rows >= 0) {
    if (row >= a.narrow_txt_rows) w_lo = a.Vt / 8 * 8; else w_hi = min((a.Vt + 63) / 64 * 64, (int)a.ld);
  }
  if (xt != a.mask_id || g == 0.f) {
    for (int c = w_lo + tid * 8; c < w_hi; c += 256 * 8) *reinterpret_cast<uint4*>(z + c) = make_uint4(0, 0, 0, 0);
    return;
  }
  int lo, hi;
  valid_range(a, row, lo, hi);
  const long x0 = a.x0[row];
  const float lse = a.lse[row];
  for (int c = w_lo + tid * 8; c < w_hi; c += 256 * 8) {
    uint4 u = *reinterpret_cast<const uint4*>(z + c);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    float v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int id = c + k;
      const bool ok = id >= lo && id < hi && id != a.mask_id;
      v[k] = ok ? g * ((id == x0 ? 1.f : 0.f) - __expf(v[k] - lse)) : 0.f;
    }
    *reinterpret_cast<uint4*>(z + c) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  }
}

// full SUBS log-probabilities [M, V] (samplers / Diffusion.forward contract, model.py:621-658)
template <bool OUT_F32>
__global__ __launch_bounds__(256) void subs_logprobs_kernel(CeArgs a, void* out, long ld_out) {
  __shared__ float sm[4], ss[4];
  const long row = blockIdx.x;
  const int tid = threadIdx.x;
  const bf16_t* z = a.logits + row * a.ld;
  const long xt = a.xt ? a.xt[row] : a.mask_id;
  const bool masked = xt == a.mask_id;
  int lo, hi;
  valid_range(a, row, lo, hi);
  float lse = 0.f;
  if (masked) {
    float m = -INFINITY, s = 0.f;
    for (int c = tid; c < a.V; c += 256) {
      const bool ok = c >= lo && c < hi && c != a.mask_id;
      if (ok) {
        const float v = bf2f(z[c]);
        if (v > m) { s = (m == -INFINITY) ? 0.f : s * __expf(m - v); m = v; }
        s += __expf(v - m);
      }
    }
    block_reduce_ms(m, s, sm, ss);
    lse = m + __logf(s);
  }
  for (int c = tid; c < a.V; c += 256) {
    float o;
    if (masked) {
      const bool ok = c >= lo && c < hi && c != a.mask_id;
      o = ok ? bf2f(z[c]) - lse : NEG;
    } else {
      o = (c == xt) ? 0.f : NEG;
    }
    if (OUT_F32) reinterpret_cast<float*>(out)[row * ld_out + c] = o;
    else reinterpret_cast<bf16_t*>(out)[row * ld_out + c] = f2bf(o);
  }
}

int check(const char* name, const void* logits, int64_t M, int64_t V, int64_t ld, int64_t Vt, int64_t mask_id) {
  UDM_CHECK_ARG(logits, "%s: null logits", name);
  UDM_CHECK_ARG(M > 0 && V > 0 && ld >= V && ld % 8 == 0, "%s: bad shape M=%ld V=%ld ld=%ld (ld must be a multiple of 8)", name, (long)M, (long)V, (long)ld);
  UDM_CHECK_ARG(Vt > 0 && Vt <= V && mask_id >= 0 && mask_id < V, "%s: bad vocabulary split", name);
  UDM_CHECK_ARG(((uintptr_t)logits % 16) == 0, "%s: logits must be 16-byte aligned", name);
  return 0;
}
}  // namespace

extern "C" int udm_subs_ce_fwd(const void* logits, int64_t ld, const int64_t* x0, const int64_t* xt, const int64_t* modality, float* log_p, float* lse,
                               int64_t M, int64_t V, int64_t Vt, int64_t mask_id, int restrict_modality, hipStream_t stream) {
  if (int rc = check("udm_subs_ce_fwd", logits, M, V, ld, Vt, mask_id)) return rc;
  UDM_CHECK_ARG(x0 && xt && log_p && lse, "udm_subs_ce_fwd: null pointer");
  UDM_CHECK_ARG(!restrict_modality || modality, "udm_subs_ce_fwd: restrict_modality needs the modality map");
  CeArgs a{(bf16_t*)logits, (long)ld, x0, xt, modality, log_p, lse, nullptr, (int)M, (int)V, (int)Vt, (int)mask_id, restrict_modality};
  hipLaunchKernelGGL(subs_ce_fwd_kernel, dim3((unsigned)M), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_subs_ce_fwd");
  return 0;
}

extern "C" int udm_subs_ce_bwd(void* logits, int64_t ld, const int64_t* x0, const int64_t* xt, const int64_t* modality, const float* lse, const float* g,
                               int64_t M, int64_t V, int64_t Vt, int64_t mask_id, int restrict_modality, int64_t narrow_txt_rows, hipStream_t stream) {
  if (int rc = check("udm_subs_ce_bwd", logits, M, V, ld, Vt, mask_id)) return rc;
  UDM_CHECK_ARG(x0 && xt && lse && g, "udm_subs_ce_bwd: null pointer");
  UDM_CHECK_ARG(!restrict_modality || modality, "udm_subs_ce_bwd: restrict_modality needs the modality map");
  UDM_CHECK_ARG(narrow_txt_rows < 0 || (restrict_modality && narrow_txt_rows <= M), "udm_subs_ce_bwd: narrow_txt_rows needs restrict_modality and lies in [0, M]");
  CeArgs a{(bf16_t*)logits, (long)ld, x0, xt, modality, nullptr, const_cast<float*>(lse), g, (int)M, (int)V, (int)Vt, (int)mask_id, restrict_modality};
  a.narrow_txt_rows = (int)narrow_txt_rows;
  hipLaunchKernelGGL(subs_ce_bwd_kernel, dim3((unsigned)M), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_subs_ce_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// One reverse-diffusion step for [MASK] rows without materialising [rows, V] probabilities:
//   p      = exp(SUBS log-probs)                      `_ddpm_forward` model_eval.py:1761-1834 (no-CFG branch), `_subs_parameterization`
//   q_i    = p_i (t - s),  q_mask = s                 `_ddpm_caching_update` model_eval.py:2073-2106
//   x_new  = argmax_i q_i / (1e-10 - log(u_i + 1e-10))   `_sample_categorical` model_utils.py:95-97  (first index on ties, like torch.argmax)
// u comes from the caller (parity runs replay the reference's rand stream) or from Philox.  greedy != 0: argmax of the log-probs instead
// (`noise_removal`, model_eval.py:2437-2444).
// ------------------------------------------------------------------------------------------------
struct SampleArgs {
  const bf16_t* logits; long ld;
  const bf16_t* logits_u;    // classifier-free guidance: logits of the same rows with the conditioning masked (nullable = no guidance)
  const float* w;            // per row guidance weight: z = (1 + w) logits - w logits_u   (`_ddpm_forward` model_eval.py:1808-1811)
  const int64_t* modality;   // per row, nullable
  const float* t; const float* s;   // per row: move chance at this step / at the next one
  const float* u; long ldu;  // explicit uniforms [rows, ldu >= V], nullable
  uint64_t seed;
  int64_t* out;
  int V, Vt, mask_id, restrict_modality, greedy;
  // categorical mode (`_maskgit_update` model_eval.py:3069-3071): draw x ~ p (mask id has probability 0) - i.e. t - s = 1, s = 0 - or take the
  // token from `given` (replay), and report log p(x) in out_logp
  int categorical;
  const int64_t* given;
  float* out_logp;
};

__global__ __launch_bounds__(256) void ddpm_sample_rows_kernel(SampleArgs a) {
  __shared__ float sm[4], ss[4];
  __shared__ float bs[4];
  __shared__ int bi[4];
  const long row = blockIdx.x;
  const int tid = threadIdx.x;
  int lo = 0, hi = a.V;
  if (a.restrict_modality) {
    const bool img = a.modality && a.modality[row] == 1;
    if (img) lo = a.Vt; else hi = a.Vt;
  }
  const bf16_t* zc = a.logits + row * a.ld;
  const bf16_t* zu = a.logits_u ? a.logits_u + row * a.ld : nullptr;
  const float gw = zu ? a.w[row] : 0.f;
  auto Z = [&](int c) {   // guided logit in fp32 (the reference mixes fp32-promoted logits, then applies SUBS)
    const float v = bf2f(zc[c]);
    return zu ? (1.0f + gw) * v - gw * bf2f(zu[c]) : v;
  };
  float m = -INFINITY, s = 0.f;
  for (int c = tid; c < a.V; c += 256) {
    const bool ok = c >= lo && c < hi && c != a.mask_id;
    if (!ok) continue;
    const float v = Z(c);
    if (v > m) { s = (m == -INFINITY) ? 0.f : s * __expf(m - v); m = v; }
    s += __expf(v - m);
  }
  block_reduce_ms(m, s, sm, ss);
  const float lse = m + __logf(s);
  const float dt = a.categorical ? 1.f : (a.greedy ? 0.f : a.t[row] - a.s[row]), sr = (a.greedy || a.categorical) ? 0.f : a.s[row];
  float best = -INFINITY;
  int besti = 0x7fffffff;
  if (a.given) {   // replayed draw: only its log-probability is wanted
    if (tid == 0) {
      const int c = (int)a.given[row];
      a.out[row] = c;
      if (a.out_logp) a.out_logp[row] = (c >= lo && c < hi && c != a.mask_id) ? Z(c) - lse : -INFINITY;
    }
    return;
  }
  for (int c = tid; c < a.V; c += 256) {
    const bool ok = c >= lo && c < hi && c != a.mask_id;
    float score;
    if (a.greedy) {
      score = ok ? Z(c) - lse : NEG;   // the reference's argmax runs over log-probs with invalid ids at -1e6
      if (c == a.mask_id) score = NEG + Z(c) - lse;
    } else {
      const float q = (c == a.mask_id) ? sr : (ok ? __expf(Z(c) - lse) * dt : 0.f);
      float uu;
      if (a.u) {
        uu = a.u[row * a.ldu + c];
      } else {
        const uint4 r = philox4x32(a.seed, (uint64_t)row * (uint64_t)((a.V + 3) / 4) + (uint64_t)(c >> 2));
        const uint32_t w = (c & 3) == 0 ? r.x : (c & 3) == 1 ? r.y : (c & 3) == 2 ? r.z : r.w;
        uu = (float)(w >> 8) * (1.0f / 16777216.0f);
      }
      score = q / (1e-10f - __logf(uu + 1e-10f));
    }
    if (score > best) { best = score; besti = c; }   // c increases per thread: the first maximum is kept
  }
  // block argmax, smallest index on ties
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float b2 = __shfl_xor(best, o, 64);
    const int i2 = __shfl_xor(besti, o, 64);
    if (b2 > best || (b2 == best && i2 < besti)) { best = b2; besti = i2; }
  }
  if ((tid & 63) == 0) { bs[tid >> 6] = best; bi[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (bs[w] > best || (bs[w] == best && bi[w] < besti)) { best = bs[w]; besti = bi[w]; }
    a.out[row] = besti;
    if (a.out_logp) a.out_logp[row] = Z(besti) - lse;
  }
}

extern "C" int udm_subs_logprobs(const void* logits, int64_t ld, const int64_t* xt, const int64_t* modality, void* out, int64_t ld_out, int out_f32, int64_t M,
                                 int64_t V, int64_t Vt, int64_t mask_id, int restrict_modality, hipStream_t stream) {
  if (int rc = check("udm_subs_logprobs", logits, M, V, ld, Vt, mask_id)) return rc;
  UDM_CHECK_ARG(out && ld_out >= V, "udm_subs_logprobs: bad output");
  UDM_CHECK_ARG(!restrict_modality || modality, "udm_subs_logprobs: restrict_modality needs the modality map");
  CeArgs a{(bf16_t*)logits, (long)ld, nullptr, xt, modality, nullptr, nullptr, nullptr, (int)M, (int)V, (int)Vt, (int)mask_id, restrict_modality};
  if (out_f32)
    hipLaunchKernelGGL(subs_logprobs_kernel<true>, dim3((unsigned)M), dim3(256), 0, stream, a, out, (long)ld_out);
  else
    hipLaunchKernelGGL(subs_logprobs_kernel<false>, dim3((unsigned)M), dim3(256), 0, stream, a, out, (long)ld_out);
  UDM_CHECK_LAUNCH("udm_subs_logprobs");
  return 0;
}

extern "C" int udm_ddpm_sample_rows_cfg(const void* logits, const void* logits_uncond, const float* w, int64_t ld, const int64_t* modality, const float* t,
                                        const float* s, const float* u, int64_t ldu, uint64_t seed, int64_t* out, int64_t M, int64_t V, int64_t Vt,
                                        int64_t mask_id, int restrict_modality, int greedy, hipStream_t stream) {
  if (int rc = check("udm_ddpm_sample_rows", logits, M, V, ld, Vt, mask_id)) return rc;
  UDM_CHECK_ARG(out && (greedy || (t && s)), "udm_ddpm_sample_rows: null pointer");
  UDM_CHECK_ARG(!u || ldu >= V, "udm_ddpm_sample_rows: noise row stride too small");
  UDM_CHECK_ARG(!restrict_modality || modality, "udm_ddpm_sample_rows: restrict_modality needs the per-row modality");
  UDM_CHECK_ARG((logits_uncond == nullptr) == (w == nullptr), "udm_ddpm_sample_rows_cfg: guidance needs both the unconditional logits and the per-row weights");
  if (M == 0) return 0;
  SampleArgs a{(const bf16_t*)logits, (long)ld, (const bf16_t*)logits_uncond, w, modality, t, s, u, (long)ldu, seed, out, (int)V, (int)Vt, (int)mask_id,
               restrict_modality, greedy, 0, nullptr, nullptr};
  hipLaunchKernelGGL(ddpm_sample_rows_kernel, dim3((unsigned)M), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_ddpm_sample_rows");
  return 0;
}

extern "C" int udm_categorical_sample_rows(const void* logits, const void* logits_uncond, const float* w, int64_t ld, const int64_t* modality, const float* u,
                                           int64_t ldu, uint64_t seed, const int64_t* given, int64_t* out, float* out_logp, int64_t M, int64_t V, int64_t Vt,
                                           int64_t mask_id, int restrict_modality, hipStream_t stream) {
  if (int rc = check("udm_categorical_sample_rows", logits, M, V, ld, Vt, mask_id)) return rc;
  UDM_CHECK_ARG(out && out_logp, "udm_categorical_sample_rows: null output");
  UDM_CHECK_ARG(!u || ldu >= V, "udm_categorical_sample_rows: noise row stride too small");
  UDM_CHECK_ARG(!restrict_modality || modality, "udm_categorical_sample_rows: restrict_modality needs the per-row modality");
  UDM_CHECK_ARG((logits_uncond == nullptr) == (w == nullptr), "udm_categorical_sample_rows: guidance needs both the unconditional logits and the per-row weights");
  if (M == 0) return 0;
  SampleArgs a{(const bf16_t*)logits, (long)ld, (const bf16_t*)logits_uncond, w, modality, nullptr, nullptr, u, (long)ldu, seed, out, (int)V, (int)Vt, (int)mask_id,
               restrict_modality, 0, 1, given, out_logp};
  hipLaunchKernelGGL(ddpm_sample_rows_kernel, dim3((unsigned)M), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_categorical_sample_rows");
  return 0;
}

extern "C" int udm_ddpm_sample_rows(const void* logits, int64_t ld, const int64_t* modality, const float* t, const float* s, const float* u, int64_t ldu,
                                    uint64_t seed, int64_t* out, int64_t M, int64_t V, int64_t Vt, int64_t mask_id, int restrict_modality, int greedy,
                                    hipStream_t stream) {
  return udm_ddpm_sample_rows_cfg(logits, nullptr, nullptr, ld, modality, t, s, u, ldu, seed, out, M, V, Vt, mask_id, restrict_modality, greedy, stream);
}

// ---------------------------------------------------------------------------------------------
// The loss arithmetic behind the per-token log-probabilities (Diffusion.compute_loss, model.py:1010-1160): per-token NLLs with the schedule weight, the
// masked mean or the modality-weighted sum (text / image losses, each mean x token fraction x weight, optional text-loss cap), and d loss / d log_p.
// In the reference this is ~75 elementwise / reduction launches on [B, L] and [B] tensors between the forward and the backward (4-5 us each: 0.35 ms of
// an 86 ms step); here it is one block.  B L is a few 10^4: one 1024-thread block walks it twice (sums, then coefficients).
// scalars: {loss, txt_loss, img_loss, txt_frac, img_frac, valid_frac, txt_count, img_count}
// ---------------------------------------------------------------------------------------------
namespace {
struct LossArgs {
  const float* log_p; const float* w_loss; const float* w_std;   // [N], [B], [B]
  const uint8_t* att; const uint8_t* mm;                          // [N] bool, [N][2] bool (text, image) or null
  float* nlls; float* coef; float* scalars;
  long N; int L; int weighted; int full_mask; float text_w, img_w, ratio;
};
__device__ __forceinline__ float block_sum_1024(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += sm[k];
  return t;
}
__global__ __launch_bounds__(1024) void diffusion_loss_kernel(LossArgs a) {
  __shared__ float sm[16];
  float s_txt = 0.f, s_img = 0.f, n_txt = 0.f, n_img = 0.f, s_all = 0.f, n_all = 0.f, n_att = 0.f;
  for (long i = threadIdx.x; i < a.N; i += 1024) {
    const int b = (int)(i / a.L);
    const float lp = a.log_p[i];
    const bool at = a.att[i] != 0;
    const float tok = -lp * a.w_loss[b];
    a.nlls[i] = at ? -lp * a.w_std[b] : 0.f;
    n_att += at ? 1.f : 0.f;
    if (a.mm) {
      const bool tx = a.mm[2 * i] != 0 && at, im = a.mm[2 * i + 1] != 0 && at;
      if (tx) { s_txt += tok; n_txt += 1.f; }
      if (im) { s_img += tok; n_img += 1.f; }
    }
    const bool am = a.full_mask || at;
    if (am) { s_all += tok; n_all += 1.f; }
  }
  s_txt = block_sum_1024(s_txt, sm); s_img = block_sum_1024(s_img, sm);
  n_txt = block_sum_1024(n_txt, sm); n_img = block_sum_1024(n_img, sm);
  s_all = block_sum_1024(s_all, sm); n_all = block_sum_1024(n_all, sm);
  n_att = block_sum_1024(n_att, sm);
  const float total = n_txt + n_img;
  const float txt_frac = n_txt / total, img_frac = n_img / total;
  float c_txt = 0.f, c_img = 0.f, c_all = 0.f, loss, txt_loss = 0.f, img_loss = 0.f;
  if (a.weighted) {
    txt_loss = ((s_txt / n_txt) * txt_frac) * a.text_w;
    img_loss = ((s_img / n_img) * img_frac) * a.img_w;
    float scale = 1.f;
    if (a.ratio >= 0.f && !(isnan(img_loss) || isnan(txt_loss))) scale = fminf(1.f, (a.ratio * img_loss) / (txt_loss + 1e-8f));
    txt_loss *= scale;
    c_txt = isnan(txt_loss) ? 0.f : (txt_frac * a.text_w * scale) / n_txt;   // nan_to_num: a NaN term contributes 0 and passes no gradient
    c_img = isnan(img_loss) ? 0.f : (img_frac * a.img_w) / n_img;
    if (isnan(txt_loss)) txt_loss = 0.f;
    if (isnan(img_loss)) img_loss = 0.f;
    loss = txt_loss + img_loss;
  } else {
    loss = s_all / n_all;
    c_all = isnan(loss) ? 0.f : 1.f / n_all;
    if (isnan(loss)) loss = 0.f;
  }
  for (long i = threadIdx.x; i < a.N; i += 1024) {
    const int b = (int)(i / a.L);
    const bool at = a.att[i] != 0;
    float c;
    if (a.weighted) c = at ? ((a.mm[2 * i] ? c_txt : 0.f) + (a.mm[2 * i + 1] ? c_img : 0.f)) : 0.f;
    else c = (a.full_mask || at) ? c_all : 0.f;
    a.coef[i] = -a.w_loss[b] * c;
  }
  if (threadIdx.x == 0) {
    a.scalars[0] = loss; a.scalars[1] = txt_loss; a.scalars[2] = img_loss; a.scalars[3] = txt_frac; a.scalars[4] = img_frac;
    a.scalars[5] = n_att / (float)a.N; a.scalars[6] = n_txt; a.scalars[7] = n_img;
  }
}
}  // namespace

extern "C" int udm_diffusion_loss(const float* log_p, const float* w_loss, const float* w_std, const void* attention_mask, const void* modality_mask, float* nlls,
                                  float* coef, float* scalars, int64_t B, int64_t L, int weighted, int full_mask, float text_w, float img_w, float ratio,
                                  hipStream_t stream) {
  UDM_CHECK_ARG(log_p && w_loss && w_std && attention_mask && nlls && coef && scalars, "udm_diffusion_loss: null pointer");
  UDM_CHECK_ARG(B > 0 && L > 0, "udm_diffusion_loss: empty batch");
  UDM_CHECK_ARG(!weighted || modality_mask, "udm_diffusion_loss: the weighted form needs the modality mask");
  LossArgs a{log_p, w_loss, w_std, (const uint8_t*)attention_mask, (const uint8_t*)modality_mask, nlls, coef, scalars, (long)(B * L), (int)L, weighted, full_mask,
             text_w, img_w, ratio};
  hipLaunchKernelGGL(diffusion_loss_kernel, dim3(1), dim3(1024), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_diffusion_loss");
  return 0;
}
