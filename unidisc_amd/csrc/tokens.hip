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

// ---------------------------------------------------------------------------------------------
// Interleaved / packed rows: the per-position LAYOUT work of a packed batch as kernels instead of ~100 small tensor statements
// (each a launch; two of them with hidden host reads - bincount's size, nonzero - that stalled the launch queue at the top of every step:
// 2.7 ms of idle GPU per 81 ms step of the packed 4608-token workload).
//
// (1) udm_interleaved_rope - models/dit.py:1421-1444 with add_img_data_to_blocks / add_txt_data_to_blocks (:122-191): per position the rotary row and
//     the image-count-embedding row.  Image RUNS are maximal stretches of modality != 0; a run whose length is one of the supported block sizes reads that
//     size's 2-D table from its own start and gets count index j = number of earlier image runs of the row that START in the same sample id; text positions
//     (modality 0, sample id >= 0) read the 1-D table from the start of their run of one sample id; everything else gets cos = sin = 0, index -1.
// (2) udm_interleaved_block_lottery - model.py:483-522: blocks = runs of constant (modality, sample id); a block longer than 4 tokens with id >= 0 is a
//     candidate; candidates are numbered in (row, position) order and read uniform r[rank]; k = index of the block among the candidates of its (row, id),
//     n their number; the block is masked as a whole iff r < fp32(mask_prob) * (fp32(k + 1) / fp32(n)) * 2 (the reference's operation order).
// (3) udm_rowgroup_sum_f32 - out[g] += sum of the rows of x whose group index is g (the image-count embedding's gradient: an index_add_ of 9 k rows onto
//     17 addresses per column is 19 M contended atomics, 225 us; here the sums are formed in LDS first).
// One workgroup per row walks it serially in lane 0 for the run structure (L <= 16 k positions: tens of microseconds, rows in parallel), all lanes then write
// the per-position outputs.  Integer / comparison work only apart from the one threshold: bit-identical to the tensor statements (tests).
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int IL_MAX_SIZES = 8;
struct RopeLayoutArgs {
  const int64_t* modality; const int64_t* sid;
  const float* img_cos; const float* img_sin;   // [sum sizes, half]
  const float* txt_cos; const float* txt_sin;   // [txt_rows, half]
  float* cos; float* sin; int64_t* count_idx;   // [B, L, half], [B, L]
  int* scratch;                                 // [B, 5, L] ints: run_start | run_len (at a run's start) | j (at a run's start) | s_run_start | list of run starts
  int L, half, txt_rows, nsizes;
  int sizes[IL_MAX_SIZES], base[IL_MAX_SIZES];
};

// Block-wide inclusive prefix MAX over the 256 per-thread values (carry of a chunked scan): returns the EXCLUSIVE prefix for this thread.
__device__ __forceinline__ int block_excl_prefix_max(int v, int* sm) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int u = threadIdx.x >= o ? sm[threadIdx.x - o] : -1;
    __syncthreads();
    sm[threadIdx.x] = max(sm[threadIdx.x], u);
    __syncthreads();
  }
  const int r = threadIdx.x ? sm[threadIdx.x - 1] : -1;
  __syncthreads();
  return r;
}
__device__ __forceinline__ int block_excl_suffix_min(int v, int* sm, int big) {   // exclusive suffix minimum (threads after this one)
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int u = threadIdx.x + o < 256 ? sm[threadIdx.x + o] : big;
    __syncthreads();
    sm[threadIdx.x] = min(sm[threadIdx.x], u);
    __syncthreads();
  }
  const int r = threadIdx.x < 255 ? sm[threadIdx.x + 1] : big;
  __syncthreads();
  return r;
}
__device__ __forceinline__ int block_excl_prefix_sum(int v, int* sm) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int u = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += u;
    __syncthreads();
  }
  const int r = threadIdx.x ? sm[threadIdx.x - 1] : 0;
  __syncthreads();
  return r;
}

// One workgroup per row.  The run structure comes from three chunked block scans (every thread owns a contiguous chunk of the row): prefix-max of the image-run starts
// and of the sample-id changes, suffix-min of the positions that end an image run; the rank j of a run start among the starts with the same sample id from the compacted
// list of starts (a row has a handful).
__global__ __launch_bounds__(256) void interleaved_rope_kernel(RopeLayoutArgs a) {
  __shared__ int sm[256];
  const int b = blockIdx.x, L = a.L, tid = threadIdx.x;
  const int64_t* mod = a.modality + (long)b * L;
  const int64_t* sid = a.sid + (long)b * L;
  int* run_start = a.scratch + (long)b * 5 * L;
  int* run_end = run_start + L;
  int* jat = run_end + L;
  int* s_start = jat + L;
  int* starts = s_start + L;
  const int C = (L + 255) / 256, l0 = min(tid * C, L), l1 = min(l0 + C, L);
  // pass A: per-chunk summaries
  int last_start = -1, last_schg = -1, first_end = L, nst = 0;
  for (int l = l0; l < l1; ++l) {
    const bool img = mod[l] != 0, pimg = l > 0 && mod[l - 1] != 0;
    if (img && !pimg) { last_start = l; ++nst; }
    if (l == 0 || sid[l] != sid[l - 1]) last_schg = l;
  }
  for (int l = l1 - 1; l >= l0; --l)
    if (mod[l] == 0) first_end = l;
  const int carry_start = block_excl_prefix_max(last_start, sm);
  const int carry_schg = block_excl_prefix_max(last_schg, sm);
  const int carry_end = block_excl_suffix_min(first_end, sm, L);
  const int st0 = block_excl_prefix_sum(nst, sm);
  // pass B: per-position values of the chunk
  {
    int rs = carry_start, ss = carry_schg, k = st0;
    for (int l = l0; l < l1; ++l) {
      const bool img = mod[l] != 0, pimg = l > 0 && mod[l - 1] != 0;
      if (img && !pimg) { rs = l; starts[k++] = l; }
      if (l == 0 || sid[l] != sid[l - 1]) ss = l;
      run_start[l] = img ? rs : -1;
      s_start[l] = ss;
    }
    int re = carry_end;
    for (int l = l1 - 1; l >= l0; --l) {
      if (mod[l] == 0) re = l;
      run_end[l] = re;
    }
  }
  __syncthreads();
  // total number of starts = prefix of the last thread + its count
  if (tid == 255) sm[0] = st0 + nst;
  __syncthreads();
  const int nstarts = sm[0];
  for (int q = tid; q < nstarts; q += 256) {
    const int l = starts[q];
    const int64_t s = sid[l];
    int j = 0;
    for (int e = 0; e < q; ++e) j += (sid[starts[e]] == s);
    jat[l] = j;
  }
  __syncthreads();
  const int half = a.half;
  for (int l = tid; l < L; l += 256) {
    const bool img = mod[l] != 0;
    const float* cs = nullptr;
    const float* sn = nullptr;
    int64_t cidx = -1;
    if (img) {
      const int rs = run_start[l], len = run_end[l] - rs;
      int base = -1;
      for (int k = 0; k < a.nsizes; ++k) base = (len == a.sizes[k]) ? a.base[k] : base;
      if (base >= 0) {
        const long row = base + (l - rs);
        cs = a.img_cos + row * half;
        sn = a.img_sin + row * half;
        cidx = jat[rs];
      }
    } else if (sid[l] >= 0) {
      int t = l - s_start[l];
      t = t > a.txt_rows - 1 ? a.txt_rows - 1 : t;
      cs = a.txt_cos + (long)t * half;
      sn = a.txt_sin + (long)t * half;
    }
    float* co = a.cos + ((long)b * L + l) * half;
    float* so = a.sin + ((long)b * L + l) * half;
    for (int c = 0; c < half; c += 4) {
      *reinterpret_cast<float4*>(co + c) = cs ? *reinterpret_cast<const float4*>(cs + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(so + c) = sn ? *reinterpret_cast<const float4*>(sn + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    a.count_idx[(long)b * L + l] = cidx;
  }
}

struct LotteryArgs {
  const int64_t* modality; const int64_t* sid; const float* r;   // r [n_r] uniforms: candidate i of the batch (row-major order) reads r[i]
  uint8_t* accum; uint8_t* rows_hit; int64_t* n_cand;             // [B, L], [B], [1]
  int* scratch;                                                   // [B, 4, L] ints: block start | block end (per position) | candidate starts (list) | hit flag (at a block's start)
  int* row_cands;                                                 // [B] candidates per row
  int B, L, n_r;
  float mask_prob;
};

// pass 1, one workgroup per row: blocks = runs of constant (modality, sample id) by two chunked scans, the row's candidate blocks compacted in position order
__global__ __launch_bounds__(256) void lottery_blocks_kernel(LotteryArgs a) {
  __shared__ int sm[256];
  const int b = blockIdx.x, L = a.L, tid = threadIdx.x;
  const int64_t* mod = a.modality + (long)b * L;
  const int64_t* sid = a.sid + (long)b * L;
  int* bstart = a.scratch + (long)b * 4 * L;
  int* bend = bstart + L;
  int* cstart = bend + L;
  const int C = (L + 255) / 256, l0 = min(tid * C, L), l1 = min(l0 + C, L);
  auto boundary = [&](int l) { return l == 0 || mod[l] != mod[l - 1] || sid[l] != sid[l - 1]; };
  int last_b = -1, first_b = L;
  for (int l = l0; l < l1; ++l)
    if (boundary(l)) { last_b = l; if (first_b == L) first_b = l; }
  const int carry_b = block_excl_prefix_max(last_b, sm);
  const int carry_e = block_excl_suffix_min(first_b, sm, L);
  {
    int bs = carry_b;
    for (int l = l0; l < l1; ++l) {
      if (boundary(l)) bs = l;
      bstart[l] = bs;
    }
    int be = carry_e;   // the next boundary strictly behind l
    for (int l = l1 - 1; l >= l0; --l) {
      bend[l] = be;
      if (boundary(l)) be = l;
    }
  }
  __syncthreads();
  int nc = 0;
  for (int l = l0; l < l1; ++l) nc += (boundary(l) && sid[l] >= 0 && bend[l] - l > 4);
  const int c0 = block_excl_prefix_sum(nc, sm);
  int k = c0;
  for (int l = l0; l < l1; ++l)
    if (boundary(l) && sid[l] >= 0 && bend[l] - l > 4) cstart[k++] = l;
  if (tid == 255) a.row_cands[b] = c0 + nc;
}

// pass 2, one workgroup per row: every candidate reads its uniform (rank = candidates of the rows before + its index in the row) and is decided; the row is painted
__global__ __launch_bounds__(256) void lottery_decide_kernel(LotteryArgs a) {
  __shared__ int s_any;
  const int b = blockIdx.x, L = a.L, tid = threadIdx.x;
  const int64_t* sid = a.sid + (long)b * L;
  const int* bstart = a.scratch + (long)b * 4 * L;
  const int* bend = bstart + L;
  const int* cstart = bend + L;
  int* hitf = a.scratch + (long)b * 4 * L + 3 * (long)L;
  int rank0 = 0;
  for (int q = 0; q < b; ++q) rank0 += a.row_cands[q];
  const int nc = a.row_cands[b];
  if (b == a.B - 1 && tid == 0) *a.n_cand = rank0 + nc;
  if (tid == 0) s_any = 0;
  __syncthreads();
  for (int q = tid; q < nc; q += 256) {
    const int l = cstart[q];
    const int64_t s = sid[l];
    int k = 0, n = 0;   // k: candidates of (row, s) in front of this block, n: all of them
    for (int e = 0; e < nc; ++e)
      if (sid[cstart[e]] == s) { ++n; k += (e < q); }
    const float thr = (a.mask_prob * ((float)(k + 1) / (float)n)) * 2.f;
    const int rank = rank0 + q;
    const bool hit = rank < a.n_r && a.r[rank] < thr;
    hitf[l] = hit ? 1 : 0;
    if (hit) s_any = 1;
  }
  __syncthreads();
  uint8_t* acc = a.accum + (long)b * L;
  for (int l = tid; l < L; l += 256) {
    const int bs = bstart[l];
    const bool cand = sid[bs] >= 0 && bend[bs] - bs > 4;
    acc[l] = (cand && hitf[bs]) ? 1 : 0;
  }
  if (tid == 0) a.rows_hit[b] = s_any ? 1 : 0;
}

// out[g, c] += sum over rows r with group[r] == g of x[r, c];  G <= 32 groups (rows with group >= G or < 0 are skipped)
__global__ __launch_bounds__(256) void rowgroup_sum_kernel(const float* __restrict__ x, const int64_t* __restrict__ group, float* __restrict__ out, long M, int d, int G,
                                                           int rows_per_block) {
  __shared__ float acc[32 * 64];
  const int c0 = blockIdx.x * 64, tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < G * 64; i += 256) acc[i] = 0.f;
  __syncthreads();
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  if (c0 + tc < d) {
    // consecutive rows mostly share a group (image blocks are runs): sum a run in a register, flush on change
    float run = 0.f;
    long g_run = -1;
    for (long r = r0 + tr; r < r1; r += 4) {
      const long g = group[r];
      if (g != g_run) {
        if (g_run >= 0 && g_run < G) atomicAdd(&acc[g_run * 64 + tc], run);
        run = 0.f;
        g_run = g;
      }
      run += x[r * d + c0 + tc];
    }
    if (g_run >= 0 && g_run < G) atomicAdd(&acc[g_run * 64 + tc], run);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G * 64; i += 256) {
    const int g = i >> 6, c = i & 63;
    if (c0 + c < d && acc[i] != 0.f) atomicAdd(out + (long)g * d + c0 + c, acc[i]);
  }
}
}  // namespace

extern "C" int udm_interleaved_rope(const int64_t* modality, const int64_t* sid, const float* img_cos, const float* img_sin, const int32_t* sizes, int64_t nsizes,
                                    const float* txt_cos, const float* txt_sin, int64_t txt_rows, int64_t B, int64_t L, int64_t half, float* cos, float* sin,
                                    int64_t* count_idx, int32_t* scratch, hipStream_t stream) {
  UDM_CHECK_ARG(modality && sid && img_cos && img_sin && txt_cos && txt_sin && cos && sin && count_idx && scratch && sizes, "udm_interleaved_rope: null pointer");
  UDM_CHECK_ARG(B > 0 && L > 0 && L < (1 << 24) && half > 0 && half % 4 == 0 && txt_rows > 0 && nsizes >= 0 && nsizes <= IL_MAX_SIZES, "udm_interleaved_rope: bad shape");
  RopeLayoutArgs a{};
  a.modality = modality; a.sid = sid; a.img_cos = img_cos; a.img_sin = img_sin; a.txt_cos = txt_cos; a.txt_sin = txt_sin;
  a.cos = cos; a.sin = sin; a.count_idx = count_idx; a.scratch = scratch;
  a.L = (int)L; a.half = (int)half; a.txt_rows = (int)txt_rows; a.nsizes = (int)nsizes;
  int base = 0;
  for (int k = 0; k < (int)nsizes; ++k) { a.sizes[k] = sizes[k]; a.base[k] = base; base += sizes[k]; }   // `sizes` is a HOST array (a handful of ints)
  hipLaunchKernelGGL(interleaved_rope_kernel, dim3((unsigned)B), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_interleaved_rope");
  return 0;
}

extern "C" int udm_interleaved_block_lottery(const int64_t* modality, const int64_t* sid, const float* r, int64_t n_r, float mask_prob, int64_t B, int64_t L,
                                             uint8_t* accum, uint8_t* rows_hit, int64_t* n_cand, int32_t* scratch, int32_t* row_cands, hipStream_t stream) {
  UDM_CHECK_ARG(modality && sid && accum && rows_hit && n_cand && scratch && row_cands && (r || n_r == 0), "udm_interleaved_block_lottery: null pointer");
  UDM_CHECK_ARG(B > 0 && L > 0 && L < (1 << 24) && n_r >= 0, "udm_interleaved_block_lottery: bad shape");
  LotteryArgs a{};
  a.modality = modality; a.sid = sid; a.r = r; a.accum = accum; a.rows_hit = rows_hit; a.n_cand = n_cand; a.scratch = scratch; a.row_cands = row_cands;
  a.B = (int)B; a.L = (int)L; a.n_r = (int)n_r; a.mask_prob = mask_prob;
  hipLaunchKernelGGL(lottery_blocks_kernel, dim3((unsigned)B), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(lottery_decide_kernel, dim3((unsigned)B), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_interleaved_block_lottery");
  return 0;
}

extern "C" int udm_rowgroup_sum_f32(const float* x, const int64_t* group, float* out, int64_t M, int64_t d, int64_t G, hipStream_t stream) {
  UDM_CHECK_ARG(x && group && out && M > 0 && d > 0 && G > 0 && G <= 32, "udm_rowgroup_sum_f32: bad arguments (at most 32 groups)");
  const int rows_per_block = 512;
  hipLaunchKernelGGL(rowgroup_sum_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, stream, x, group, out, (long)M,
                     (int)d, (int)G, rows_per_block);
  UDM_CHECK_LAUNCH("udm_rowgroup_sum_f32");
  return 0;
}
