// Row-wise (HBM-bound) kernels of the DiT block: RMSNorm / LayerNorm (+adaLN modulate), the residual
// branch update (sandwich norm, gate, dropout), QK-LayerNorm + rotary, token embedding, small adaLN helpers.
//
// One 64-lane wave owns one row: the row lives in registers (16-byte vector loads, lane-contiguous =
// fully coalesced), row statistics are wavefront-shuffle reductions, nothing is re-read from HBM.
// Reference ops replaced: models/dit.py:77-100 (RMSNorm), :383-403 (LayerNorm), :229-253
// (bias_dropout_add_scale), :263-304 (modulate_fused), :680-682 (qk LayerNorm), models/standalone_rotary.py:14-31
// (rotary), :1036-1043 + :1402-1411 (embedding + modality embedding), :415-449 (timestep embedding).
#include "common.h"
#include <stdlib.h>
#include "../../include/unidisc_hip.h"

namespace {
using namespace udm;

constexpr int ROWS_PER_BLOCK = 4;  // 256 threads = 4 waves = 4 rows in flight per block

__device__ __forceinline__ void load8_f32(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8_f32(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load8_bf16(const bf16_t* p, float (&v)[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __uint_as_float(w[k] << 16);
    v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store8_bf16(bf16_t* p, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

// dropout keep-mask for 8 consecutive elements starting at flat element index e0 (multiple of 8)
// One Philox4x32-10 call (40 quarter-rate integer multiplies) serves all 8 elements: 16 random bits per decision, i.e. the drop
// probability is p rounded to a multiple of 2^-16 (0.1 -> 0.100006).  Half the generator work of one 32-bit word per element - the
// regeneration was about half of the residual kernels' time at p > 0 (forward and backward both rebuild the mask from (seed, index)).
__device__ __forceinline__ void dropout_keep8(uint64_t seed, uint64_t e0, float p, bool (&keep)[8]) {
  const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  const uint4 r = philox4x32(seed, e0 >> 3);
  keep[0] = (r.x & 0xffffu) >= thr; keep[1] = (r.x >> 16) >= thr; keep[2] = (r.y & 0xffffu) >= thr; keep[3] = (r.y >> 16) >= thr;
  keep[4] = (r.z & 0xffffu) >= thr; keep[5] = (r.z >> 16) >= thr; keep[6] = (r.w & 0xffffu) >= thr; keep[7] = (r.w >> 16) >= thr;
}

struct NormArgs {
  const float* x;        // [M,d] fp32 residual stream
  bf16_t* y;             // [M,d] bf16 (GEMM input)
  float* rstd;           // [M]
  float* mean;           // [M] (LayerNorm only)
  const float* w;        // [d]
  const bf16_t* shift;   // adaLN: [B, mod_stride] slices, nullable
  const bf16_t* scale;
  const int64_t* modality;  // [M], nullable
  const int* any_img;       // device flag: modality.any() (reference dit.py:266-268), nullable
  long mod_stride;
  int M, d, L;
  int norm_type;         // 0 rms, 1 layernorm (no bias)
  float eps;
};

template <int NCH>
__global__ __launch_bounds__(256) void norm_fwd_kernel(NormArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool img_only = a.modality && (!a.any_img || *a.any_img != 0);
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < a.M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    float v[NCH][8];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < a.d) {
        load8_f32(a.x + row * a.d + c, v[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) s1 += (a.norm_type == 0) ? v[i][k] * v[i][k] : v[i][k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
      }
    }
    s1 = wave_sum(s1);
    float mu = 0.f, rs;
    if (a.norm_type == 0) {
      rs = rsqrtf(s1 / a.d + a.eps);
    } else {
      mu = s1 / a.d;
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < a.d) {
#pragma unroll
          for (int k = 0; k < 8; ++k) { float t = v[i][k] - mu; s2 += t * t; }
        }
      }
      rs = rsqrtf(wave_sum(s2) / a.d + a.eps);
    }
    if (lane == 0) {
      a.rstd[row] = rs;
      if (a.mean) a.mean[row] = mu;
    }
    const int b = (int)(row / a.L);
    const bool modulate = a.shift && (!img_only || a.modality[row] == 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c >= a.d) continue;
      float w8[8], o[8];
      load8_f32(a.w + c, w8);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu) * rs * w8[k];
      if (modulate) {
        float sh[8], sc[8];
        load8_bf16(a.shift + (long)b * a.mod_stride + c, sh);
        load8_bf16(a.scale + (long)b * a.mod_stride + c, sc);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = o[k] * (1.f + sc[k]) + sh[k];
      }
      store8_bf16(a.y + row * a.d + c, o);
    }
  }
}

struct NormBwdArgs {
  const bf16_t* dy;   // [M,d] grad wrt the (modulated) norm output
  const float* x;     // [M,d]
  const float* rstd;
  const float* mean;  // LN only
  const float* w;
  const bf16_t* shift;
  const bf16_t* scale;
  const int64_t* modality;
  const int* any_img;
  float* dx;          // [M,d] fp32, accumulated into (dx += ...) when accumulate != 0
  float* dw;          // [d] fp32 atomics
  float* dshift;      // [B, mod_stride] fp32 atomics (nullable)
  float* dscale;
  long mod_stride;
  int M, d, L, norm_type, accumulate;
  float* ws;          // optional [gridDim.x, d] fp32: per-block dw partials (two-phase column reduction instead of deep atomic chains)
  int bpb;            // MOD form: blocks per batch element (grid = B * bpb; a block's rows lie inside ONE batch element)
};

// cross-wave sum of one 512-column chunk of per-lane partials through LDS: fn(column within the chunk, sum) on the 512 columns
template <typename F>
__device__ __forceinline__ void block_colsum512(float (*red)[64 * 8 + 8], const float (&acc)[8], int wave, int lane, F fn) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = acc[k];
  __syncthreads();
  for (int t = threadIdx.x; t < 512; t += 256) fn(t, red[0][t] + red[1][t] + red[2][t] + red[3][t]);
}

// MOD (adaLN modulation, models/dit.py:263-304): the gradients of the per-batch-element shift / scale vectors are column sums over that element's L rows.
// A block owns a contiguous run of rows of ONE batch element, keeps the two sums in registers and leaves with one atomic per column (round 5; before: one
// atomic per ELEMENT onto B x d addresses - 1280 deep same-address chains, 1.3 ms per launch at M = 10 240, d = 2048 against 60 us without modulation).
template <int NCH, bool MOD = false>
__global__ __launch_bounds__(256) void norm_bwd_kernel(NormBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool img_only = a.modality && (!a.any_img || *a.any_img != 0);
  float dw_acc[NCH][8];
  float dsh_acc[MOD ? NCH : 1][8], dsc_acc[MOD ? NCH : 1][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dw_acc[i][k] = 0.f;
      if (MOD) { dsh_acc[i][k] = 0.f; dsc_acc[i][k] = 0.f; }
    }
  long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave, row_end = a.M, row_step = (long)gridDim.x * ROWS_PER_BLOCK;
  if (MOD) {
    const int bb = blockIdx.x / a.bpb, chunk = (a.L + a.bpb - 1) / a.bpb;
    const long r0 = (long)bb * a.L + (long)(blockIdx.x % a.bpb) * chunk;
    row = r0 + wave;
    row_end = min((long)(bb + 1) * a.L, r0 + chunk);
    row_step = ROWS_PER_BLOCK;
  }
  for (; row < row_end; row += row_step) {
    const float rs = a.rstd[row];
    const float mu = a.norm_type ? a.mean[row] : 0.f;
    const int b = (int)(row / a.L);
    const bool modulate = a.shift && (!img_only || a.modality[row] == 1);
    float xh[NCH][8], g[NCH][8];
    float s_g = 0.f, s_gx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < a.d) {
        float x8[8], dy8[8], w8[8];
        load8_f32(a.x + row * a.d + c, x8);
        load8_bf16(a.dy + row * a.d + c, dy8);
        load8_f32(a.w + c, w8);
#pragma unroll
        for (int k = 0; k < 8; ++k) xh[i][k] = (x8[k] - mu) * rs;
        if (modulate) {
          float sc[8];
          load8_bf16(a.scale + (long)b * a.mod_stride + c, sc);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (MOD) {
              dsh_acc[i][k] += dy8[k];
              dsc_acc[i][k] += dy8[k] * xh[i][k] * w8[k];
            }
            dy8[k] *= (1.f + sc[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          dw_acc[i][k] += dy8[k] * xh[i][k];
          g[i][k] = dy8[k] * w8[k];
          s_g += g[i][k];
          s_gx += g[i][k] * xh[i][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { xh[i][k] = 0.f; g[i][k] = 0.f; }
      }
    }
    s_gx = wave_sum(s_gx) / a.d;
    s_g = a.norm_type ? wave_sum(s_g) / a.d : 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c >= a.d) continue;
      float o[8];
      if (a.accumulate) load8_f32(a.dx + row * a.d + c, o);
      else {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += rs * (g[i][k] - s_g - xh[i][k] * s_gx);
      store8_f32(a.dx + row * a.d + c, o);
    }
  }
  // cross-wave reduction of dw through LDS, then one atomic per column per block
  __shared__ float red[ROWS_PER_BLOCK][64 * 8 + 8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    block_colsum512(red, dw_acc[i], wave, lane, [&](int t, float s) {
      const int c = i * 512 + t;
      if (c < a.d) {
        if (a.ws) a.ws[(long)blockIdx.x * a.d + c] = s;
        else atomicAdd(a.dw + c, s);
      }
    });
    if (MOD) {
      float* dsh = a.dshift + (long)(blockIdx.x / a.bpb) * a.mod_stride;
      float* dsc = a.dscale + (long)(blockIdx.x / a.bpb) * a.mod_stride;
      block_colsum512(red, dsh_acc[i], wave, lane, [&](int t, float s) { if (i * 512 + t < a.d && s != 0.f) atomicAdd(dsh + i * 512 + t, s); });
      block_colsum512(red, dsc_acc[i], wave, lane, [&](int t, float s) { if (i * 512 + t < a.d && s != 0.f) atomicAdd(dsc + i * 512 + t, s); });
    }
  }
}

// ---------------------------------------------------------------------------------------------
// residual branch:  x_out = x_in + T(branch)
//   T = [sandwich norm(branch; w_b)] -> dropout(p) -> * gate      (gate / dropout skipped on text rows when a
//   modality map is given: reference bias_dropout_add_scale, models/dit.py:239-251)
// ---------------------------------------------------------------------------------------------
struct ResidArgs {
  const float* x_in;
  const bf16_t* branch;
  float* x_out;
  const float* w_b;     // sandwich norm weight, nullable
  float* rstd_b;        // [M] saved (sandwich)
  float* mean_b;        // LN sandwich only
  const bf16_t* gate;   // [B, mod_stride], nullable
  const int64_t* modality;  // nullable: when set, gate+dropout apply to rows with modality==1 only
  long mod_stride;
  int M, d, L, norm_type;
  float eps, p_drop;
  uint64_t seed;
  // optional fused NEXT norm (unmodulated): h_out = norm(x_out; w_n) as bf16 -- saves the separate norm kernel's read of x_out
  const float* w_n;
  bf16_t* h_out;
  float* rstd_n;
  float* mean_n;
  // ... modulated (adaLN-Zero, models/dit.py:263-304): h_out = norm(x_out; w_n) * (1 + scale) + shift on the rows norm_fwd would modulate
  const bf16_t* n_shift = nullptr; const bf16_t* n_scale = nullptr; long n_mod_stride = 0; const int64_t* n_modality = nullptr; const int* n_any_img = nullptr;
};

template <int NCH>
__global__ __launch_bounds__(256) void residual_fwd_kernel(ResidArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < a.M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    float v[NCH][8];
    float xin[NCH][8];   // the residual stream's row is requested together with the branch's (behind the row statistics it would wait a second memory round trip)
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < a.d) load8_f32(a.x_in + row * a.d + c, xin[i]);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < a.d) {
        load8_bf16(a.branch + row * a.d + c, v[i]);
        if (a.w_b) {
#pragma unroll
          for (int k = 0; k < 8; ++k) s1 += (a.norm_type == 0) ? v[i][k] * v[i][k] : v[i][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
      }
    }
    float mu = 0.f, rs = 1.f;
    if (a.w_b) {
      s1 = wave_sum(s1);
      if (a.norm_type == 0) {
        rs = rsqrtf(s1 / a.d + a.eps);
      } else {
        mu = s1 / a.d;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = (i * 64 + lane) * 8;
          if (c < a.d) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { float t = v[i][k] - mu; s2 += t * t; }
          }
        }
        rs = rsqrtf(wave_sum(s2) / a.d + a.eps);
      }
      if (lane == 0) {
        a.rstd_b[row] = rs;
        if (a.mean_b) a.mean_b[row] = mu;
      }
    }
    const int b = (int)(row / a.L);
    const bool special = !a.modality || a.modality[row] == 1;  // row receives gate + dropout
    const float keep_scale = 1.f / (1.f - a.p_drop);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c >= a.d) continue;
      float o[8];
      if (a.w_b) {
        float w8[8];
        load8_f32(a.w_b + c, w8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float n = (v[i][k] - mu) * rs;
          if (a.norm_type == 0) n = rbf(n);  // RMSNorm: `.type_as(x)` on a bf16 input (dit.py:98-100)
          o[k] = n * w8[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = v[i][k];
      }
      if (special) {
        if (a.p_drop > 0.f) {
          bool keep[8];
          dropout_keep8(a.seed, (uint64_t)row * a.d + c, a.p_drop, keep);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = keep[k] ? o[k] * keep_scale : 0.f;
        }
        if (a.gate) {
          float g8[8];
          load8_bf16(a.gate + (long)b * a.mod_stride + c, g8);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] *= g8[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += xin[i][k];
      store8_f32(a.x_out + row * a.d + c, o);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[i][k] = o[k];   // the row of x_out stays in registers for the fused next norm
    }
    if (a.w_n) {  // same arithmetic as norm_fwd_kernel (unmodulated)
      float t1 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < a.d) {
#pragma unroll
          for (int k = 0; k < 8; ++k) t1 += (a.norm_type == 0) ? v[i][k] * v[i][k] : v[i][k];
        }
      }
      t1 = wave_sum(t1);
      float mu2 = 0.f, rs2;
      if (a.norm_type == 0) {
        rs2 = rsqrtf(t1 / a.d + a.eps);
      } else {
        mu2 = t1 / a.d;
        float t2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int c = (i * 64 + lane) * 8;
          if (c < a.d) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { float t = v[i][k] - mu2; t2 += t * t; }
          }
        }
        rs2 = rsqrtf(wave_sum(t2) / a.d + a.eps);
      }
      if (lane == 0) {
        a.rstd_n[row] = rs2;
        if (a.mean_n) a.mean_n[row] = mu2;
      }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c >= a.d) continue;
        float w8[8], o[8];
        load8_f32(a.w_n + c, w8);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (v[i][k] - mu2) * rs2 * w8[k];
        if (a.n_shift && (!(a.n_modality && (!a.n_any_img || *a.n_any_img != 0)) || a.n_modality[row] == 1)) {   // norm_fwd_kernel's rule and arithmetic
          float sh[8], sc[8];
          load8_bf16(a.n_shift + (long)b * a.n_mod_stride + c, sh);
          load8_bf16(a.n_scale + (long)b * a.n_mod_stride + c, sc);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = o[k] * (1.f + sc[k]) + sh[k];
        }
        store8_bf16(a.h_out + row * a.d + c, o);
      }
    }
  }
}

struct ResidBwdArgs {
  const float* dx;        // [M,d] grad wrt x_out (== grad wrt x_in: passes through untouched)
  const bf16_t* branch;   // saved branch output
  bf16_t* dbranch;        // [M,d] out
  const float* w_b;
  const float* rstd_b;
  const float* mean_b;
  const bf16_t* gate;
  const int64_t* modality;
  float* dw_b;            // [d] atomics
  float* ws;              // optional [gridDim.x, d] partial-sum workspace (block-per-row form): no same-address atomic chains
  float* dgate;           // [B, mod_stride] atomics
  long mod_stride;
  int M, d, L, norm_type;
  float p_drop;
  uint64_t seed;
  int bpb;                // GATED form: blocks per batch element (grid = B * bpb), see norm_bwd_kernel<.., MOD>
};

// GATED (adaLN-Zero gate, models/dit.py:229-253): the gate's gradient is a column sum over the batch element's rows - kept in registers by a block whose rows lie
// inside one batch element, one atomic per column per block (before: one atomic per element).
template <int NCH, bool GATED = false>
__global__ __launch_bounds__(256) void residual_bwd_kernel(ResidBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dw_acc[NCH][8];
  float dg_acc[GATED ? NCH : 1][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dw_acc[i][k] = 0.f;
      if (GATED) dg_acc[i][k] = 0.f;
    }
  const float keep_scale = 1.f / (1.f - a.p_drop);
  long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave, row_end = a.M, row_step = (long)gridDim.x * ROWS_PER_BLOCK;
  if (GATED) {
    const int bb = blockIdx.x / a.bpb, chunk = (a.L + a.bpb - 1) / a.bpb;
    const long r0 = (long)bb * a.L + (long)(blockIdx.x % a.bpb) * chunk;
    row = r0 + wave;
    row_end = min((long)(bb + 1) * a.L, r0 + chunk);
    row_step = ROWS_PER_BLOCK;
  }
  for (; row < row_end; row += row_step) {
    const int b = (int)(row / a.L);
    const bool special = !a.modality || a.modality[row] == 1;
    const float rs = a.w_b ? a.rstd_b[row] : 1.f;
    const float mu = (a.w_b && a.norm_type) ? a.mean_b[row] : 0.f;
    float nh[NCH][8], g[NCH][8];
    float s_g = 0.f, s_gx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < a.d) {
        float br[8], dn[8], w8[8];
        load8_bf16(a.branch + row * a.d + c, br);
        load8_f32(a.dx + row * a.d + c, dn);
        if (a.w_b) load8_f32(a.w_b + c, w8);
        // recompute T's intermediate n (post-norm, pre-dropout) for dgate / dw_b
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float n = (br[k] - mu) * rs;
          nh[i][k] = n;
        }
        if (special) {
          float g8[8];
          if (a.gate) load8_bf16(a.gate + (long)b * a.mod_stride + c, g8);
          bool keep[8];
          if (a.p_drop > 0.f) dropout_keep8(a.seed, (uint64_t)row * a.d + c, a.p_drop, keep);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float dm = (a.p_drop > 0.f) ? (keep[k] ? keep_scale : 0.f) : 1.f;
            if (a.gate) {
              float nn = a.w_b ? ((a.norm_type == 0 ? rbf(nh[i][k]) : nh[i][k]) * w8[k]) : br[k];
              if (GATED) dg_acc[i][k] += dn[k] * nn * dm;
              else atomicAdd(a.dgate + (long)b * a.mod_stride + c + k, dn[k] * nn * dm);
              dn[k] *= g8[k];
            }
            dn[k] *= dm;
          }
        }
        if (a.w_b) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float nr = (a.norm_type == 0) ? rbf(nh[i][k]) : nh[i][k];
            dw_acc[i][k] += dn[k] * nr;
            g[i][k] = dn[k] * w8[k];
            s_g += g[i][k];
            s_gx += g[i][k] * nh[i][k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) g[i][k] = dn[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { nh[i][k] = 0.f; g[i][k] = 0.f; }
      }
    }
    if (a.w_b) {
      s_gx = wave_sum(s_gx) / a.d;
      s_g = a.norm_type ? wave_sum(s_g) / a.d : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c >= a.d) continue;
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = a.w_b ? rs * (g[i][k] - s_g - nh[i][k] * s_gx) : g[i][k];
      store8_bf16(a.dbranch + row * a.d + c, o);
    }
  }
  if (!a.w_b && !(GATED && a.gate)) return;
  __shared__ float red[ROWS_PER_BLOCK][64 * 8 + 8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (a.w_b)
      block_colsum512(red, dw_acc[i], wave, lane, [&](int t, float v) {
        const int c = i * 512 + t;
        if (c < a.d) {
          if (a.ws) a.ws[(long)blockIdx.x * a.d + c] = v;   // two-phase column reduction (colreduce_kernel finishes it)
          else atomicAdd(a.dw_b + c, v);
        }
      });
    if (GATED && a.gate) {
      float* dg = a.dgate + (long)(blockIdx.x / a.bpb) * a.mod_stride;
      block_colsum512(red, dg_acc[i], wave, lane, [&](int t, float v) { if (i * 512 + t < a.d && v != 0.f) atomicAdd(dg + i * 512 + t, v); });
    }
  }
}

// ---------------------------------------------------------------------------------------------
// QK LayerNorm (over the full hidden dim, affine, eps 1e-5) + NeoX rotary on q and k.
// Lane item t -> (part, head, j): columns c_lo = part*d + head*D + 4j .. +3 and c_hi = c_lo + D/2 (the
// rotation partners live in the same lane, so no cross-lane traffic).  NIT items per lane.
// ---------------------------------------------------------------------------------------------
struct QkArgs {
  const bf16_t* qkv;   // [M, 3d]
  bf16_t* qkr;         // [M, 2d] normalised + rotated q | k
  const float* gq; const float* bq; const float* gk; const float* bk;  // [d] each, nullable (qk_norm off)
  float* stats;        // [M, 4] = mean_q, rstd_q, mean_k, rstd_k
  const float* cos_t;  // [L, D/2] or [B*L, D/2]
  const float* sin_t;
  int M, d, L, D, rope_per_sample;
  float eps;
  float q_scale = 1.f;        // the rotated q is stored as bf16(q * q_scale) - ONE rounding, where the reference rounds q and flash-attn scales the fp32 scores: the
                              // attention kernels then find log2(e) / sqrt(D) already folded into their operand (udm_attention_fwd flag UDM_ATTN_Q_PRESCALED)
};

__device__ __forceinline__ void load4_bf16(const bf16_t* p, float (&v)[4]) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ void store4_bf16(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}
__device__ __forceinline__ void load4_f32(const float* p, float (&v)[4]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}

// Lane item t -> (part, head, j): 8 columns at c_lo = part*d + head*D + 8j and 8 at c_hi = c_lo + D/2 (16-byte accesses; the
// rotation partners live in the same lane).  The LayerNorm affine vectors are staged once per block in LDS.
template <int NIT>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(QkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float aff[];  // gq | bq | gk | bk, d floats each (only with qk-norm)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = a.D / 2, per_head = a.D / 16, per_part = a.d / 16;
  const bool do_norm = a.gq != nullptr;
  if (do_norm) {
    for (int c = threadIdx.x * 4; c < a.d; c += 1024) {
      *reinterpret_cast<float4*>(aff + c) = *reinterpret_cast<const float4*>(a.gq + c);
      *reinterpret_cast<float4*>(aff + a.d + c) = *reinterpret_cast<const float4*>(a.bq + c);
      *reinterpret_cast<float4*>(aff + 2 * a.d + c) = *reinterpret_cast<const float4*>(a.gk + c);
      *reinterpret_cast<float4*>(aff + 3 * a.d + c) = *reinterpret_cast<const float4*>(a.bk + c);
    }
    __syncthreads();
  }
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < a.M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    float lo[NIT][8], hi[NIT][8];
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int t = i * 64 + lane;
      if (t < 2 * per_part) {
        const int part = t / per_part, r = t % per_part;
        const int c = part * a.d + (r / per_head) * a.D + (r % per_head) * 8;
        load8_bf16(a.qkv + row * 3 * a.d + c, lo[i]);
        load8_bf16(a.qkv + row * 3 * a.d + c + half, hi[i]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lo[i][k] + hi[i][k];
        if (part == 0) sq += s; else sk += s;
      }
    }
    float mq = 0.f, mk = 0.f, rq = 1.f, rk = 1.f;
    if (do_norm) {
      mq = wave_sum(sq) / a.d;
      mk = wave_sum(sk) / a.d;
      float vq = 0.f, vk = 0.f;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int t = i * 64 + lane;
        if (t < 2 * per_part) {
          const int part = t / per_part;
          const float m = part ? mk : mq;
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) { float u = lo[i][k] - m, w = hi[i][k] - m; s += u * u + w * w; }
          if (part == 0) vq += s; else vk += s;
        }
      }
      rq = rsqrtf(wave_sum(vq) / a.d + a.eps);
      rk = rsqrtf(wave_sum(vk) / a.d + a.eps);
      if (lane == 0) *reinterpret_cast<float4*>(a.stats + row * 4) = make_float4(mq, rq, mk, rk);
    }
    const long trow = a.rope_per_sample ? row : (row % a.L);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int t = i * 64 + lane;
      if (t >= 2 * per_part) continue;
      const int part = t / per_part, r = t % per_part;
      const int hc = (r / per_head) * a.D + (r % per_head) * 8;  // column within the part
      float xl[8], xh[8];
      if (do_norm) {
        const float* g = aff + part * 2 * a.d;
        const float* bb = g + a.d;
        const float m = part ? mk : mq, rs = part ? rk : rq;
        float g0[8], g1[8], b0[8], b1[8];
        load8_f32(g + hc, g0); load8_f32(g + hc + half, g1); load8_f32(bb + hc, b0); load8_f32(bb + hc + half, b1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xl[k] = rbf((lo[i][k] - m) * rs * g0[k] + b0[k]);  // LayerNorm result is written back in bf16 (dit.py:681-682)
          xh[k] = rbf((hi[i][k] - m) * rs * g1[k] + b1[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { xl[k] = lo[i][k]; xh[k] = hi[i][k]; }
      }
      float cs[8], sn[8], ol[8], oh[8];
      const float qs = part ? 1.f : a.q_scale;
      const int pc = (r % per_head) * 8;
      load8_f32(a.cos_t + trow * half + pc, cs);
      load8_f32(a.sin_t + trow * half + pc, sn);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ol[k] = (xl[k] * cs[k] - xh[k] * sn[k]) * qs;
        oh[k] = (xh[k] * cs[k] + xl[k] * sn[k]) * qs;
      }
      store8_bf16(a.qkr + row * 2 * a.d + part * a.d + hc, ol);
      store8_bf16(a.qkr + row * 2 * a.d + part * a.d + hc + half, oh);
    }
  }
}

struct QkBwdArgs {
  const bf16_t* dqkr;  // [M, 2d] grads wrt rotated q|k
  const bf16_t* qkv;   // [M, 3d] saved raw projections
  bf16_t* dqkv;        // [M, 3d]: columns [0,2d) written here (v columns are written by the attention backward)
  const float* gq; const float* gk;
  const float* stats;
  const float* cos_t; const float* sin_t;
  float* dgq; float* dbq; float* dgk; float* dbk;  // [d] atomics
  float* ws;           // optional [gridDim.x, 4, d] partial-sum workspace (block-per-row form)
  int M, d, L, D, rope_per_sample;
  float q_scale = 1.f;  // the incoming dq is the gradient wrt q * q_scale (see QkArgs)
};

template <int NIT>
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(QkBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float aff[];  // gq | gk (d floats each), then the reduction scratch
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = a.D / 2, per_head = a.D / 16, per_part = a.d / 16;
  const bool do_norm = a.gq != nullptr;
  if (do_norm) {
    for (int c = threadIdx.x * 4; c < a.d; c += 1024) {
      *reinterpret_cast<float4*>(aff + c) = *reinterpret_cast<const float4*>(a.gq + c);
      *reinterpret_cast<float4*>(aff + a.d + c) = *reinterpret_cast<const float4*>(a.gk + c);
    }
    __syncthreads();
  }
  float dg_acc[NIT][16], db_acc[NIT][16];
#pragma unroll
  for (int i = 0; i < NIT; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) { dg_acc[i][k] = 0.f; db_acc[i][k] = 0.f; }
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < a.M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    const long trow = a.rope_per_sample ? row : (row % a.L);
    float mq = 0.f, rq = 1.f, mk = 0.f, rk = 1.f;
    if (do_norm) {
      const float4 st = *reinterpret_cast<const float4*>(a.stats + row * 4);
      mq = st.x; rq = st.y; mk = st.z; rk = st.w;
    }
    float gl[NIT][8], gh[NIT][8], xl[NIT][8], xh[NIT][8];
    float sgq = 0.f, sgxq = 0.f, sgk = 0.f, sgxk = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int t = i * 64 + lane;
      if (t < 2 * per_part) {
        const int part = t / per_part, r = t % per_part;
        const int hc = (r / per_head) * a.D + (r % per_head) * 8;
        const int pc = (r % per_head) * 8;
        float dl[8], dh[8], cs[8], sn[8];
        load8_bf16(a.dqkr + row * 2 * a.d + part * a.d + hc, dl);
        load8_bf16(a.dqkr + row * 2 * a.d + part * a.d + hc + half, dh);
        if (part == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) { dl[k] *= a.q_scale; dh[k] *= a.q_scale; }
        }
        load8_f32(a.cos_t + trow * half + pc, cs);
        load8_f32(a.sin_t + trow * half + pc, sn);
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // transpose of the rotation
          gl[i][k] = dl[k] * cs[k] + dh[k] * sn[k];
          gh[i][k] = dh[k] * cs[k] - dl[k] * sn[k];
        }
        if (do_norm) {
          const float* g = aff + part * a.d;
          const float m = part ? mk : mq, rs = part ? rk : rq;
          float g0[8], g1[8], r0[8], r1[8];
          load8_f32(g + hc, g0); load8_f32(g + hc + half, g1);
          load8_bf16(a.qkv + row * 3 * a.d + part * a.d + hc, r0);
          load8_bf16(a.qkv + row * 3 * a.d + part * a.d + hc + half, r1);
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            xl[i][k] = (r0[k] - m) * rs;
            xh[i][k] = (r1[k] - m) * rs;
            dg_acc[i][k] += gl[i][k] * xl[i][k];
            dg_acc[i][k + 8] += gh[i][k] * xh[i][k];
            db_acc[i][k] += gl[i][k];
            db_acc[i][k + 8] += gh[i][k];
            gl[i][k] *= g0[k];
            gh[i][k] *= g1[k];
            s1 += gl[i][k] + gh[i][k];
            s2 += gl[i][k] * xl[i][k] + gh[i][k] * xh[i][k];
          }
          if (part == 0) { sgq += s1; sgxq += s2; } else { sgk += s1; sgxk += s2; }
        }
      }
    }
    if (do_norm) {
      sgq = wave_sum(sgq) / a.d; sgxq = wave_sum(sgxq) / a.d;
      sgk = wave_sum(sgk) / a.d; sgxk = wave_sum(sgxk) / a.d;
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int t = i * 64 + lane;
      if (t >= 2 * per_part) continue;
      const int part = t / per_part, r = t % per_part;
      const int hc = (r / per_head) * a.D + (r % per_head) * 8;
      float ol[8], oh[8];
      if (do_norm) {
        const float rs = part ? rk : rq, sg = part ? sgk : sgq, sgx = part ? sgxk : sgxq;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          ol[k] = rs * (gl[i][k] - sg - xl[i][k] * sgx);
          oh[k] = rs * (gh[i][k] - sg - xh[i][k] * sgx);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ol[k] = gl[i][k]; oh[k] = gh[i][k]; }
      }
      store8_bf16(a.dqkv + row * 3 * a.d + part * a.d + hc, ol);
      store8_bf16(a.dqkv + row * 3 * a.d + part * a.d + hc + half, oh);
    }
  }
  if (!do_norm) return;
  // reduce dgamma/dbeta over the block's 4 waves via LDS (after the affine copy), then one atomic per column per block
  float* red = aff + 2 * a.d;  // [4 waves][64 lanes * 16]
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; ++k) red[wave * 1024 + lane * 16 + k] = pass ? db_acc[i][k] : dg_acc[i][k];
      __syncthreads();
      for (int u = threadIdx.x; u < 1024; u += 256) {
        const int ln = u >> 4, k = u & 15;
        const int t = i * 64 + ln;
        if (t < 2 * per_part) {
          const int part = t / per_part, r = t % per_part;
          const int hc = (r / per_head) * a.D + (r % per_head) * 8 + (k & 7) + (k >> 3) * half;
          const float s = red[u] + red[1024 + u] + red[2048 + u] + red[3072 + u];
          if (a.ws) {   // two-phase column reduction, workspace row layout [dgq | dbq | dgk | dbk] (colreduce_kernel finishes it)
            a.ws[(long)blockIdx.x * 4 * a.d + part * 2 * a.d + pass * a.d + hc] = s;
          } else {
            float* dst = pass ? (part ? a.dbk : a.dbq) : (part ? a.dgk : a.dgq);
            atomicAdd(dst + hc, s);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// token embedding gather (+ modality embedding) and its scatter-add backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ E, const int64_t* __restrict__ modality,
                                                           const float* __restrict__ Em, float* __restrict__ x, long M, int d, long V) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    long id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const float* e = E + id * d;
    const float* em = Em ? Em + (modality[row] == 0 ? 0 : d) : nullptr;
    for (int c = lane * 4; c < d; c += 256) {
      float4 v = *reinterpret_cast<const float4*>(e + c);
      if (em) {
        float4 m = *reinterpret_cast<const float4*>(em + c);
        v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
      }
      *reinterpret_cast<float4*>(x + row * d + c) = v;
    }
  }
}

// A block owns a chunk of `rows_per_block` rows x 1024 columns (grid.y column chunks); a thread owns 4 consecutive columns (16-byte loads, eight rows requested
// before the first is used; the kernel is bound by its same-address atomics - the [MASK] row and the two modality rows - not by the reads).  Rows whose id is
// `hot_id` (the [MASK] token: about half of all rows under the absorbing schedule) are summed in registers and leave the block as ONE atomic per column; other
// rows scatter with atomics directly.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ modality, const float* __restrict__ dx,
                                                           float* __restrict__ dE, float* __restrict__ dEm, long M, int d, long V, long hot_id,
                                                           int rows_per_block) {
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  const int c = blockIdx.y * 1024 + threadIdx.x * 4;
  if (c >= d) return;
  float hot[4] = {0.f, 0.f, 0.f, 0.f}, m0[4] = {0.f, 0.f, 0.f, 0.f}, m1[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int RU = 8;
  for (long rb = r0; rb < r1; rb += RU) {
    float4 g[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (rb + u < r1) g[u] = *reinterpret_cast<const float4*>(dx + (rb + u) * d + c);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (rb + u >= r1) break;
      const long id = ids[rb + u];
      const float gv[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
      if (id == hot_id) {
#pragma unroll
        for (int e = 0; e < 4; ++e) hot[e] += gv[e];
      } else if (id >= 0 && id < V) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dE + id * d + c + e, gv[e]);
      }
      if (dEm) {
        const bool txt = modality[rb + u] == 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { m0[e] += txt ? gv[e] : 0.f; m1[e] += txt ? 0.f : gv[e]; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (hot[e] != 0.f) atomicAdd(dE + hot_id * d + c + e, hot[e]);
    if (dEm) { atomicAdd(dEm + c + e, m0[e]); atomicAdd(dEm + d + c + e, m1[e]); }
  }
}

// ---------------------------------------------------------------------------------------------
// small adaLN helpers
// ---------------------------------------------------------------------------------------------
// sinusoidal timestep features (reference TimestepEmbedder.timestep_embedding, dit.py:428-444) -> bf16 [B, dim]
__global__ void timestep_embedding_kernel(const float* __restrict__ sigma, bf16_t* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * dim) return;
  const int b = i / dim, j = i % dim;
  float v = 0.f;
  if (j < 2 * half) {
    const int jj = j < half ? j : j - half;
    const float f = expf(-logf(10000.f) * (float)jj / (float)half);
    const float arg = sigma[b] * f;
    v = j < half ? cosf(arg) : sinf(arg);
  }
  out[i] = f2bf(v);
}

__global__ void silu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = bf2f(x[i]);
  y[i] = f2bf(v / (1.f + __expf(-v)));
}
__global__ void silu_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = bf2f(x[i]);
  const float s = 1.f / (1.f + __expf(-v));
  dx[i] = f2bf(bf2f(dy[i]) * (s + v * s * (1.f - s)));
}
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<const float4*>(x + i);
    *reinterpret_cast<uint2*>(y + i) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
  } else {
    for (long k = i; k < n; ++k) y[k] = f2bf(x[k]);
  }
}
// bf16 -> fp32 with scale; used by the gradient-bucket decompress (reference DDP bf16 compress hook, main.py:645)
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n, float scale) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    uint2 u = *reinterpret_cast<const uint2*>(x + i);
    *reinterpret_cast<float4*>(y + i) = make_float4(__uint_as_float(u.x << 16) * scale, __uint_as_float(u.x & 0xffff0000u) * scale,
                                                   __uint_as_float(u.y << 16) * scale, __uint_as_float(u.y & 0xffff0000u) * scale);
  } else {
    for (long k = i; k < n; ++k) y[k] = bf2f(x[k]) * scale;
  }
}
__global__ void scale_cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n, float scale) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<const float4*>(x + i);
    // reference order: cast to bf16 first, then divide by world size in bf16 (torch _compress_hook)
    *reinterpret_cast<uint2*>(y + i) = make_uint2(pack2bf(rbf(v.x) * scale, rbf(v.y) * scale), pack2bf(rbf(v.z) * scale, rbf(v.w) * scale));
  } else {
    for (long k = i; k < n; ++k) y[k] = f2bf(rbf(x[k]) * scale);
  }
}

// ---------------------------------------------------------------------------------------------
// Block-per-row variants for wide rows (d >= 1024): the 4 waves of a block share ONE row, so a thread holds 8*NCB
// elements instead of 32+ and 6-8 waves fit per SIMD (the wave-per-row forms of these three kernels sit at 1-2
// waves per SIMD and reach only 30-50 % of HBM bandwidth).  Row statistics are combined across the 4 waves in LDS.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void block_sum4(float (&v)[N], float* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) sm[wave * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = sm[k] + sm[N + k] + sm[2 * N + k] + sm[3 * N + k];
  __syncthreads();
}

template <int NCB>  // chunks of 8 columns per thread: column = (i * 256 + tid) * 8
__global__ __launch_bounds__(256) void residual_bwd_brow_kernel(ResidBwdArgs a) {
  __shared__ float sm[8];
  const int tid = threadIdx.x;
  float dw_acc[NCB][8];
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) dw_acc[i][k] = 0.f;
  const float keep_scale = 1.f / (1.f - a.p_drop);
  // The next row's operands are requested before the current row is reduced: one row per block iteration with two block-wide
  // reductions in it otherwise exposes the full HBM latency per row (measured 3.3 TB/s at d = 2048).
  float br_n[NCB][8], dn_n[NCB][8];
  auto fetch = [&](long row) {
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = (i * 256 + tid) * 8;
      if (c < a.d && row < a.M) {
        load8_bf16(a.branch + row * a.d + c, br_n[i]);
        load8_f32(a.dx + row * a.d + c, dn_n[i]);
      }
    }
  };
  fetch(blockIdx.x);
  for (long row = blockIdx.x; row < a.M; row += gridDim.x) {
    const int b = (int)(row / a.L);
    const bool special = !a.modality || a.modality[row] == 1;
    const float rs = a.w_b ? a.rstd_b[row] : 1.f;
    const float mu = (a.w_b && a.norm_type) ? a.mean_b[row] : 0.f;
    float nh[NCB][8], g[NCB][8];
    float red[2] = {0.f, 0.f};
    float br_c[NCB][8], dn_c[NCB][8];
#pragma unroll
    for (int i = 0; i < NCB; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) { br_c[i][k] = br_n[i][k]; dn_c[i][k] = dn_n[i][k]; }
    fetch(row + gridDim.x);
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = (i * 256 + tid) * 8;
      if (c < a.d) {
        float br[8], dn[8], w8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { br[k] = br_c[i][k]; dn[k] = dn_c[i][k]; }
        if (a.w_b) load8_f32(a.w_b + c, w8);
#pragma unroll
        for (int k = 0; k < 8; ++k) nh[i][k] = (br[k] - mu) * rs;
        if (special) {
          float g8[8];
          if (a.gate) load8_bf16(a.gate + (long)b * a.mod_stride + c, g8);
          bool keep[8];
          if (a.p_drop > 0.f) dropout_keep8(a.seed, (uint64_t)row * a.d + c, a.p_drop, keep);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float dm = (a.p_drop > 0.f) ? (keep[k] ? keep_scale : 0.f) : 1.f;
            if (a.gate) {
              float nn = a.w_b ? ((a.norm_type == 0 ? rbf(nh[i][k]) : nh[i][k]) * w8[k]) : br[k];
              atomicAdd(a.dgate + (long)b * a.mod_stride + c + k, dn[k] * nn * dm);
              dn[k] *= g8[k];
            }
            dn[k] *= dm;
          }
        }
        if (a.w_b) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float nr = (a.norm_type == 0) ? rbf(nh[i][k]) : nh[i][k];
            dw_acc[i][k] += dn[k] * nr;
            g[i][k] = dn[k] * w8[k];
            red[0] += g[i][k];
            red[1] += g[i][k] * nh[i][k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) g[i][k] = dn[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { nh[i][k] = 0.f; g[i][k] = 0.f; }
      }
    }
    float s_g = 0.f, s_gx = 0.f;
    if (a.w_b) {
      block_sum4<2>(red, sm);
      s_gx = red[1] / a.d;
      s_g = a.norm_type ? red[0] / a.d : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = (i * 256 + tid) * 8;
      if (c >= a.d) continue;
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = a.w_b ? rs * (g[i][k] - s_g - nh[i][k] * s_gx) : g[i][k];
      store8_bf16(a.dbranch + row * a.d + c, o);
    }
  }
  if (!a.w_b) return;
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int c = (i * 256 + tid) * 8;
    if (c < a.d) {
      if (a.ws) {
        store8_f32(a.ws + (long)blockIdx.x * a.d + c, dw_acc[i]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(a.dw_b + c + k, dw_acc[i][k]);
      }
    }
  }
}

// out[c] += sum_r ws[r, c]: finishes the two-phase column reductions of the block-per-row backward kernels.  A 1024-deep chain of
// fp32 atomics on one address costs ~190 us on MI355X (cross-XCD atomics serialise at the memory side); here the depth is gridDim.y.
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int nrows, int ncols) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int per = (nrows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  float s = 0.f;
  if (col < ncols) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four loads in flight per thread (the loop is latency-bound otherwise)
    int r = r0 + sub;
    for (; r + 12 < r1; r += 16) {
      s0 += ws[(long)r * ncols + col];
      s1 += ws[(long)(r + 4) * ncols + col];
      s2 += ws[(long)(r + 8) * ncols + col];
      s3 += ws[(long)(r + 12) * ncols + col];
    }
    for (; r < r1; r += 4) s0 += ws[(long)r * ncols + col];
    s = (s0 + s1) + (s2 + s3);
  }
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && col < ncols) atomicAdd(out + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// Fused backward of  h = norm(x; w)  FOLLOWED BY the backward of the residual branch that produced x:
//     dx   (+)= d norm / dx (dy)                                (udm_norm_bwd, unmodulated)
//     dbr   =  d [x = x_in + dropout(sandwich_norm(branch; w_b))] / d branch  evaluated at the UPDATED dx   (udm_residual_bwd, no gate)
// The two kernels always run back to back in the block's backward (norm2 -> attention branch, norm1 -> the previous block's MLP branch,
// final norm -> the last block's MLP branch) and the second re-reads the fp32 dx the first has just written: fused, a row moves
// 12 B read + 6 B written per element instead of 14 + 8.  Block per row (d = 2048 / 4096): a thread holds 8 NCB columns of every operand,
// the next row's four operands are requested before the current row's two block-wide reductions.
// Column sums (dw of the norm, dw_b of the sandwich norm) go through ONE workspace [gridDim.x][2][d] and one reduction launch.
// ---------------------------------------------------------------------------------------------
struct NormResidBwdArgs {
  const bf16_t* dy; const float* x; const float* rstd; const float* mean; const float* w;   // the norm whose input is x
  float* dx; float* dw; int accumulate;
  const bf16_t* branch; bf16_t* dbranch; const float* w_b; const float* rstd_b; const float* mean_b; float* dw_b;   // the residual branch
  float* dbias;   // optional [d]: += column sums of the (bf16-rounded) d branch = bias gradient of the Linear that produced the branch
  float* ws;
  int M, d, norm_type;
  float p_drop;
  uint64_t seed;
  // ADA form (adaLN-Zero, models/dit.py:263-304 modulate_fused + :229-253 bias_dropout_add_scale): the norm is modulated by shift / scale [B, mod_stride] (rows with
  // modality == 1 only when a modality map is given and *any_img != 0), the residual branch optionally gated (gate [B, mod_stride]; gate and dropout on rows with
  // modality_r == 1 only when that map is given).  Blocks own runs of rows of ONE batch element (grid = B * bpb).
  const bf16_t* shift = nullptr; const bf16_t* scale = nullptr; float* dshift = nullptr; float* dscale = nullptr;
  const bf16_t* gate = nullptr; float* dgate = nullptr;
  long mod_stride = 0;
  const int64_t* modality = nullptr; const int* any_img = nullptr; const int64_t* modality_r = nullptr;
  int L = 0, bpb = 0;
};

template <int NCB, bool ADA = false>
__global__ __launch_bounds__(256) void norm_residual_bwd_kernel(NormResidBwdArgs a) {
  __shared__ float sm[8];
  const int tid = threadIdx.x;
  float dwn[NCB][8], dwb[NCB][8], dbs[NCB][8];
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) { dwn[i][k] = 0.f; dwb[i][k] = 0.f; dbs[i][k] = 0.f; }
  // ADA: this block's batch element, its run of rows, that element's scale / gate vectors and the three column sums a thread owns outright (block per row: a
  // thread keeps the same 8 columns for every row - no cross-thread reduction, one atomic per column and block at the end)
  long row_first = blockIdx.x, row_end = a.M, row_step = gridDim.x;
  float sc8[ADA ? NCB : 1][8], gt8[ADA ? NCB : 1][8], dsh[ADA ? NCB : 1][8], dsc[ADA ? NCB : 1][8], dgt[ADA ? NCB : 1][8];
  bool img_only = false;
  int ada_b = 0;
  if (ADA) {
    ada_b = blockIdx.x / a.bpb;
    const int chunk = (a.L + a.bpb - 1) / a.bpb;
    row_first = (long)ada_b * a.L + (long)(blockIdx.x % a.bpb) * chunk;
    row_end = min((long)(ada_b + 1) * a.L, row_first + chunk);
    row_step = 1;
    img_only = a.modality && (!a.any_img || *a.any_img != 0);
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = (i * 256 + tid) * 8;
      if (a.scale) load8_bf16(a.scale + (long)ada_b * a.mod_stride + c, sc8[i]);
      if (a.gate) load8_bf16(a.gate + (long)ada_b * a.mod_stride + c, gt8[i]);
#pragma unroll
      for (int k = 0; k < 8; ++k) { dsh[i][k] = 0.f; dsc[i][k] = 0.f; dgt[i][k] = 0.f; }
    }
  }
  const float keep_scale = 1.f / (1.f - a.p_drop);
  float dy_n[NCB][8], x_n[NCB][8], dx_n[NCB][8], br_n[NCB][8];
  auto fetch = [&](long row) {
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = (i * 256 + tid) * 8;
      if (row < a.M) {
        load8_bf16(a.dy + row * a.d + c, dy_n[i]);
        load8_f32(a.x + row * a.d + c, x_n[i]);
        if (a.accumulate) load8_f32(a.dx + row * a.d + c, dx_n[i]);
        load8_bf16(a.branch + row * a.d + c, br_n[i]);
      }
    }
  };
  auto fetch_in_range = [&](long row) { fetch(row < row_end ? row : (long)a.M); };   // (past the block's run: nothing is requested)
  fetch_in_range(row_first);
  float w8[NCB][8], wb8[NCB][8];
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int c = (i * 256 + tid) * 8;
    load8_f32(a.w + c, w8[i]);
    if (a.w_b) load8_f32(a.w_b + c, wb8[i]);
  }
  for (long row = row_first; row < row_end; row += row_step) {
    const bool modulate = ADA && a.shift && (!img_only || a.modality[row] == 1);
    const bool special = !ADA || !a.modality_r || a.modality_r[row] == 1;     // the row receives gate + dropout
    const float rs = a.rstd[row];
    const float mu = a.norm_type ? a.mean[row] : 0.f;
    const float rsb = a.w_b ? a.rstd_b[row] : 1.f;
    const float mub = (a.w_b && a.norm_type) ? a.mean_b[row] : 0.f;
    float xh[NCB][8], g[NCB][8], dxo[NCB][8], br[NCB][8];
    float red[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NCB; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        xh[i][k] = (x_n[i][k] - mu) * rs;
        float dyv = dy_n[i][k];
        if (ADA && modulate) {
          dsh[i][k] += dyv;
          dsc[i][k] += dyv * xh[i][k] * w8[i][k];
          dyv *= 1.f + sc8[i][k];
        }
        dwn[i][k] += dyv * xh[i][k];
        g[i][k] = dyv * w8[i][k];
        red[0] += g[i][k];
        red[1] += g[i][k] * xh[i][k];
        dxo[i][k] = a.accumulate ? dx_n[i][k] : 0.f;
        br[i][k] = br_n[i][k];
      }
    fetch_in_range(row + row_step);
    block_sum4<2>(red, sm);
    const float s_gx = red[1] / a.d, s_g = a.norm_type ? red[0] / a.d : 0.f;
    float nh[NCB][8], g2[NCB][8];
    float red2[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = (i * 256 + tid) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) dxo[i][k] += rs * (g[i][k] - s_g - xh[i][k] * s_gx);
      store8_f32(a.dx + row * a.d + c, dxo[i]);
      float dn[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) dn[k] = dxo[i][k];
      if (a.w_b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) nh[i][k] = (br[i][k] - mub) * rsb;
      }
      if (special) {
        float dm8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dm8[k] = 1.f;
        if (a.p_drop > 0.f) {
          bool keep[8];
          dropout_keep8(a.seed, (uint64_t)row * a.d + c, a.p_drop, keep);
#pragma unroll
          for (int k = 0; k < 8; ++k) dm8[k] = keep[k] ? keep_scale : 0.f;
        }
        if (ADA && a.gate) {   // the gate's gradient sees the branch as the forward gated it: (sandwich-)normalised, after dropout
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float nn = a.w_b ? ((a.norm_type == 0 ? rbf(nh[i][k]) : nh[i][k]) * wb8[i][k]) : br[i][k];
            dgt[i][k] += dn[k] * nn * dm8[k];
            dn[k] *= gt8[i][k];
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) dn[k] *= dm8[k];
      }
      if (a.w_b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float nr = (a.norm_type == 0) ? rbf(nh[i][k]) : nh[i][k];
          dwb[i][k] += dn[k] * nr;
          g2[i][k] = dn[k] * wb8[i][k];
          red2[0] += g2[i][k];
          red2[1] += g2[i][k] * nh[i][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) g2[i][k] = dn[k];
      }
    }
    float s2_gx = 0.f, s2_g = 0.f;
    if (a.w_b) {
      block_sum4<2>(red2, sm);
      s2_gx = red2[1] / a.d;
      s2_g = a.norm_type ? red2[0] / a.d : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = (i * 256 + tid) * 8;
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = a.w_b ? rsb * (g2[i][k] - s2_g - nh[i][k] * s2_gx) : g2[i][k];
      store8_bf16(a.dbranch + row * a.d + c, o);
      if (a.dbias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) dbs[i][k] += rbf(o[k]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int c = (i * 256 + tid) * 8;
    store8_f32(a.ws + ((long)blockIdx.x * 3) * a.d + c, dwn[i]);
    if (a.w_b) store8_f32(a.ws + ((long)blockIdx.x * 3 + 1) * a.d + c, dwb[i]);
    if (a.dbias) store8_f32(a.ws + ((long)blockIdx.x * 3 + 2) * a.d + c, dbs[i]);
    if (ADA) {   // second workspace region [grid][3][d]: the block's shift / scale / gate column sums; ada_reduce_kernel adds each batch element's blocks into dmod
      float* w2 = a.ws + (long)gridDim.x * 3 * a.d;   // (one atomic per column and block here - 4.7 M atomics in 96-deep chains - cost as much as the pass itself)
      if (a.shift) {
        store8_f32(w2 + ((long)blockIdx.x * 3) * a.d + c, dsh[i]);
        store8_f32(w2 + ((long)blockIdx.x * 3 + 1) * a.d + c, dsc[i]);
      }
      if (a.gate) store8_f32(w2 + ((long)blockIdx.x * 3 + 2) * a.d + c, dgt[i]);
    }
  }
}

// out_k[b][c] += sum over the bpb blocks of batch element b of ws2[block][k][c]  (k = 0 shift, 1 scale, 2 gate; a null target is skipped).  grid (ceil(d / 256), B, 3)
__global__ __launch_bounds__(256) void ada_reduce_kernel(const float* __restrict__ ws2, float* dshift, float* dscale, float* dgate, long mod_stride, int bpb, int d) {
  const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, k = blockIdx.z;
  float* out = k == 0 ? dshift : (k == 1 ? dscale : dgate);
  if (!out || c >= d) return;
  const float* p = ws2 + ((long)b * bpb * 3 + k) * d + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int j = 0;
  for (; j + 3 < bpb; j += 4) {
    s0 += p[(long)j * 3 * d]; s1 += p[(long)(j + 1) * 3 * d]; s2 += p[(long)(j + 2) * 3 * d]; s3 += p[(long)(j + 3) * 3 * d];
  }
  for (; j < bpb; ++j) s0 += p[(long)j * 3 * d];
  out[(long)b * mod_stride + c] += (s0 + s1) + (s2 + s3);
}

// The same fused pass for narrow rows (d <= 2048, e.g. UniDisc-S d = 768): a WAVE per row (lane owns 8 columns of each 512-column chunk), no block-wide
// barrier in the row loop; the four waves' column sums meet in LDS at the end and leave one [3][d] workspace row per block.
template <int NCH>
__global__ __launch_bounds__(256) void norm_residual_bwd_wrow_kernel(NormResidBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dwn[NCH][8], dwb[NCH][8], dbs[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) { dwn[i][k] = 0.f; dwb[i][k] = 0.f; dbs[i][k] = 0.f; }
  const float keep_scale = 1.f / (1.f - a.p_drop);
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < a.M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    const float rs = a.rstd[row];
    const float mu = a.norm_type ? a.mean[row] : 0.f;
    const float rsb = a.w_b ? a.rstd_b[row] : 1.f;
    const float mub = (a.w_b && a.norm_type) ? a.mean_b[row] : 0.f;
    float xh[NCH][8], g[NCH][8], dxo[NCH][8], br[NCH][8];
    float s_g = 0.f, s_gx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c < a.d) {
        float x8[8], dy8[8], w8[8];
        load8_f32(a.x + row * a.d + c, x8);
        load8_bf16(a.dy + row * a.d + c, dy8);
        load8_bf16(a.branch + row * a.d + c, br[i]);
        if (a.accumulate) load8_f32(a.dx + row * a.d + c, dxo[i]);
        load8_f32(a.w + c, w8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xh[i][k] = (x8[k] - mu) * rs;
          dwn[i][k] += dy8[k] * xh[i][k];
          g[i][k] = dy8[k] * w8[k];
          s_g += g[i][k];
          s_gx += g[i][k] * xh[i][k];
          if (!a.accumulate) dxo[i][k] = 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { xh[i][k] = 0.f; g[i][k] = 0.f; dxo[i][k] = 0.f; br[i][k] = 0.f; }
      }
    }
    s_gx = wave_sum(s_gx) / a.d;
    s_g = a.norm_type ? wave_sum(s_g) / a.d : 0.f;
    float nh[NCH][8], g2[NCH][8];
    float r2_g = 0.f, r2_gx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c >= a.d) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { nh[i][k] = 0.f; g2[i][k] = 0.f; }
        continue;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) dxo[i][k] += rs * (g[i][k] - s_g - xh[i][k] * s_gx);
      store8_f32(a.dx + row * a.d + c, dxo[i]);
      float dn[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) dn[k] = dxo[i][k];
      if (a.p_drop > 0.f) {
        bool keep[8];
        dropout_keep8(a.seed, (uint64_t)row * a.d + c, a.p_drop, keep);
#pragma unroll
        for (int k = 0; k < 8; ++k) dn[k] *= keep[k] ? keep_scale : 0.f;
      }
      if (a.w_b) {
        float wb8[8];
        load8_f32(a.w_b + c, wb8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          nh[i][k] = (br[i][k] - mub) * rsb;
          const float nr = (a.norm_type == 0) ? rbf(nh[i][k]) : nh[i][k];
          dwb[i][k] += dn[k] * nr;
          g2[i][k] = dn[k] * wb8[k];
          r2_g += g2[i][k];
          r2_gx += g2[i][k] * nh[i][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { g2[i][k] = dn[k]; nh[i][k] = 0.f; }
      }
    }
    float s2_gx = 0.f, s2_g = 0.f;
    if (a.w_b) {
      s2_gx = wave_sum(r2_gx) / a.d;
      s2_g = a.norm_type ? wave_sum(r2_g) / a.d : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (i * 64 + lane) * 8;
      if (c >= a.d) continue;
      float o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = a.w_b ? rsb * (g2[i][k] - s2_g - nh[i][k] * s2_gx) : g2[i][k];
      store8_bf16(a.dbranch + row * a.d + c, o);
      if (a.dbias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) dbs[i][k] += rbf(o[k]);
      }
    }
  }
  // the four waves' column sums meet in LDS; one [3][d] workspace row per block (colreduce3 finishes it)
  __shared__ float red[ROWS_PER_BLOCK][64 * 8 + 8];
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    if ((which == 1 && !a.w_b) || (which == 2 && !a.dbias)) continue;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = which == 0 ? dwn[i][k] : (which == 1 ? dwb[i][k] : dbs[i][k]);
      __syncthreads();
      for (int t = threadIdx.x; t < 512; t += 256) {
        const int c = i * 512 + t;
        if (c < a.d) a.ws[((long)blockIdx.x * 3 + which) * a.d + c] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
      }
    }
  }
}

// out_j[c] += sum_r ws[r][j][c], j = blockIdx.z in 0..2  (ws is [nrows][3][d]; out1 / out2 may be null)
__global__ __launch_bounds__(256) void colreduce3_kernel(const float* __restrict__ ws, float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2,
                                                         int nrows, int d) {
  __shared__ float red[4][64];
  const int which = blockIdx.z;
  float* const outp = which == 0 ? out0 : (which == 1 ? out1 : out2);
  if (!outp) return;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int per = (nrows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  const float* base = ws + (long)which * d;
  const long rstride = 3L * d;
  float s = 0.f;
  if (col < d) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0 + sub;
    for (; r + 12 < r1; r += 16) {
      s0 += base[(long)r * rstride + col];
      s1 += base[(long)(r + 4) * rstride + col];
      s2 += base[(long)(r + 8) * rstride + col];
      s3 += base[(long)(r + 12) * rstride + col];
    }
    for (; r < r1; r += 4) s0 += base[(long)r * rstride + col];
    s = (s0 + s1) + (s2 + s3);
  }
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && col < d) atomicAdd(outp + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// item t = i * 256 + tid -> (part, head, j): 8 columns at c_lo and 8 at c_lo + D/2 (see the wave-per-row kernels above)
template <int NIB>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_brow_kernel(QkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float aff[];  // gq | bq | gk | bk
  __shared__ float sm[8];
  const int tid = threadIdx.x;
  const int half = a.D / 2, per_head = a.D / 16, per_part = a.d / 16;
  const bool do_norm = a.gq != nullptr;
  if (do_norm) {
    for (int c = tid * 4; c < a.d; c += 1024) {
      *reinterpret_cast<float4*>(aff + c) = *reinterpret_cast<const float4*>(a.gq + c);
      *reinterpret_cast<float4*>(aff + a.d + c) = *reinterpret_cast<const float4*>(a.bq + c);
      *reinterpret_cast<float4*>(aff + 2 * a.d + c) = *reinterpret_cast<const float4*>(a.gk + c);
      *reinterpret_cast<float4*>(aff + 3 * a.d + c) = *reinterpret_cast<const float4*>(a.bk + c);
    }
    __syncthreads();
  }
  for (long row = blockIdx.x; row < a.M; row += gridDim.x) {
    float lo[NIB][8], hi[NIB][8];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int t = i * 256 + tid;
      if (t < 2 * per_part) {
        const int part = t / per_part, r = t % per_part;
        const int c = part * a.d + (r / per_head) * a.D + (r % per_head) * 8;
        load8_bf16(a.qkv + row * 3 * a.d + c, lo[i]);
        load8_bf16(a.qkv + row * 3 * a.d + c + half, hi[i]);
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) u += lo[i][k] + hi[i][k];
        s[part] += u;
      }
    }
    float mq = 0.f, mk = 0.f, rq = 1.f, rk = 1.f;
    if (do_norm) {
      block_sum4<2>(s, sm);
      mq = s[0] / a.d;
      mk = s[1] / a.d;
      float v[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NIB; ++i) {
        const int t = i * 256 + tid;
        if (t < 2 * per_part) {
          const int part = t / per_part;
          const float m = part ? mk : mq;
          float u = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) { float x = lo[i][k] - m, y = hi[i][k] - m; u += x * x + y * y; }
          v[part] += u;
        }
      }
      block_sum4<2>(v, sm);
      rq = rsqrtf(v[0] / a.d + a.eps);
      rk = rsqrtf(v[1] / a.d + a.eps);
      if (tid == 0) *reinterpret_cast<float4*>(a.stats + row * 4) = make_float4(mq, rq, mk, rk);
    }
    const long trow = a.rope_per_sample ? row : (row % a.L);
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int t = i * 256 + tid;
      if (t >= 2 * per_part) continue;
      const int part = t / per_part, r = t % per_part;
      const int hc = (r / per_head) * a.D + (r % per_head) * 8;
      float xl[8], xh[8];
      if (do_norm) {
        const float* g = aff + part * 2 * a.d;
        const float* bb = g + a.d;
        const float m = part ? mk : mq, rs = part ? rk : rq;
        float g0[8], g1[8], b0[8], b1[8];
        load8_f32(g + hc, g0); load8_f32(g + hc + half, g1); load8_f32(bb + hc, b0); load8_f32(bb + hc + half, b1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xl[k] = rbf((lo[i][k] - m) * rs * g0[k] + b0[k]);
          xh[k] = rbf((hi[i][k] - m) * rs * g1[k] + b1[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { xl[k] = lo[i][k]; xh[k] = hi[i][k]; }
      }
      float cs[8], sn[8], ol[8], oh[8];
      const float qs = part ? 1.f : a.q_scale;
      const int pc = (r % per_head) * 8;
      load8_f32(a.cos_t + trow * half + pc, cs);
      load8_f32(a.sin_t + trow * half + pc, sn);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ol[k] = (xl[k] * cs[k] - xh[k] * sn[k]) * qs;
        oh[k] = (xh[k] * cs[k] + xl[k] * sn[k]) * qs;
      }
      store8_bf16(a.qkr + row * 2 * a.d + part * a.d + hc, ol);
      store8_bf16(a.qkr + row * 2 * a.d + part * a.d + hc + half, oh);
    }
  }
}

// d = 2048 (one item per thread): R rows per block iteration.  The kernel is bound by bytes in flight, not by bandwidth: with one row per iteration and
// 32 KB of LDS for the affine vectors, 5 blocks per CU hold 5 x 8 KB of loads in flight - at ~3 us of loaded memory latency that is 3.4 TB/s, what was
// measured.  Here a thread keeps the affine values of its 16 columns in registers (they are the same for every row: no LDS at all), and every iteration
// loads R rows before the first reduction, so the two block-wide reductions of a row are shared by R rows as well.
template <int R>
__global__ __launch_bounds__(256) void qknorm_rope_fwd_brow_rows_kernel(QkArgs a) {
  __shared__ float sm[4 * 2 * R];
  const int tid = threadIdx.x;
  const int half = a.D / 2, per_head = a.D / 16, per_part = a.d / 16;
  const bool do_norm = a.gq != nullptr;
  const int part = tid / per_part, r = tid % per_part;   // 2 * per_part == 256
  const int hc = (r / per_head) * a.D + (r % per_head) * 8;
  const int pc = (r % per_head) * 8;
  const int c = part * a.d + hc;
  float g0[8], g1[8], b0[8], b1[8];
  if (do_norm) {
    const float* g = part ? a.gk : a.gq;
    const float* bb = part ? a.bk : a.bq;
    load8_f32(g + hc, g0); load8_f32(g + hc + half, g1); load8_f32(bb + hc, b0); load8_f32(bb + hc + half, b1);
  }
  for (long row0 = (long)blockIdx.x * R; row0 < a.M; row0 += (long)gridDim.x * R) {
    float lo[R][8], hi[R][8];
    float s[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const long row = row0 + j < a.M ? row0 + j : a.M - 1;   // (a ragged last group recomputes the last row: same values, benign)
      load8_bf16(a.qkv + row * 3 * a.d + c, lo[j]);
      load8_bf16(a.qkv + row * 3 * a.d + c + half, hi[j]);
    }
    float mrow[R], rrow[R];
    if (do_norm) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) u += lo[j][k] + hi[j][k];
        s[2 * j + part] = u;
        s[2 * j + (part ^ 1)] = 0.f;
      }
      block_sum4<2 * R>(s, sm);
      float v[2 * R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const float m = s[2 * j + part] / a.d;
        mrow[j] = m;
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { float x = lo[j][k] - m, y = hi[j][k] - m; u += x * x + y * y; }
        v[2 * j + part] = u;
        v[2 * j + (part ^ 1)] = 0.f;
      }
      block_sum4<2 * R>(v, sm);
#pragma unroll
      for (int j = 0; j < R; ++j) {
        rrow[j] = rsqrtf(v[2 * j + part] / a.d + a.eps);
        if (tid == 0 && row0 + j < a.M)
          *reinterpret_cast<float4*>(a.stats + (row0 + j) * 4) = make_float4(s[2 * j] / a.d, rsqrtf(v[2 * j] / a.d + a.eps), s[2 * j + 1] / a.d, rsqrtf(v[2 * j + 1] / a.d + a.eps));
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (row0 + j >= a.M) break;
      const long row = row0 + j;
      const long trow = a.rope_per_sample ? row : (row % a.L);
      float xl[8], xh[8];
      if (do_norm) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xl[k] = rbf((lo[j][k] - mrow[j]) * rrow[j] * g0[k] + b0[k]);
          xh[k] = rbf((hi[j][k] - mrow[j]) * rrow[j] * g1[k] + b1[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { xl[k] = lo[j][k]; xh[k] = hi[j][k]; }
      }
      float cs[8], sn[8], ol[8], oh[8];
      const float qs = part ? 1.f : a.q_scale;
      load8_f32(a.cos_t + trow * half + pc, cs);
      load8_f32(a.sin_t + trow * half + pc, sn);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ol[k] = (xl[k] * cs[k] - xh[k] * sn[k]) * qs;
        oh[k] = (xh[k] * cs[k] + xl[k] * sn[k]) * qs;
      }
      store8_bf16(a.qkr + row * 2 * a.d + part * a.d + hc, ol);
      store8_bf16(a.qkr + row * 2 * a.d + part * a.d + hc + half, oh);
    }
  }
}

template <int NIB>
__global__ __launch_bounds__(256) void qknorm_rope_bwd_brow_kernel(QkBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float aff[];  // gq | gk
  __shared__ float sm[16];
  const int tid = threadIdx.x;
  const int half = a.D / 2, per_head = a.D / 16, per_part = a.d / 16;
  const bool do_norm = a.gq != nullptr;
  if (do_norm) {
    for (int c = tid * 4; c < a.d; c += 1024) {
      *reinterpret_cast<float4*>(aff + c) = *reinterpret_cast<const float4*>(a.gq + c);
      *reinterpret_cast<float4*>(aff + a.d + c) = *reinterpret_cast<const float4*>(a.gk + c);
    }
    __syncthreads();
  }
  float dg_acc[NIB][16], db_acc[NIB][16];
#pragma unroll
  for (int i = 0; i < NIB; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) { dg_acc[i][k] = 0.f; db_acc[i][k] = 0.f; }
  for (long row = blockIdx.x; row < a.M; row += gridDim.x) {
    const long trow = a.rope_per_sample ? row : (row % a.L);
    float mq = 0.f, rq = 1.f, mk = 0.f, rk = 1.f;
    if (do_norm) {
      const float4 st = *reinterpret_cast<const float4*>(a.stats + row * 4);
      mq = st.x; rq = st.y; mk = st.z; rk = st.w;
    }
    float gl[NIB][8], gh[NIB][8], xl[NIB][8], xh[NIB][8];
    float red[4] = {0.f, 0.f, 0.f, 0.f};  // sum g (q), sum g*xhat (q), sum g (k), sum g*xhat (k)
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int t = i * 256 + tid;
      if (t < 2 * per_part) {
        const int part = t / per_part, r = t % per_part;
        const int hc = (r / per_head) * a.D + (r % per_head) * 8;
        const int pc = (r % per_head) * 8;
        float dl[8], dh[8], cs[8], sn[8];
        load8_bf16(a.dqkr + row * 2 * a.d + part * a.d + hc, dl);
        load8_bf16(a.dqkr + row * 2 * a.d + part * a.d + hc + half, dh);
        if (part == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) { dl[k] *= a.q_scale; dh[k] *= a.q_scale; }
        }
        load8_f32(a.cos_t + trow * half + pc, cs);
        load8_f32(a.sin_t + trow * half + pc, sn);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          gl[i][k] = dl[k] * cs[k] + dh[k] * sn[k];
          gh[i][k] = dh[k] * cs[k] - dl[k] * sn[k];
        }
        if (do_norm) {
          const float* g = aff + part * a.d;
          const float m = part ? mk : mq, rs = part ? rk : rq;
          float g0[8], g1[8], r0[8], r1[8];
          load8_f32(g + hc, g0); load8_f32(g + hc + half, g1);
          load8_bf16(a.qkv + row * 3 * a.d + part * a.d + hc, r0);
          load8_bf16(a.qkv + row * 3 * a.d + part * a.d + hc + half, r1);
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            xl[i][k] = (r0[k] - m) * rs;
            xh[i][k] = (r1[k] - m) * rs;
            dg_acc[i][k] += gl[i][k] * xl[i][k];
            dg_acc[i][k + 8] += gh[i][k] * xh[i][k];
            db_acc[i][k] += gl[i][k];
            db_acc[i][k + 8] += gh[i][k];
            gl[i][k] *= g0[k];
            gh[i][k] *= g1[k];
            s1 += gl[i][k] + gh[i][k];
            s2 += gl[i][k] * xl[i][k] + gh[i][k] * xh[i][k];
          }
          red[2 * part] += s1;
          red[2 * part + 1] += s2;
        }
      }
    }
    if (do_norm) {
      block_sum4<4>(red, sm);
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k] /= a.d;
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int t = i * 256 + tid;
      if (t >= 2 * per_part) continue;
      const int part = t / per_part, r = t % per_part;
      const int hc = (r / per_head) * a.D + (r % per_head) * 8;
      float ol[8], oh[8];
      if (do_norm) {
        const float rs = part ? rk : rq, sg = red[2 * part], sgx = red[2 * part + 1];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          ol[k] = rs * (gl[i][k] - sg - xl[i][k] * sgx);
          oh[k] = rs * (gh[i][k] - sg - xh[i][k] * sgx);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ol[k] = gl[i][k]; oh[k] = gh[i][k]; }
      }
      store8_bf16(a.dqkv + row * 3 * a.d + part * a.d + hc, ol);
      store8_bf16(a.dqkv + row * 3 * a.d + part * a.d + hc + half, oh);
    }
  }
  if (!do_norm) return;
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int t = i * 256 + tid;
    if (t < 2 * per_part) {
      const int part = t / per_part, r = t % per_part;
      const int hc = (r / per_head) * a.D + (r % per_head) * 8;
      if (a.ws) {  // workspace row layout: [dgq | dbq | dgk | dbk], d floats each
        float* wr = a.ws + (long)blockIdx.x * 4 * a.d + part * 2 * a.d;
        float t0[8], t1[8], t2[8], t3[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { t0[k] = dg_acc[i][k]; t1[k] = dg_acc[i][k + 8]; t2[k] = db_acc[i][k]; t3[k] = db_acc[i][k + 8]; }
        store8_f32(wr + hc, t0); store8_f32(wr + hc + half, t1);
        store8_f32(wr + a.d + hc, t2); store8_f32(wr + a.d + hc + half, t3);
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int col = hc + (k & 7) + (k >> 3) * half;
          atomicAdd((part ? a.dgk : a.dgq) + col, dg_acc[i][k]);
          atomicAdd((part ? a.dbk : a.dbq) + col, db_acc[i][k]);
        }
      }
    }
  }
}

inline int grid_rows(long M) {
  long g = (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (int)(g < 2048 ? (g < 1 ? 1 : g) : 2048);
}
inline int nch_for(long d) { return (int)((d + 511) / 512); }

#define DISPATCH_NCH_MOD(nch, KERNEL, grid, stream, args)                                            \
  switch (nch) {                                                                                     \
    case 1: hipLaunchKernelGGL((KERNEL<1, true>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 2: hipLaunchKernelGGL((KERNEL<2, true>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 3: hipLaunchKernelGGL((KERNEL<3, true>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 4: hipLaunchKernelGGL((KERNEL<4, true>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 8: hipLaunchKernelGGL((KERNEL<8, true>), dim3(grid), dim3(256), 0, stream, args); break;   \
    default: udm_set_error(#KERNEL ": unsupported hidden size (d <= 2048 or d == 4096, d %% 8 == 0)"); return 2; \
  }
#define DISPATCH_NCH(nch, KERNEL, grid, stream, args)                                          \
  switch (nch) {                                                                               \
    case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 3: hipLaunchKernelGGL((KERNEL<3>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(grid), dim3(256), 0, stream, args); break;   \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(grid), dim3(256), 0, stream, args); break;   \
    default: udm_set_error(#KERNEL ": unsupported hidden size (d <= 2048 or d == 4096, d %% 8 == 0)"); return 2; \
  }
}  // namespace

extern "C" int udm_norm_fwd(const float* x, void* y, float* rstd, float* mean, const float* w, const void* shift, const void* scale, int64_t mod_stride,
                            const int64_t* modality, const int* any_img, int64_t M, int64_t d, int64_t L, int norm_type, float eps, hipStream_t stream) {
  UDM_CHECK_ARG(x && y && rstd && w, "udm_norm_fwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && L > 0, "udm_norm_fwd: bad shape M=%ld d=%ld L=%ld", (long)M, (long)d, (long)L);
  UDM_CHECK_ARG(norm_type == 0 || mean, "udm_norm_fwd: LayerNorm needs a mean buffer");
  UDM_CHECK_ARG((shift == nullptr) == (scale == nullptr), "udm_norm_fwd: shift and scale go together");
  NormArgs a{x, (bf16_t*)y, rstd, norm_type ? mean : nullptr, w, (const bf16_t*)shift, (const bf16_t*)scale, modality, any_img, (long)mod_stride,
             (int)M, (int)d, (int)L, norm_type, eps};
  int nch = nch_for(d); if (nch > 4) nch = 8;
  DISPATCH_NCH(nch, norm_fwd_kernel, grid_rows(M), stream, a);
  UDM_CHECK_LAUNCH("udm_norm_fwd");
  return 0;
}

extern "C" int udm_norm_bwd(const void* dy, const float* x, const float* rstd, const float* mean, const float* w, const void* shift, const void* scale,
                            int64_t mod_stride, const int64_t* modality, const int* any_img, float* dx, float* dw, float* dshift, float* dscale,
                            int64_t M, int64_t d, int64_t L, int norm_type, int accumulate, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(dy && x && rstd && w && dx && dw, "udm_norm_bwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && L > 0, "udm_norm_bwd: bad shape");
  UDM_CHECK_ARG(!shift || (scale && dshift && dscale), "udm_norm_bwd: modulated norm needs scale, dshift, dscale");
  NormBwdArgs a{(const bf16_t*)dy, x, rstd, mean, w, (const bf16_t*)shift, (const bf16_t*)scale, modality, any_img, dx, dw, dshift, dscale,
                (long)mod_stride, (int)M, (int)d, (int)L, norm_type, accumulate, nullptr, 0};
  int nch = nch_for(d); if (nch > 4) nch = 8;
  int grid = min(grid_rows(M), d < 2048 ? 1024 : 512);   // (measured: 48.6 vs 51.6 us at d = 768 with 1024 blocks, 60.7 vs 57.9 us at d = 2048)
  if (shift) {   // modulated: whole blocks per batch element (M = B L)
    UDM_CHECK_ARG(M % L == 0, "udm_norm_bwd: modulated norm needs M = B * L");
    const int B = (int)(M / L);
    a.bpb = max(1, min(grid / B, (int)((L + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)));
    grid = B * a.bpb;
  }
  if (ws && ws_elems >= (int64_t)grid * d && grid >= 64) a.ws = ws;   // short chains (few blocks) stay on atomics
  else if (!shift) grid = min(grid, 512);
  if (shift) { DISPATCH_NCH_MOD(nch, norm_bwd_kernel, grid, stream, a); }
  else { DISPATCH_NCH(nch, norm_bwd_kernel, grid, stream, a); }
  UDM_CHECK_LAUNCH("udm_norm_bwd");
  if (a.ws) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((d + 63) / 64), 16), dim3(256), 0, stream, (const float*)ws, dw, grid, (int)d);
    UDM_CHECK_LAUNCH("udm_norm_bwd(colreduce)");
  }
  return 0;
}

// residual add + the next pre-norm WITH adaLN modulation in one pass (time_conditioning = True)
extern "C" int udm_residual_norm_fwd_ada(const float* x_in, const void* branch, float* x_out, const float* w_b, float* rstd_b, float* mean_b, const void* gate,
                                         int64_t mod_stride, const int64_t* modality, int64_t M, int64_t d, int64_t L, int norm_type, float eps, float p_drop,
                                         uint64_t seed, const float* w_next, void* h_out, float* rstd_next, float* mean_next, const void* next_shift,
                                         const void* next_scale, int64_t next_mod_stride, const int64_t* next_modality, const int* next_any_img, hipStream_t stream) {
  UDM_CHECK_ARG(x_in && branch && x_out && w_next && h_out && rstd_next && next_shift && next_scale, "udm_residual_norm_fwd_ada: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && L > 0, "udm_residual_norm_fwd_ada: bad shape");
  UDM_CHECK_ARG(!w_b || rstd_b, "udm_residual_norm_fwd_ada: sandwich norm needs rstd buffer");
  UDM_CHECK_ARG(!(w_b && norm_type) || mean_b, "udm_residual_norm_fwd_ada: sandwich LayerNorm needs mean buffer");
  UDM_CHECK_ARG(!norm_type || mean_next, "udm_residual_norm_fwd_ada: LayerNorm needs mean_next");
  UDM_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "udm_residual_norm_fwd_ada: dropout p out of range");
  ResidArgs a{x_in, (const bf16_t*)branch, x_out, w_b, rstd_b, (w_b && norm_type) ? mean_b : nullptr, (const bf16_t*)gate, modality, (long)mod_stride,
              (int)M, (int)d, (int)L, norm_type, eps, p_drop, seed, w_next, (bf16_t*)h_out, rstd_next, norm_type ? mean_next : nullptr};
  a.n_shift = (const bf16_t*)next_shift; a.n_scale = (const bf16_t*)next_scale; a.n_mod_stride = (long)next_mod_stride; a.n_modality = next_modality;
  a.n_any_img = next_any_img;
  int nch = nch_for(d); if (nch > 4) nch = 8;
  DISPATCH_NCH(nch, residual_fwd_kernel, grid_rows(M), stream, a);
  UDM_CHECK_LAUNCH("udm_residual_norm_fwd_ada");
  return 0;
}

extern "C" int udm_residual_fwd(const float* x_in, const void* branch, float* x_out, const float* w_b, float* rstd_b, float* mean_b, const void* gate,
                                int64_t mod_stride, const int64_t* modality, int64_t M, int64_t d, int64_t L, int norm_type, float eps, float p_drop,
                                uint64_t seed, hipStream_t stream) {
  UDM_CHECK_ARG(x_in && branch && x_out, "udm_residual_fwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && L > 0, "udm_residual_fwd: bad shape");
  UDM_CHECK_ARG(!w_b || rstd_b, "udm_residual_fwd: sandwich norm needs rstd buffer");
  UDM_CHECK_ARG(!(w_b && norm_type) || mean_b, "udm_residual_fwd: sandwich LayerNorm needs mean buffer");
  UDM_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "udm_residual_fwd: dropout p out of range");
  ResidArgs a{x_in, (const bf16_t*)branch, x_out, w_b, rstd_b, (w_b && norm_type) ? mean_b : nullptr, (const bf16_t*)gate, modality, (long)mod_stride,
              (int)M, (int)d, (int)L, norm_type, eps, p_drop, seed, nullptr, nullptr, nullptr, nullptr};
  int nch = nch_for(d); if (nch > 4) nch = 8;
  DISPATCH_NCH(nch, residual_fwd_kernel, grid_rows(M), stream, a);
  UDM_CHECK_LAUNCH("udm_residual_fwd");
  return 0;
}

extern "C" int udm_residual_norm_fwd(const float* x_in, const void* branch, float* x_out, const float* w_b, float* rstd_b, float* mean_b, const void* gate,
                                     int64_t mod_stride, const int64_t* modality, int64_t M, int64_t d, int64_t L, int norm_type, float eps, float p_drop,
                                     uint64_t seed, const float* w_next, void* h_out, float* rstd_next, float* mean_next, hipStream_t stream) {
  UDM_CHECK_ARG(x_in && branch && x_out && w_next && h_out && rstd_next, "udm_residual_norm_fwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && L > 0, "udm_residual_norm_fwd: bad shape");
  UDM_CHECK_ARG(!w_b || rstd_b, "udm_residual_norm_fwd: sandwich norm needs rstd buffer");
  UDM_CHECK_ARG(!(w_b && norm_type) || mean_b, "udm_residual_norm_fwd: sandwich LayerNorm needs mean buffer");
  UDM_CHECK_ARG(!norm_type || mean_next, "udm_residual_norm_fwd: LayerNorm needs mean_next");
  UDM_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "udm_residual_norm_fwd: dropout p out of range");
  ResidArgs a{x_in, (const bf16_t*)branch, x_out, w_b, rstd_b, (w_b && norm_type) ? mean_b : nullptr, (const bf16_t*)gate, modality, (long)mod_stride,
              (int)M, (int)d, (int)L, norm_type, eps, p_drop, seed, w_next, (bf16_t*)h_out, rstd_next, norm_type ? mean_next : nullptr};
  int nch = nch_for(d); if (nch > 4) nch = 8;
  DISPATCH_NCH(nch, residual_fwd_kernel, grid_rows(M), stream, a);
  UDM_CHECK_LAUNCH("udm_residual_norm_fwd");
  return 0;
}

extern "C" int udm_residual_bwd(const float* dx, const void* branch, void* dbranch, const float* w_b, const float* rstd_b, const float* mean_b,
                                const void* gate, int64_t mod_stride, const int64_t* modality, float* dw_b, float* dgate, int64_t M, int64_t d, int64_t L,
                                int norm_type, float p_drop, uint64_t seed, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(dx && branch && dbranch, "udm_residual_bwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && d % 8 == 0 && L > 0, "udm_residual_bwd: bad shape");
  UDM_CHECK_ARG(!w_b || (rstd_b && dw_b), "udm_residual_bwd: sandwich norm needs rstd and dw");
  UDM_CHECK_ARG(!gate || dgate, "udm_residual_bwd: gate needs dgate");
  ResidBwdArgs a{dx, (const bf16_t*)branch, (bf16_t*)dbranch, w_b, rstd_b, mean_b, (const bf16_t*)gate, modality, dw_b, nullptr, dgate, (long)mod_stride,
                 (int)M, (int)d, (int)L, norm_type, p_drop, seed, 0};
  if (gate) {   // gated (adaLN-Zero): whole blocks per batch element, the gate gradient's column sums in registers
    UDM_CHECK_ARG(M % L == 0, "udm_residual_bwd: a gate needs M = B * L");
    const int B = (int)(M / L);
    int nch = nch_for(d); if (nch > 4) nch = 8;
    a.bpb = max(1, min(1024 / B, (int)((L + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)));
    const int grid = B * a.bpb;
    if (w_b && ws && ws_elems >= (int64_t)grid * d && grid >= 64) a.ws = ws;
    DISPATCH_NCH_MOD(nch, residual_bwd_kernel, grid, stream, a);
    UDM_CHECK_LAUNCH("udm_residual_bwd(gated)");
    if (a.ws) {
      hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((d + 63) / 64), 16), dim3(256), 0, stream, (const float*)ws, dw_b, grid, (int)d);
      UDM_CHECK_LAUNCH("udm_residual_bwd(colreduce)");
    }
    return 0;
  }
  if (d == 2048 && (!w_b || (ws && ws_elems >= (int64_t)1024 * d))) {
    // d = 2048 still fits a wave per row (32 values per lane): no block-wide reductions; measured 41.8 us vs 52.9 us for the
    // block-per-row form without dropout, equal with dropout (Philox regeneration dominates there)
    const int grid = 1024;
    a.ws = w_b ? ws : nullptr;
    hipLaunchKernelGGL((residual_bwd_kernel<4>), dim3(grid), dim3(256), 0, stream, a);
    UDM_CHECK_LAUNCH("udm_residual_bwd");
    if (a.ws) hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((d + 63) / 64), 16), dim3(256), 0, stream, (const float*)ws, dw_b, grid, (int)d);
    return 0;
  }
  if (d >= 2048 && d <= 4096) {  // wide rows: block-per-row form (8 elements per thread, high occupancy)
    int g = (int)(M < 1536 ? M : 1536);
    if (w_b && ws && ws_elems >= (int64_t)g * d) a.ws = ws;
    else if (w_b) g = g < 256 ? g : 256;  // no workspace: keep the same-address atomic chains short
    if (d <= 2048) hipLaunchKernelGGL((residual_bwd_brow_kernel<1>), dim3(g), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((residual_bwd_brow_kernel<2>), dim3(g), dim3(256), 0, stream, a);
    UDM_CHECK_LAUNCH("udm_residual_bwd");
    if (a.ws) {
      hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((d + 63) / 64), 16), dim3(256), 0, stream, (const float*)ws, dw_b, g, (int)d);
      UDM_CHECK_LAUNCH("udm_residual_bwd(colreduce)");
    }
    return 0;
  }
  int nch = nch_for(d); if (nch > 4) nch = 8;
  int grid = min(grid_rows(M), 512);
  if (w_b && ws && grid_rows(M) >= 1024 && ws_elems >= (int64_t)1024 * d) { grid = 1024; a.ws = ws; }   // wide grid, column sums through the workspace
  DISPATCH_NCH(nch, residual_bwd_kernel, grid, stream, a);
  UDM_CHECK_LAUNCH("udm_residual_bwd");
  if (a.ws) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((d + 63) / 64), 16), dim3(256), 0, stream, (const float*)ws, dw_b, grid, (int)d);
    UDM_CHECK_LAUNCH("udm_residual_bwd(colreduce)");
  }
  return 0;
}

extern "C" int udm_norm_residual_bwd(const void* dy, const float* x, const float* rstd, const float* mean, const float* w, float* dx, float* dw, int accumulate,
                                     const void* branch, void* dbranch, const float* w_b, const float* rstd_b, const float* mean_b, float* dw_b, float* dbias,
                                     int64_t M, int64_t d, int norm_type, float p_drop, uint64_t seed, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(dy && x && rstd && w && dx && dw && branch && dbranch && ws, "udm_norm_residual_bwd: null pointer");
  UDM_CHECK_ARG(M > 0 && (d == 2048 || d == 4096 || (d % 8 == 0 && d >= 64 && d < 2048)),
                "udm_norm_residual_bwd: the fused form is built for d = 2048 / 4096 (block per row) and d < 2048, d %% 8 == 0 (wave per row); got %ld", (long)d);
  UDM_CHECK_ARG(norm_type == 0 || mean, "udm_norm_residual_bwd: LayerNorm needs the saved mean");
  UDM_CHECK_ARG(!w_b || (rstd_b && dw_b && (norm_type == 0 || mean_b)), "udm_norm_residual_bwd: sandwich norm needs rstd_b, dw_b (and mean_b for LayerNorm)");
  // 3 blocks per CU: every block leaves a [3][d] fp32 partial for colreduce3, and at 1536 blocks that workspace (38 MB written + read per call) cost more
  // than the extra occupancy gave (in the step: 3.95 ms at 1536 blocks, 4.12 at 1024, 3.73 at 768, 3.77 at 512)
  const bool wrow = d < 2048;
  const int grid = wrow ? min(grid_rows(M), 1024) : (int)(M < 768 ? M : 768);
  UDM_CHECK_ARG(ws_elems >= (int64_t)grid * 3 * d, "udm_norm_residual_bwd: workspace too small (need %ld floats)", (long)grid * 3 * d);
  NormResidBwdArgs a{(const bf16_t*)dy, x, rstd, mean, w, dx, dw, accumulate, (const bf16_t*)branch, (bf16_t*)dbranch, w_b, rstd_b, mean_b, dw_b, dbias, ws,
                     (int)M, (int)d, norm_type, p_drop, seed};
  if (wrow) {
    switch (nch_for(d)) {
      case 1: hipLaunchKernelGGL((norm_residual_bwd_wrow_kernel<1>), dim3(grid), dim3(256), 0, stream, a); break;
      case 2: hipLaunchKernelGGL((norm_residual_bwd_wrow_kernel<2>), dim3(grid), dim3(256), 0, stream, a); break;
      case 3: hipLaunchKernelGGL((norm_residual_bwd_wrow_kernel<3>), dim3(grid), dim3(256), 0, stream, a); break;
      default: hipLaunchKernelGGL((norm_residual_bwd_wrow_kernel<4>), dim3(grid), dim3(256), 0, stream, a); break;
    }
  } else if (d == 2048) hipLaunchKernelGGL((norm_residual_bwd_kernel<1>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((norm_residual_bwd_kernel<2>), dim3(grid), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_norm_residual_bwd");
  hipLaunchKernelGGL(colreduce3_kernel, dim3((unsigned)((d + 63) / 64), 16, 3), dim3(256), 0, stream, (const float*)ws, dw, w_b ? dw_b : nullptr, dbias, grid, (int)d);
  UDM_CHECK_LAUNCH("udm_norm_residual_bwd(colreduce)");
  return 0;
}

// the fused pass with adaLN-Zero modulation of the norm and / or a gated residual branch (d = 2048 / 4096: the block-per-row form)
extern "C" int udm_norm_residual_bwd_ada(const void* dy, const float* x, const float* rstd, const float* mean, const float* w, float* dx, float* dw, int accumulate,
                                         const void* branch, void* dbranch, const float* w_b, const float* rstd_b, const float* mean_b, float* dw_b, float* dbias,
                                         const void* shift, const void* scale, float* dshift, float* dscale, const void* gate, float* dgate, int64_t mod_stride,
                                         const int64_t* modality, const int* any_img, const int64_t* modality_r, int64_t M, int64_t d, int64_t L, int norm_type,
                                         float p_drop, uint64_t seed, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(dy && x && rstd && w && dx && dw && branch && dbranch && ws, "udm_norm_residual_bwd_ada: null pointer");
  UDM_CHECK_ARG(M > 0 && L > 0 && M % L == 0 && (d == 2048 || d == 4096), "udm_norm_residual_bwd_ada: d = 2048 / 4096 and M = B * L (got M=%ld L=%ld d=%ld)", (long)M, (long)L, (long)d);
  UDM_CHECK_ARG(norm_type == 0 || mean, "udm_norm_residual_bwd_ada: LayerNorm needs the saved mean");
  UDM_CHECK_ARG(!w_b || (rstd_b && dw_b && (norm_type == 0 || mean_b)), "udm_norm_residual_bwd_ada: sandwich norm needs rstd_b, dw_b (and mean_b for LayerNorm)");
  UDM_CHECK_ARG(!shift || (scale && dshift && dscale), "udm_norm_residual_bwd_ada: a modulated norm needs scale, dshift, dscale");
  UDM_CHECK_ARG(!gate || dgate, "udm_norm_residual_bwd_ada: a gate needs dgate");
  const int B = (int)(M / L);
  const int bpb = max(1, min(768 / B, (int)L));
  const int grid = B * bpb;
  UDM_CHECK_ARG(ws_elems >= (int64_t)grid * 6 * d, "udm_norm_residual_bwd_ada: workspace too small (need %ld floats)", (long)grid * 6 * d);
  NormResidBwdArgs a{(const bf16_t*)dy, x, rstd, mean, w, dx, dw, accumulate, (const bf16_t*)branch, (bf16_t*)dbranch, w_b, rstd_b, mean_b, dw_b, dbias, ws,
                     (int)M, (int)d, norm_type, p_drop, seed};
  a.shift = (const bf16_t*)shift; a.scale = (const bf16_t*)scale; a.dshift = dshift; a.dscale = dscale; a.gate = (const bf16_t*)gate; a.dgate = dgate;
  a.mod_stride = (long)mod_stride; a.modality = modality; a.any_img = any_img; a.modality_r = modality_r; a.L = (int)L; a.bpb = bpb;
  if (d == 2048) hipLaunchKernelGGL((norm_residual_bwd_kernel<1, true>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((norm_residual_bwd_kernel<2, true>), dim3(grid), dim3(256), 0, stream, a);
  UDM_CHECK_LAUNCH("udm_norm_residual_bwd_ada");
  hipLaunchKernelGGL(colreduce3_kernel, dim3((unsigned)((d + 63) / 64), 16, 3), dim3(256), 0, stream, (const float*)ws, dw, w_b ? dw_b : nullptr, dbias, grid, (int)d);
  UDM_CHECK_LAUNCH("udm_norm_residual_bwd_ada(colreduce)");
  hipLaunchKernelGGL(ada_reduce_kernel, dim3((unsigned)((d + 255) / 256), B, 3), dim3(256), 0, stream, (const float*)(ws + (long)grid * 3 * d), shift ? dshift : nullptr,
                     shift ? dscale : nullptr, gate ? dgate : nullptr, (long)mod_stride, bpb, (int)d);
  UDM_CHECK_LAUNCH("udm_norm_residual_bwd_ada(ada reduce)");
  return 0;
}

extern "C" int udm_qknorm_rope_fwd(const void* qkv, void* qkr, const float* gq, const float* bq, const float* gk, const float* bk, float* stats,
                                   const float* cos_t, const float* sin_t, int rope_per_sample, int64_t M, int64_t d, int64_t L, int64_t D, float eps,
                                   float q_scale, hipStream_t stream) {
  UDM_CHECK_ARG(qkv && qkr && cos_t && sin_t, "udm_qknorm_rope_fwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && D > 0 && d % D == 0 && D % 16 == 0, "udm_qknorm_rope_fwd: bad shape d=%ld D=%ld", (long)d, (long)D);
  UDM_CHECK_ARG(!gq || (bq && gk && bk && stats), "udm_qknorm_rope_fwd: qk-norm needs all four affine vectors and stats");
  QkArgs a{(const bf16_t*)qkv, (bf16_t*)qkr, gq, bq, gk, bk, stats, cos_t, sin_t, (int)M, (int)d, (int)L, (int)D, rope_per_sample, eps, q_scale};
  UDM_CHECK_ARG(D % 16 == 0 && d % 16 == 0, "udm_qknorm_rope_fwd: head_dim and hidden size must be multiples of 16");
  const int nit = (int)((2 * (d / 16) + 63) / 64);
  UDM_CHECK_ARG(nit <= 8, "udm_qknorm_rope_fwd: hidden size too large");
  const int nch = nit <= 1 ? 1 : (nit == 2 ? 2 : (nit == 3 ? 3 : (nit == 4 ? 4 : 8)));
  const size_t lds = gq ? (size_t)4 * d * sizeof(float) : 0;
  if (d == 2048) {   // two rows per block iteration, 1024 blocks (in the step: 1.08-1.10 ms against 1.22-1.24 for one row per iteration; 3 rows 1.18, 4 rows 1.41)
    const long groups = (M + 1) / 2;
    const int g = (int)(groups < 1024 ? groups : 1024);
    hipLaunchKernelGGL((qknorm_rope_fwd_brow_rows_kernel<2>), dim3(g), dim3(256), 0, stream, a);
    UDM_CHECK_LAUNCH("udm_qknorm_rope_fwd");
    return 0;
  }
  if (d >= 2048 && d <= 4096) {
    const int g = (int)(M < 2048 ? M : 2048);
    if (d <= 2048) hipLaunchKernelGGL((qknorm_rope_fwd_brow_kernel<1>), dim3(g), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((qknorm_rope_fwd_brow_kernel<2>), dim3(g), dim3(256), lds, stream, a);
    UDM_CHECK_LAUNCH("udm_qknorm_rope_fwd");
    return 0;
  }
  const int grid = min(grid_rows(M), 1024);
  switch (nch) {
    case 1: hipLaunchKernelGGL((qknorm_rope_fwd_kernel<1>), dim3(grid), dim3(256), lds, stream, a); break;
    case 2: hipLaunchKernelGGL((qknorm_rope_fwd_kernel<2>), dim3(grid), dim3(256), lds, stream, a); break;
    case 3: hipLaunchKernelGGL((qknorm_rope_fwd_kernel<3>), dim3(grid), dim3(256), lds, stream, a); break;
    case 4: hipLaunchKernelGGL((qknorm_rope_fwd_kernel<4>), dim3(grid), dim3(256), lds, stream, a); break;
    default: hipLaunchKernelGGL((qknorm_rope_fwd_kernel<8>), dim3(grid), dim3(256), lds, stream, a); break;
  }
  UDM_CHECK_LAUNCH("udm_qknorm_rope_fwd");
  return 0;
}

extern "C" int udm_qknorm_rope_bwd(const void* dqkr, const void* qkv, void* dqkv, const float* gq, const float* gk, const float* stats, const float* cos_t,
                                   const float* sin_t, int rope_per_sample, float* dgq, float* dbq, float* dgk, float* dbk, int64_t M, int64_t d, int64_t L,
                                   int64_t D, float q_scale, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(dqkr && qkv && dqkv && cos_t && sin_t, "udm_qknorm_rope_bwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && D > 0 && d % D == 0 && D % 8 == 0, "udm_qknorm_rope_bwd: bad shape");
  UDM_CHECK_ARG(!gq || (gk && stats && dgq && dbq && dgk && dbk), "udm_qknorm_rope_bwd: qk-norm needs gk, stats and the four gradient vectors");
  QkBwdArgs a{(const bf16_t*)dqkr, (const bf16_t*)qkv, (bf16_t*)dqkv, gq, gk, stats, cos_t, sin_t, dgq, dbq, dgk, dbk, nullptr, (int)M, (int)d, (int)L, (int)D,
              rope_per_sample, q_scale};
  UDM_CHECK_ARG(D % 16 == 0 && d % 16 == 0, "udm_qknorm_rope_bwd: head_dim and hidden size must be multiples of 16");
  const int nit = (int)((2 * (d / 16) + 63) / 64);
  UDM_CHECK_ARG(nit <= 8, "udm_qknorm_rope_bwd: hidden size too large");
  const int nch = nit <= 1 ? 1 : (nit == 2 ? 2 : (nit == 3 ? 3 : (nit == 4 ? 4 : 8)));
  const size_t lds = gq ? ((size_t)2 * d + 4096) * sizeof(float) : 0;
  if (d >= 2048 && d <= 4096) {
    int g = (int)(M < 1024 ? M : 1024);   // (swept 256 .. 2048 in the step: 2.92 / 1.91 / 1.69 / 1.60 / 1.88 / 1.82 / 1.87 ms per step at 256 / 512 / 768 / 1024 / 1280 / 1536 / 2048)
    UDM_CHECK_ARG(!gq || (dbq == dgq + d && dgk == dgq + 2 * d && dbk == dgq + 3 * d) || !ws, "udm_qknorm_rope_bwd: the workspace form needs dgq|dbq|dgk|dbk contiguous");
    if (gq && ws && ws_elems >= (int64_t)g * 4 * d) a.ws = ws;
    else if (gq) g = g < 256 ? g : 256;
    const size_t l2 = gq ? (size_t)2 * d * sizeof(float) : 0;
    if (d <= 2048) hipLaunchKernelGGL((qknorm_rope_bwd_brow_kernel<1>), dim3(g), dim3(256), l2, stream, a);
    else hipLaunchKernelGGL((qknorm_rope_bwd_brow_kernel<2>), dim3(g), dim3(256), l2, stream, a);
    UDM_CHECK_LAUNCH("udm_qknorm_rope_bwd");
    if (a.ws) {
      hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((4 * d + 63) / 64), 16), dim3(256), 0, stream, (const float*)ws, dgq, g, (int)(4 * d));
      UDM_CHECK_LAUNCH("udm_qknorm_rope_bwd(colreduce)");
    }
    return 0;
  }
  // narrow rows (wave per row): with a workspace the column sums go through it and the grid can be wide enough to hide HBM latency (a same-address
  // atomic per block and column limited it to 256 blocks: 117 us at d = 768, M = 24576); without one, the short atomic chains stay
  int grid = min(grid_rows(M), 256);
  if (gq && ws && dbq == dgq + d && dgk == dgq + 2 * d && dbk == dgq + 3 * d) {
    const int wide = min(grid_rows(M), 1024);
    if (ws_elems >= (int64_t)wide * 4 * d && wide >= 64) { grid = wide; a.ws = ws; }
  }
  switch (nch) {
    case 1: hipLaunchKernelGGL((qknorm_rope_bwd_kernel<1>), dim3(grid), dim3(256), lds, stream, a); break;
    case 2: hipLaunchKernelGGL((qknorm_rope_bwd_kernel<2>), dim3(grid), dim3(256), lds, stream, a); break;
    case 3: hipLaunchKernelGGL((qknorm_rope_bwd_kernel<3>), dim3(grid), dim3(256), lds, stream, a); break;
    case 4: hipLaunchKernelGGL((qknorm_rope_bwd_kernel<4>), dim3(grid), dim3(256), lds, stream, a); break;
    default: hipLaunchKernelGGL((qknorm_rope_bwd_kernel<8>), dim3(grid), dim3(256), lds, stream, a); break;
  }
  UDM_CHECK_LAUNCH("udm_qknorm_rope_bwd");
  if (a.ws) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((4 * d + 63) / 64), 16), dim3(256), 0, stream, (const float*)ws, dgq, grid, (int)(4 * d));
    UDM_CHECK_LAUNCH("udm_qknorm_rope_bwd(colreduce)");
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Linear backward for a SMALL batch (adaLN_modulation: c [B, cond_dim] -> mod [B, 6 d], models/dit.py:922-925): dW = dY^T X, db += colsum(dY), dX += dY W in ONE
// launch.  dY arrives fp32 (it is the atomically accumulated shift / scale / gate gradient) and is rounded to bf16 on load - the values the autocast backward
// multiplies.  Replaces cast + two transposes + a K = B weight-gradient GEMM + an M = B, one-tile, K = 6 d input-gradient GEMM (~140 us: a single workgroup walking
// 12 288 k) per block.  A block owns 32 output features: dW rows written whole, dX partials by atomics (B x in addresses, out / 32 adds each).
// ---------------------------------------------------------------------------------------------
constexpr int SBL_OB = 32, SBL_MAX_B = 64, SBL_MAX_IN = 128, SBL_MAX_BLOCKS = 128;
__global__ __launch_bounds__(256) void small_batch_linear_bwd_kernel(const float* __restrict__ dY, long lddy, const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ W,
                                                                     long ldw, float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dX, long lddx,
                                                                     float* __restrict__ dXp, int Bp, int out, int in, int chunks_per_block) {
  __shared__ float dm[SBL_MAX_B][SBL_OB + 1];
  __shared__ __attribute__((aligned(16))) float xs[SBL_MAX_B][SBL_MAX_IN];
  __shared__ __attribute__((aligned(16))) float ws[SBL_OB][SBL_MAX_IN + 4];
  const int tid = threadIdx.x;
  const int vpr = in >> 3;   // 16-byte vectors per row of X / W (in % 8 == 0)
  for (int v = tid; v < Bp * vpr; v += 256) {
    float f8[8];
    load8_bf16(X + (long)(v / vpr) * ldx + (v % vpr) * 8, f8);
#pragma unroll
    for (int k = 0; k < 8; ++k) xs[v / vpr][(v % vpr) * 8 + k] = f8[k];
  }
  // dX partials of this block's features stay in registers over its chunks.  Thread = (batch row mod 8, 4-column group): all 256 threads work at any batch size
  float accx[SBL_MAX_B / 8][4];
#pragma unroll
  for (int pss = 0; pss < SBL_MAX_B / 8; ++pss)
#pragma unroll
    for (int j = 0; j < 4; ++j) accx[pss][j] = 0.f;
  const int xb = tid >> 5, xn0 = (tid & 31) * 4;
  for (int ch = 0; ch < chunks_per_block; ++ch) {
    const int o0 = (blockIdx.x * chunks_per_block + ch) * SBL_OB;
    if (o0 >= out) break;   // block-uniform
    __syncthreads();        // (the previous chunk's readers are done with dm / ws; first pass: xs is complete)
    for (int idx = tid; idx < Bp * SBL_OB; idx += 256) {
      const int b = idx / SBL_OB, ol = idx % SBL_OB;
      dm[b][ol] = (o0 + ol < out) ? rbf(dY[(long)b * lddy + o0 + ol]) : 0.f;
    }
    {   // this chunk's 32 weight rows: at most two 16-byte vectors per thread, both requested before either is stored
      float f8[2][8];
      bool have[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int v = tid + 256 * u, ol = v / vpr;
        have[u] = v < SBL_OB * vpr;
        if (have[u] && o0 + ol < out) load8_bf16(W + (long)(o0 + ol) * ldw + (v % vpr) * 8, f8[u]);
        else {
#pragma unroll
          for (int k = 0; k < 8; ++k) f8[u][k] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int v = tid + 256 * u;
        if (have[u]) {
#pragma unroll
          for (int k = 0; k < 8; ++k) ws[v / vpr][(v % vpr) * 8 + k] = f8[u][k];
        }
      }
    }
    __syncthreads();
    if (db && tid < SBL_OB && o0 + tid < out) {
      float sacc = 0.f;
      for (int b = 0; b < Bp; ++b) sacc += dm[b][tid];
      db[o0 + tid] += sacc;
    }
    {   // dW[o, n] = sum_b dY[b, o] X[b, n]: thread = (feature ol, 16-column group); X rows read as 16-byte vectors
      const int ol = tid >> 3, n0 = (tid & 7) * 16;
      if (o0 + ol < out && n0 < in) {
        float4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < Bp; ++b) {
          const float dv = dm[b][ol];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[b][n0 + 4 * q]);   // (columns past `in` hold stale finite values: never stored)
            acc[q].x += dv * xv.x; acc[q].y += dv * xv.y; acc[q].z += dv * xv.z; acc[q].w += dv * xv.w;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n0 + 4 * q < in) *reinterpret_cast<float4*>(dW + (long)(o0 + ol) * in + n0 + 4 * q) = acc[q];
      }
    }
    // dX[b, n] += sum_{o in chunk} dY[b, o] W[o, n]
    if (xn0 < in) {
#pragma unroll
      for (int pss = 0; pss < SBL_MAX_B / 8; ++pss) {
        const int b = xb + 8 * pss;
        if (b < Bp) {
          for (int ol = 0; ol < SBL_OB; ++ol) {
            const float dv = dm[b][ol];
            const float4 wv = *reinterpret_cast<const float4*>(&ws[ol][xn0]);
            accx[pss][0] += dv * wv.x; accx[pss][1] += dv * wv.y; accx[pss][2] += dv * wv.z; accx[pss][3] += dv * wv.w;
          }
        }
      }
    }
  }
  if (xn0 < in) {
#pragma unroll
    for (int pss = 0; pss < SBL_MAX_B / 8; ++pss) {
      const int b = xb + 8 * pss;
      if (b < Bp) {
        if (dXp) {   // this block's partial tile, plain stores: the caller sums the tiles (no same-address atomic chains: 23 of 85 us)
          *reinterpret_cast<float4*>(dXp + ((long)blockIdx.x * Bp + b) * in + xn0) = make_float4(accx[pss][0], accx[pss][1], accx[pss][2], accx[pss][3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicAdd(dX + (long)b * lddx + xn0 + j, accx[pss][j]);
        }
      }
    }
  }
}

extern "C" int udm_small_batch_linear_bwd_blocks(int64_t out) {   // number of partial tiles udm_small_batch_linear_bwd writes into dX_parts for `out` features
  const int chunks = (int)((out + SBL_OB - 1) / SBL_OB);
  const int cpb = (chunks + SBL_MAX_BLOCKS - 1) / SBL_MAX_BLOCKS;
  return (chunks + cpb - 1) / cpb;
}

extern "C" int udm_small_batch_linear_bwd(const float* dY, int64_t lddy, const void* X, int64_t ldx, const void* W, int64_t ldw, float* dW, float* db, float* dX,
                                          int64_t lddx, float* dX_parts, int64_t B, int64_t out, int64_t in, hipStream_t stream) {
  UDM_CHECK_ARG(dY && X && W && dW && (dX || dX_parts), "udm_small_batch_linear_bwd: null pointer");
  UDM_CHECK_ARG(B > 0 && B <= SBL_MAX_B && in > 0 && in <= SBL_MAX_IN && in % 8 == 0 && out > 0,
                "udm_small_batch_linear_bwd: B <= 64 and in_features <= 128, a multiple of 8 (got B=%ld in=%ld out=%ld)", (long)B, (long)in, (long)out);
  UDM_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && (uintptr_t)X % 16 == 0 && (uintptr_t)W % 16 == 0 && (uintptr_t)dW % 16 == 0 && (uintptr_t)dX_parts % 16 == 0,
                "udm_small_batch_linear_bwd: X / W rows, dW and dX_parts must be 16-byte aligned");
  const int chunks = (int)((out + SBL_OB - 1) / SBL_OB);
  const int cpb = (chunks + SBL_MAX_BLOCKS - 1) / SBL_MAX_BLOCKS;
  const int grid = (chunks + cpb - 1) / cpb;
  hipLaunchKernelGGL(small_batch_linear_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, stream, dY, (long)lddy, (const bf16_t*)X, (long)ldx, (const bf16_t*)W, (long)ldw, dW, db,
                     dX, (long)lddx, dX_parts, (int)B, (int)out, (int)in, cpb);
  UDM_CHECK_LAUNCH("udm_small_batch_linear_bwd");
  return 0;
}

extern "C" int udm_embedding_fwd(const int64_t* ids, const float* E, const int64_t* modality, const float* Em, float* x, int64_t M, int64_t d, int64_t V,
                                 hipStream_t stream) {
  UDM_CHECK_ARG(ids && E && x, "udm_embedding_fwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && d % 4 == 0 && V > 0, "udm_embedding_fwd: bad shape");
  UDM_CHECK_ARG(!Em || modality, "udm_embedding_fwd: modality embedding needs the modality map");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_rows(M)), dim3(256), 0, stream, ids, E, modality, Em, x, (long)M, (int)d, (long)V);
  UDM_CHECK_LAUNCH("udm_embedding_fwd");
  return 0;
}

extern "C" int udm_embedding_bwd(const int64_t* ids, const int64_t* modality, const float* dx, float* dE, float* dEm, int64_t M, int64_t d, int64_t V,
                                 int64_t hot_id, hipStream_t stream) {
  UDM_CHECK_ARG(ids && dx && dE, "udm_embedding_bwd: null pointer");
  UDM_CHECK_ARG(M > 0 && d > 0 && V > 0, "udm_embedding_bwd: bad shape");
  UDM_CHECK_ARG(!dEm || modality, "udm_embedding_bwd: modality embedding grad needs the modality map");
  UDM_CHECK_ARG(d % 4 == 0 && ((uintptr_t)dx % 16 == 0), "udm_embedding_bwd: d must be a multiple of 4 and dx 16-byte aligned");
  const int rows_per_block = 128;   // in the 1.4 B step: 32 rows 0.208 ms, 128 rows 0.178, 256 rows 0.231 (fewer blocks = fewer same-address atomics, until the grid no longer fills the chip)
  const int grid = (int)((M + rows_per_block - 1) / rows_per_block);
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid, (unsigned)((d + 1023) / 1024)), dim3(256), 0, stream, ids, modality, dx, dE, dEm, (long)M, (int)d, (long)V,
                     (long)hot_id, rows_per_block);
  UDM_CHECK_LAUNCH("udm_embedding_bwd");
  return 0;
}

extern "C" int udm_timestep_embedding(const float* sigma, void* out, int64_t B, int64_t dim, hipStream_t stream) {
  UDM_CHECK_ARG(sigma && out && B > 0 && dim > 0, "udm_timestep_embedding: bad argument");
  const long n = B * dim;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sigma, (bf16_t*)out, (int)B, (int)dim);
  UDM_CHECK_LAUNCH("udm_timestep_embedding");
  return 0;
}
extern "C" int udm_silu_fwd(const void* x, void* y, int64_t n, hipStream_t stream) {
  UDM_CHECK_ARG(x && y && n > 0, "udm_silu_fwd: bad argument");
  hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, (long)n);
  UDM_CHECK_LAUNCH("udm_silu_fwd");
  return 0;
}
extern "C" int udm_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, hipStream_t stream) {
  UDM_CHECK_ARG(x && dy && dx && n > 0, "udm_silu_bwd: bad argument");
  hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, (long)n);
  UDM_CHECK_LAUNCH("udm_silu_bwd");
  return 0;
}
extern "C" int udm_cast_f32_bf16(const float* x, void* y, int64_t n, float scale, hipStream_t stream) {
  UDM_CHECK_ARG(x && y && n > 0, "udm_cast_f32_bf16: bad argument");
  UDM_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0), "udm_cast_f32_bf16: misaligned");
  const unsigned grid = (unsigned)((n / 4 + 256) / 256);
  if (scale == 1.f)
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, stream, x, (bf16_t*)y, (long)n);
  else
    hipLaunchKernelGGL(scale_cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, stream, x, (bf16_t*)y, (long)n, scale);
  UDM_CHECK_LAUNCH("udm_cast_f32_bf16");
  return 0;
}
extern "C" int udm_cast_bf16_f32(const void* x, float* y, int64_t n, float scale, hipStream_t stream) {
  UDM_CHECK_ARG(x && y && n > 0, "udm_cast_bf16_f32: bad argument");
  UDM_CHECK_ARG(((uintptr_t)y % 16 == 0) && ((uintptr_t)x % 8 == 0), "udm_cast_bf16_f32: misaligned");
  const unsigned grid = (unsigned)((n / 4 + 256) / 256);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, y, (long)n, scale);
  UDM_CHECK_LAUNCH("udm_cast_bf16_f32");
  return 0;
}
