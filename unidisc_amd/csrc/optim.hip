// Fused AdamW with global-norm gradient clipping and bf16 weight shadows.
//
// Replaces torch.optim.AdamW(fused=True) (reference model_setup.py:385-424, betas/eps/weight_decay from config.optim) and
// accelerator.clip_grad_norm_ (model.py:1516-1520).  fp32 master weights, fp32 moments.  For the GEMM weights the SAME pass also writes
// the bf16 shadow W [out, in] and its transpose W^T [in, out] that the forward / dgrad GEMMs consume, so the per-forward weight cast that
// autocast performs in the reference (and cast_transpose performs here, ~3 % of a 1.4 B step) disappears from a training step.
//
// Arithmetic = torch's AdamW (single-tensor form, amsgrad off, maximize off), all in fp32:
//     g  <- g * clip,  clip = min(1, max_norm / (||g||_2 + 1e-6))            (clip_grad_norm_)
//     p  <- p * (1 - lr * wd)
//     m  <- m + (1 - b1) (g - m);   v <- b2 v + (1 - b2) g g
//     p  <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The global norm comes from udm_sumsq_f32 over the engine's flat gradient buffer and is read from device memory (no host sync).
#include "common.h"
#include "../../include/unidisc_hip.h"

#include <math.h>

namespace {
using namespace udm;

struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  float lr, b1, b2, eps, decay, bc1, rsqrt_bc2;   // decay = 1 - lr * wd; bc1 = 1 - b1^t; rsqrt_bc2 = 1 / sqrt(1 - b2^t)
  const float* gnorm_sq;                          // nullable: sum of squares of ALL gradients (device scalar)
  float max_norm;
  float* ema;                                     // nullable: exponential moving average of p (models/ema.py:44-53), updated in the same pass
  float ema_omd;                                  // 1 - decay of this update
};
__device__ __forceinline__ void ema4(const AdamArgs& a, long i4, const float4& p) {   // s <- s - (1 - decay) (s - p)
  float4 e = reinterpret_cast<float4*>(a.ema)[i4];
  e.x -= a.ema_omd * (e.x - p.x); e.y -= a.ema_omd * (e.y - p.y); e.z -= a.ema_omd * (e.z - p.z); e.w -= a.ema_omd * (e.w - p.w);
  reinterpret_cast<float4*>(a.ema)[i4] = e;
}

__device__ __forceinline__ float clip_coef(const AdamArgs& a) {
  if (!a.gnorm_sq) return 1.f;
  const float c = a.max_norm / (sqrtf(*a.gnorm_sq) + 1e-6f);
  return c < 1.f ? c : 1.f;
}
__device__ __forceinline__ float adam_update(const AdamArgs& a, float p, float g, float& m, float& v) {
  p *= a.decay;
  m = m + (1.f - a.b1) * (g - m);
  v = a.b2 * v + (1.f - a.b2) * g * g;
  const float denom = sqrtf(v) * a.rsqrt_bc2 + a.eps;
  return p - (a.lr / a.bc1) * (m / denom);
}

// flat tensors (norm / bias / embedding parameters): 16-byte accesses
__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a, long n) {
  const float clip = clip_coef(a);
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 p = reinterpret_cast<float4*>(a.p)[i], m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    p.x = adam_update(a, p.x, g.x * clip, m.x, v.x);
    p.y = adam_update(a, p.y, g.y * clip, m.y, v.y);
    p.z = adam_update(a, p.z, g.z * clip, m.z, v.z);
    p.w = adam_update(a, p.w, g.w * clip, m.w, v.w);
    reinterpret_cast<float4*>(a.p)[i] = p; reinterpret_cast<float4*>(a.m)[i] = m; reinterpret_cast<float4*>(a.v)[i] = v;
    if (a.ema) ema4(a, i, p);
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {  // tail (n not a multiple of 4)
    const long i = n4 * 4 + threadIdx.x;
    float m = a.m[i], v = a.v[i];
    const float pn = adam_update(a, a.p[i], a.g[i] * clip, m, v);
    a.p[i] = pn;
    a.m[i] = m; a.v[i] = v;
    if (a.ema) a.ema[i] -= a.ema_omd * (a.ema[i] - pn);
  }
}

// every flat parameter of a model in ONE launch (a 1.4 B DiT has ~250 norm / bias / embedding tensors: one launch each is ~250 x (launch + a wave of idle CUs)):
// device job table, chunk c of 1024 elements belongs to the job with chunk0 <= c < next chunk0 (bisection)
struct AdamJob { float* p; const float* g; float* m; float* v; float* ema; long n; long chunk0; };
constexpr int ADAM_CHUNK = 1024;   // elements per block iteration (256 threads x float4)
__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamArgs a, const AdamJob* __restrict__ jobs, int njobs, long nchunks) {
  const float clip = clip_coef(a);
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {   // last job with chunk0 <= c (block-uniform)
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].chunk0 <= c) lo = mid; else hi = mid - 1;
    }
    const AdamJob j = jobs[lo];
    const long i = (c - j.chunk0) * ADAM_CHUNK + threadIdx.x * 4;
    if (i + 3 < j.n) {
      float4 p = *reinterpret_cast<float4*>(j.p + i), m = *reinterpret_cast<float4*>(j.m + i), v = *reinterpret_cast<float4*>(j.v + i);
      const float4 g = *reinterpret_cast<const float4*>(j.g + i);
      p.x = adam_update(a, p.x, g.x * clip, m.x, v.x);
      p.y = adam_update(a, p.y, g.y * clip, m.y, v.y);
      p.z = adam_update(a, p.z, g.z * clip, m.z, v.z);
      p.w = adam_update(a, p.w, g.w * clip, m.w, v.w);
      *reinterpret_cast<float4*>(j.p + i) = p; *reinterpret_cast<float4*>(j.m + i) = m; *reinterpret_cast<float4*>(j.v + i) = v;
      if (j.ema) {
        float4 e = *reinterpret_cast<float4*>(j.ema + i);
        e.x -= a.ema_omd * (e.x - p.x); e.y -= a.ema_omd * (e.y - p.y); e.z -= a.ema_omd * (e.z - p.z); e.w -= a.ema_omd * (e.w - p.w);
        *reinterpret_cast<float4*>(j.ema + i) = e;
      }
    } else {
      for (long k = i; k < j.n && k < i + 4; ++k) {
        float m = j.m[k], v = j.v[k];
        const float pn = adam_update(a, j.p[k], j.g[k] * clip, m, v);
        j.p[k] = pn; j.m[k] = m; j.v[k] = v;
        if (j.ema) j.ema[k] -= a.ema_omd * (j.ema[k] - pn);
      }
    }
  }
}

// 2-D GEMM weight [R, C] row-major: update + bf16 shadow [R, ld16] + transposed bf16 shadow [C, ldt] through a 64 x 64 LDS tile
constexpr int TT = 64, TPAD = 8;
__device__ __forceinline__ void adamw_shadow_tile(const AdamArgs& a, int R, int C, bf16_t* __restrict__ w16, long ld16, bf16_t* __restrict__ w16t, long ldt, int r0, int c0,
                                                  bf16_t (*tile)[TT + TPAD]) {
  const float clip = clip_coef(a);
  const int tid = threadIdx.x;
  const int cs = (tid & 15) * 4;
  const bool vec = (C % 4 == 0);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = (tid >> 4) + pass * 16;
    bf16_t b[4] = {0, 0, 0, 0};
    if (r0 + r < R) {
      const long base = (long)(r0 + r) * C + c0 + cs;
      if (vec && c0 + cs + 3 < C) {
        float4 p = *reinterpret_cast<float4*>(a.p + base), m = *reinterpret_cast<float4*>(a.m + base), v = *reinterpret_cast<float4*>(a.v + base);
        const float4 g = *reinterpret_cast<const float4*>(a.g + base);
        p.x = adam_update(a, p.x, g.x * clip, m.x, v.x);
        p.y = adam_update(a, p.y, g.y * clip, m.y, v.y);
        p.z = adam_update(a, p.z, g.z * clip, m.z, v.z);
        p.w = adam_update(a, p.w, g.w * clip, m.w, v.w);
        *reinterpret_cast<float4*>(a.p + base) = p; *reinterpret_cast<float4*>(a.m + base) = m; *reinterpret_cast<float4*>(a.v + base) = v;
        if (a.ema) ema4(a, base / 4, p);
        b[0] = f2bf(p.x); b[1] = f2bf(p.y); b[2] = f2bf(p.z); b[3] = f2bf(p.w);
        if (w16) {
          bf16_t* op = w16 + (long)(r0 + r) * ld16 + c0 + cs;
          if (ld16 % 4 == 0) *reinterpret_cast<uint2*>(op) = make_uint2((uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16));
          else for (int k = 0; k < 4; ++k) op[k] = b[k];
        }
      } else {
        for (int k = 0; k < 4; ++k) {
          if (c0 + cs + k >= C) continue;
          float m = a.m[base + k], v = a.v[base + k];
          const float p = adam_update(a, a.p[base + k], a.g[base + k] * clip, m, v);
          a.p[base + k] = p; a.m[base + k] = m; a.v[base + k] = v;
          if (a.ema) a.ema[base + k] -= a.ema_omd * (a.ema[base + k] - p);
          b[k] = f2bf(p);
          if (w16) w16[(long)(r0 + r) * ld16 + c0 + cs + k] = b[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[cs + k][r] = b[k];
  }
  if (!w16t) return;
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = (tid >> 3) + pass * 32;
    const int rs = (tid & 7) * 8;
    if (c0 + c >= C || r0 + rs >= R) continue;
    bf16_t* op = w16t + (long)(c0 + c) * ldt + r0 + rs;
    if (r0 + rs + 7 < R && (ldt % 8 == 0)) {
      *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(&tile[c][rs]);
    } else {
      for (int k = 0; k < 8; ++k)
        if (r0 + rs + k < R) op[k] = tile[c][rs + k];
    }
  }
}
__global__ __launch_bounds__(256) void adamw_shadow_kernel(AdamArgs a, int R, int C, bf16_t* __restrict__ w16, long ld16, bf16_t* __restrict__ w16t, long ldt) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[TT][TT + TPAD];
  adamw_shadow_tile(a, R, C, w16, ld16, w16t, ldt, blockIdx.y * TT, blockIdx.x * TT, tile);
}
// every GEMM weight of a model in ONE launch (97 at 1.4 B: a launch each ramps up and drains the chip 97 times): device job table, block -> job by bisection on tile0
struct AdamShadowJob { float* p; const float* g; float* m; float* v; float* ema; bf16_t* w16; bf16_t* w16t; long ld16, ldt; int R, C, tile0, tiles_c; };
__global__ __launch_bounds__(256) void adamw_shadow_multi_kernel(AdamArgs a0, const AdamShadowJob* __restrict__ jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[TT][TT + TPAD];
  int lo = 0, hi = njobs - 1;
  const int t = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
  }
  const AdamShadowJob j = jobs[lo];
  AdamArgs a = a0;
  a.p = j.p; a.g = j.g; a.m = j.m; a.v = j.v; a.ema = j.ema;
  const int lt = t - j.tile0;
  adamw_shadow_tile(a, j.R, j.C, j.w16, j.ld16, j.w16t, j.ldt, (lt / j.tiles_c) * TT, (lt % j.tiles_c) * TT, tile);
}

// sum of squares, two phases (no deep atomic chains): per-block partials, then one block folds them
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) { const float t = x[n4 * 4 + threadIdx.x]; s += t * t; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

int fill_args(AdamArgs& a, const char* name, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay,
              int64_t step, const float* grad_norm_sq, float max_grad_norm) {
  UDM_CHECK_ARG(p && g && m && v, "%s: null pointer", name);
  UDM_CHECK_ARG(step >= 1 && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, "%s: bad hyper-parameters (step %ld)", name, (long)step);
  UDM_CHECK_ARG(!grad_norm_sq || max_grad_norm > 0.f, "%s: clipping needs max_grad_norm > 0", name);
  a.p = p; a.g = g; a.m = m; a.v = v;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  a.decay = 1.f - lr * weight_decay;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  a.gnorm_sq = grad_norm_sq; a.max_norm = max_grad_norm;
  a.ema = nullptr; a.ema_omd = 0.f;
  return 0;
}
int set_ema(AdamArgs& a, const char* name, float* ema, float ema_decay) {
  if (!ema) return 0;
  UDM_CHECK_ARG(ema_decay >= 0.f && ema_decay <= 1.f, "%s: EMA decay must be in [0, 1]", name);
  UDM_CHECK_ARG((uintptr_t)ema % 16 == 0, "%s: ema must be 16-byte aligned", name);
  a.ema = ema; a.ema_omd = 1.f - ema_decay;
  return 0;
}
}  // namespace

extern "C" int udm_sumsq_f32(const float* x, int64_t n, float* out, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(x && out && ws && n > 0, "udm_sumsq_f32: null pointer / empty");
  UDM_CHECK_ARG(((uintptr_t)x % 16 == 0), "udm_sumsq_f32: x must be 16-byte aligned");
  int grid = (int)((n / 4 + 255) / 256);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  UDM_CHECK_ARG(ws_elems >= grid, "udm_sumsq_f32: workspace needs %d floats", grid);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(256), 0, stream, x, (long)n, ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, grid, out);
  UDM_CHECK_LAUNCH("udm_sumsq_f32");
  return 0;
}

extern "C" int udm_adamw_step_ema(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  int64_t step, const float* grad_norm_sq, float max_grad_norm, float* ema, float ema_decay, hipStream_t stream) {
  AdamArgs a;
  if (int rc = fill_args(a, "udm_adamw_step", p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm)) return rc;
  if (int rc = set_ema(a, "udm_adamw_step", ema, ema_decay)) return rc;
  UDM_CHECK_ARG(n > 0, "udm_adamw_step: empty tensor");
  UDM_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "udm_adamw_step: tensors must be 16-byte aligned");
  long grid = (n / 4 + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a, (long)n);
  UDM_CHECK_LAUNCH("udm_adamw_step");
  return 0;
}

// jobs: device array of njobs records {p, g, m, v, ema (nullable), n, chunk0} (7 x 8 bytes; chunk0 = sum over the jobs before of ceil(n / 1024)), nchunks = the total
extern "C" int udm_adamw_step_multi(const void* jobs, int64_t njobs, int64_t nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                    const float* grad_norm_sq, float max_grad_norm, float ema_decay, hipStream_t stream) {
  UDM_CHECK_ARG(jobs && njobs > 0 && nchunks > 0, "udm_adamw_step_multi: empty job table");
  AdamArgs a;
  float dummy = 0.f;
  if (int rc = fill_args(a, "udm_adamw_step_multi", &dummy, &dummy, &dummy, &dummy, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm)) return rc;
  UDM_CHECK_ARG(ema_decay >= 0.f && ema_decay <= 1.f, "udm_adamw_step_multi: EMA decay must be in [0, 1]");
  a.p = nullptr; a.g = nullptr; a.m = nullptr; a.v = nullptr;
  a.ema_omd = 1.f - ema_decay;
  const long grid = nchunks < 4096 ? nchunks : 4096;
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)grid), dim3(256), 0, stream, a, (const AdamJob*)jobs, (int)njobs, (long)nchunks);
  UDM_CHECK_LAUNCH("udm_adamw_step_multi");
  return 0;
}

// jobs: device array of records {p, g, m, v, ema (nullable), w16 (nullable), w16t (nullable), ld16, ldt, R, C, tile0, tiles_c} (see AdamShadowJob), ntiles = total 64 x 64 tiles
extern "C" int udm_adamw_step_shadow_multi(const void* jobs, int64_t njobs, int64_t ntiles, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                           const float* grad_norm_sq, float max_grad_norm, float ema_decay, hipStream_t stream) {
  UDM_CHECK_ARG(jobs && njobs > 0 && ntiles > 0 && ntiles < (1ll << 31), "udm_adamw_step_shadow_multi: empty / oversized job table");
  AdamArgs a;
  float dummy = 0.f;
  if (int rc = fill_args(a, "udm_adamw_step_shadow_multi", &dummy, &dummy, &dummy, &dummy, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm)) return rc;
  UDM_CHECK_ARG(ema_decay >= 0.f && ema_decay <= 1.f, "udm_adamw_step_shadow_multi: EMA decay must be in [0, 1]");
  a.p = nullptr; a.g = nullptr; a.m = nullptr; a.v = nullptr;
  a.ema_omd = 1.f - ema_decay;
  hipLaunchKernelGGL(adamw_shadow_multi_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, a, (const AdamShadowJob*)jobs, (int)njobs);
  UDM_CHECK_LAUNCH("udm_adamw_step_shadow_multi");
  return 0;
}

extern "C" int udm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int64_t step, const float* grad_norm_sq, float max_grad_norm, hipStream_t stream) {
  return udm_adamw_step_ema(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm, nullptr, 0.f, stream);
}

extern "C" int udm_adamw_step_shadow_ema(float* p, const float* g, float* m, float* v, int64_t R, int64_t C, float lr, float beta1, float beta2, float eps,
                                         float weight_decay, int64_t step, const float* grad_norm_sq, float max_grad_norm, void* w16, int64_t ld16, void* w16t,
                                         int64_t ldt, float* ema, float ema_decay, hipStream_t stream) {
  AdamArgs a;
  if (int rc = fill_args(a, "udm_adamw_step_shadow", p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm)) return rc;
  if (int rc = set_ema(a, "udm_adamw_step_shadow", ema, ema_decay)) return rc;
  UDM_CHECK_ARG(R > 0 && C > 0 && (w16 || w16t), "udm_adamw_step_shadow: bad shape / no shadow requested");
  UDM_CHECK_ARG((!w16 || ld16 >= C) && (!w16t || ldt >= R), "udm_adamw_step_shadow: shadow strides too small");
  UDM_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "udm_adamw_step_shadow: tensors must be 16-byte aligned");
  dim3 grid((unsigned)((C + TT - 1) / TT), (unsigned)((R + TT - 1) / TT));
  hipLaunchKernelGGL(adamw_shadow_kernel, grid, dim3(256), 0, stream, a, (int)R, (int)C, (bf16_t*)w16, (long)ld16, (bf16_t*)w16t, (long)ldt);
  UDM_CHECK_LAUNCH("udm_adamw_step_shadow");
  return 0;
}

extern "C" int udm_adamw_step_shadow(float* p, const float* g, float* m, float* v, int64_t R, int64_t C, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, int64_t step, const float* grad_norm_sq, float max_grad_norm, void* w16, int64_t ld16, void* w16t,
                                     int64_t ldt, hipStream_t stream) {
  return udm_adamw_step_shadow_ema(p, g, m, v, R, C, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm, w16, ld16, w16t, ldt, nullptr, 0.f,
                                   stream);
}
