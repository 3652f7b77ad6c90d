// Shared device helpers for the UniDisc MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes everywhere; bf16 is carried as raw uint16 (RNE conversions below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define UDM_LDS __attribute__((address_space(3)))

namespace udm {

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16 (round-to-nearest-even) with the gfx950 hardware converter: the compiler lowers a float->__bf16 vector
// conversion to one v_cvt_pk_bf16_f32 per PAIR of values (vs ~8 integer VALU ops per value in software).  Deliberately NOT
// inline asm: the hazard recogniser does not look inside asm blocks, and a cvt issued right behind the v_exp_f32 that
// produced its input (transcendental-result hazard) read stale data in the attention softmax.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// tanh-approximate GELU and its derivative (nn.GELU(approximate="tanh"), reference models/dit.py:918), written through the logistic
// function: 0.5 (1 + tanh u) = sigma(2u), so gelu(x) = x sigma(2u) and gelu'(x) = sigma (1 + 2 x u' (1 - sigma)) with
// u = k0 (x + k1 x^3).  One exp2 and one rcp per value and about half the VALU operations of the tanh form (these run inside the GEMM
// epilogues, 160 values per thread per tile).
__device__ __forceinline__ float gelu_tanh(float x) {
  const float c0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, c1 = c0 * 0.044715f;  // exp(-2u) = exp2(x (c0 + c1 x^2))
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(x * (c0 + c1 * x2));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float c0 = -2.0f * 1.4426950408889634f * k0, c1 = c0 * k1;
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(fminf(x * (c0 + c1 * x2), 80.0f));   // clamp: e s below must not become inf * 0 for x << 0
  const float s = __builtin_amdgcn_rcpf(1.0f + e);           // sigma(2u)
  const float du = k0 + 3.0f * k0 * k1 * x2;                 // u'
  return s + (2.0f * x * du) * (e * s) * s;                  // 1 - sigma = e sigma
}

// gelu(x) and gelu'(x) from ONE exp2 and ONE rcp (the forward epilogue of mlp.0 stores the derivative for the backward epilogue, which then
// is a single multiply): with s = sigma(2u), e = exp(-2u):  gelu = x s,  gelu' = s + (x s)(e s)(2 u'),  2u' = 2 k0 + 6 k0 k1 x^2.
__device__ __forceinline__ void gelu_tanh_both(float x, float& g, float& dg) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float c0 = -2.0f * 1.4426950408889634f * k0, c1 = c0 * k1;
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(fminf(x * (c0 + c1 * x2), 80.0f));
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  g = x * s;
  dg = s + (g * (e * s)) * (2.0f * k0 + 6.0f * k0 * k1 * x2);
}

// Philox4x32-10 counter RNG for dropout masks: (seed, 64-bit element-group counter) -> 4 uniform uint32.
__device__ __forceinline__ uint4 philox4x32(uint64_t seed, uint64_t ctr) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x5bd1e995u, c3 = 0x27d4eb2fu;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): give each XCD a contiguous
// chunk of the logical tile order so neighbouring tiles share one L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, loc = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}

}  // namespace udm

// Experiment switches for in-process A/B runs (udm_debug_set("exp", bits) or env UDM_EXP at load; 0 in production).  Bit meanings live next to the code they switch.
int udm_exp_flags();

// ---- host-side error plumbing (no exceptions cross the C ABI) -----------------------------------
void udm_set_error(const char* fmt, ...);
#define UDM_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      udm_set_error(__VA_ARGS__);           \
      return 2;                             \
    }                                       \
  } while (0)
#define UDM_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      udm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return 1;                                                                  \
    }                                                                            \
  } while (0)
