// fp8 (OCP e4m3) quantisation helpers shared by the fused qk-norm + rope kernel (rowops.hip) and the attention fp8 path (attention_fp8.hip).
//
// Scales are powers of two carried as E8M0 bytes (value 2^(e - 127)) - the block-scale format of v_mfma_scale_f32_32x32x64_f8f6f4, which applies
// them in hardware.  e4m3 is a floating-point format: a power-of-two scale costs no precision (only which values fall into the subnormal range), so
// the scale is simply the smallest 2^k with amax * 2^-k <= 448.
#pragma once
#include "common.h"

namespace udm {

// E8M0 byte of the smallest power of two 2^k with amax * 2^-k <= 448 = 1.75 * 2^8 (k clamped to [-126, 126]; amax = 0 -> k = 0)
__device__ __forceinline__ int e8m0_for_amax(float amax) {
  if (!(amax > 0.f)) return 127;
  const unsigned u = __float_as_uint(amax);
  int k = (int)((u >> 23) & 0xffu) - 127 - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
  k = k < -126 ? -126 : (k > 126 ? 126 : k);
  return k + 127;
}
__device__ __forceinline__ float e8m0_inv_scale(int e8) { return __uint_as_float((unsigned)(254 - e8) << 23); }   // 2^-(e8 - 127)
__device__ __forceinline__ float e8m0_scale(int e8) { return __uint_as_float((unsigned)e8 << 23); }               // 2^(e8 - 127)

// 8 floats -> 8 e4m3 bytes (round to nearest even: v_cvt_pk_fp8_f32), low byte first
__device__ __forceinline__ uint2 pack8_e4m3(const float* p) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[4], p[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[6], p[7], hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}
// ... and back (exact)
__device__ __forceinline__ void unpack8_e4m3(uint2 u, float* p) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.x, true);
  f2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.y, true);
  p[0] = a[0]; p[1] = a[1]; p[2] = b[0]; p[3] = b[1]; p[4] = c[0]; p[5] = c[1]; p[6] = d[0]; p[7] = d[1];
}
// quantise 8 values with the scale of `e8` and return both the bytes and the dequantised values (what the MFMA will see: exactly bf16-representable)
__device__ __forceinline__ uint2 quant8_e4m3(float* v, int e8) {
  const float inv = e8m0_inv_scale(e8), sc = e8m0_scale(e8);
  float t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = v[j] * inv;
  const uint2 r = pack8_e4m3(t);
  unpack8_e4m3(r, t);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = t[j] * sc;
  return r;
}

}  // namespace udm
