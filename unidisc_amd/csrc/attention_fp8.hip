// fp8 (OCP e4m3) attention forward on gfx950: S^T = K Q^T and O^T = V^T P^T through v_mfma_f32_32x32x16_fp8_fp8, fp32 softmax.
//
// BASELINE config E ("interleaved long context, fp8 MFMA attention path"); the reference has no fp8 path (the parity
// target is this repository's own bf16 kernel, attention.hip, under a stated looser tolerance).  The kernel keeps the bf16 kernel's
// structure - scores computed transposed so that a lane owns one query column, lane-local online softmax, P^T fed to the second MFMA
// straight from registers - because the fp8 32x32x16 MFMA has the same operand geometry as the bf16 one (8 values per lane: k-block
// lane >> 5, row lane & 31), with HALF the bytes per fragment.  That is the point here: the bf16 forward is bound by LDS fragment traffic
// (1 KiB per MFMA and wave), not by the matrix pipe; on gfx950 this MFMA runs at the bf16 rate, the doubled fp8 rate needs the
// 32x32x64 f8f6f4 form.
//
// Operands are prepared by udm_attention_quantize_fp8 (one HBM-bound pass): per-tensor scales amax / 448 for q, k and v; q8, k8 row-major
// like their sources; v8t TRANSPOSED per head, [B*H][D][Lp] with Lp = ceil(L / 64) * 64 (zero padded), the keys of every 16-key chunk
// stored in the order 0-3, 8-11, 4-7, 12-15 - the order in which a lane's accumulator registers walk the keys - so that the V^T fragment
// of a lane is one 8-byte LDS read.  P is scaled by 2^8 before the conversion (softmax values below 2^-9 would otherwise flush to zero;
// the running maximum is exact here, not lazy, so p <= 1 and 256 p <= 256 < 448); the row sum uses the unquantised p.
#include "attention_common.h"

#include <algorithm>

namespace {

constexpr int BQ8 = 128, BKV8 = 64;
constexpr float P_SCALE = 256.0f;

struct Fp8Args {
  const uint8_t* q8; const uint8_t* k8; const uint8_t* v8t;   // q8/k8: [B*L][H*D]; v8t: [B*H][D][Lp]
  const float* scales;                                         // {sq, sk, sv}: x = x8 * s
  bf16_t* out; float* lse;
  const int64_t* sample_ids; const int* doc_ranges;
  long out_stride;
  int B, H, L, Lp;
  float scale_log2;
};

__device__ __forceinline__ long pack8_fp8(const float* p) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[4], p[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[6], p[7], hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int D, bool HAS_SID>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(Fp8Args a) {
  constexpr int KS = D / 16, DB = D / 32;
  constexpr int KT = BKV8 * D, VT = D * BKV8;          // bytes per K8 / V8T tile
  // Byte geometry of the tiles = bf16 tiles of half the width, so the LDS-DMA stager and the XOR swizzle of the bf16 kernels are reused as is:
  // K8 [64 keys][D bytes] = [64][D/2 bf16], V8T [D rows][64 bytes] = [D][32 bf16].  A fragment is the 8-byte half `hi` of a 16-byte slot.
  using StgK = DmaStager<D / 2, BKV8>;
  using StgV = DmaStager<32, D>;
  __shared__ __attribute__((aligned(16))) char smem[2 * (KT + VT)];   // K0 | K1 | V0 | V1
  __shared__ long sid_s[2][BKV8];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);   // tile-major 1-D grid: one (b,h) per XCD (see attention.hip)
  const int b = bh / a.H, h = bh % a.H;
  const int qi = tile_x * BQ8 + wave * 32 + l31;
  const bool q_ok = qi < a.L;
  const long rowbase = (long)b * a.L;
  const int d = a.H * D;

  long qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = q_ok ? *reinterpret_cast<const long*>(a.q8 + (rowbase + qi) * d + h * D + ks * 16 + hi * 8) : 0L;
  long sid_q = 0;
  if (HAS_SID) sid_q = q_ok ? a.sample_ids[rowbase + qi] : -1;
  float c = a.scale_log2 * a.scales[0] * a.scales[1];
  float sv = a.scales[2];
  // (as in the bf16 kernel: the waits for these global loads must land before the loop, whose only vector-memory traffic is the inline-asm DMA)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  asm volatile("" : "+v"(c), "+v"(sv));
  if (HAS_SID) asm volatile("" : "+v"(sid_q));

  f32x16_t oT[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  const bf16_t* kbase = reinterpret_cast<const bf16_t*>(a.k8 + rowbase * d + h * D);     // row stride d bytes = d / 2 "bf16"
  const uint8_t* vbase = a.v8t + (long)bh * D * a.Lp;                                     // row stride Lp bytes
  const long kstride = d / 2, vstride = a.Lp / 2;
  const int nkv = (a.L + BKV8 - 1) / BKV8;
  int t_begin = 0, t_end = nkv, blk_id = -1;
  if (HAS_SID) { const DocSpan sp = doc_tile_span(a.doc_ranges, b, a.L, tile_x, nkv); t_begin = sp.t_begin; t_end = sp.t_end; blk_id = sp.blk_id; }
  if (t_begin < t_end) {
    StgK::issue(kbase, kstride, t_begin * BKV8, a.L, smem + (t_begin & 1) * KT, wave, lane);
    StgV::issue(reinterpret_cast<const bf16_t*>(vbase + t_begin * BKV8), vstride, 0, D, smem + 2 * KT + (t_begin & 1) * VT, wave, lane);
  }
  for (int t = t_begin; t < t_end; ++t) {
    const int kv0 = t * BKV8, st = t & 1;
    const char* Ks = smem + st * KT;
    const char* Vs = smem + 2 * KT + st * VT;
    const bool id_test = HAS_SID && doc_pair_needs_mask(a.doc_ranges, b, a.L, t, blk_id);
    if (HAS_SID && tid < BKV8) sid_s[st][tid] = (kv0 + tid < a.L) ? a.sample_ids[rowbase + kv0 + tid] : -2;
    wait_all_vmem();   // this wave's share of tile t has landed
    __syncthreads();   // ... and everybody's; all waves are done with tile t-1, so its stage may be refilled
    if (t + 1 < t_end) {
      StgK::issue(kbase, kstride, kv0 + BKV8, a.L, smem + (st ^ 1) * KT, wave, lane);
      StgV::issue(reinterpret_cast<const bf16_t*>(vbase + kv0 + BKV8), vstride, 0, D, smem + 2 * KT + (st ^ 1) * VT, wave, lane);
    }

    f32x16_t sT[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[f][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const long kf = *reinterpret_cast<const long*>(Ks + tile_off<D / 2>(f * 32 + l31, ks) + hi * 8);
        sT[f] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[ks], sT[f], 0, 0, 0);
      }
    if (id_test || kv0 + BKV8 > a.L) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          bool ok = kv0 + kl < a.L;
          if (HAS_SID) ok = ok && (!id_test || ((sid_s[st][kl] == sid_q) && (sid_q >= 0)));
          if (!ok) sT[f][r] = -INFINITY;
        }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sT[f][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // exact running maximum (p <= 1, see the header); c > 0
    if (__builtin_amdgcn_ballot_w64(q_ok && mloc > m) != 0) {
      const float m_new = fmaxf(m, mloc);
      const float alpha = __builtin_amdgcn_exp2f((m - ((m_new == -INFINITY) ? 0.f : m_new)) * c);
      lsum *= alpha;
      m = m_new;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[i][r] *= alpha;
    }
    const float mc = (m == -INFINITY) ? 0.f : m * c;
    float psum = 0.f;
    long pb[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      float pq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = __builtin_amdgcn_exp2f(sT[cc >> 1][8 * (cc & 1) + j] * c - mc);
        psum += pv;
        pq[j] = pv * P_SCALE;
      }
      pb[cc] = pack8_fp8(pq);
    }
    lsum += psum;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int i = 0; i < DB; ++i) {
        const long vt = *reinterpret_cast<const long*>(Vs + tile_off<32>(i * 32 + l31, cc) + hi * 8);
        oT[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vt, pb[cc], oT[i], 0, 0, 0);
      }
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = ltot > 0.f ? sv / (ltot * P_SCALE) : 0.f;
  if (q_ok) {
    bf16_t* op = a.out + (rowbase + qi) * a.out_stride + h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = i * 32 + 8 * rg + 4 * hi;
        *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(oT[i][rg * 4] * inv, oT[i][rg * 4 + 1] * inv), pack2bf(oT[i][rg * 4 + 2] * inv, oT[i][rg * 4 + 3] * inv));
      }
    if (hi == 0) a.lse[((long)b * a.H + h) * a.L + qi] = ltot > 0.f ? m * c + log2f(ltot) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------ quantisation
// amax[0..2] (fp32 bit patterns of non-negative values order like unsigned integers) of q, k, v
__global__ __launch_bounds__(256) void fp8_amax_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long qs, long ks,
                                                      long vs, long M, int d, unsigned* __restrict__ amax) {
  const int per_row = d / 8;
  float mx[3] = {0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M * per_row; i += (long)gridDim.x * 256) {
    const long row = i / per_row;
    const int c8 = (int)(i % per_row) * 8;
    const bf16_t* src[3] = {q + row * qs + c8, k + row * ks + c8, v + row * vs + c8};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const uint4 u = *reinterpret_cast<const uint4*>(src[s]);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) mx[s] = fmaxf(mx[s], fmaxf(fabsf(__uint_as_float(w[j] << 16)), fabsf(__uint_as_float(w[j] & 0xffff0000u))));
    }
  }
  __shared__ float red[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float r = wave_max(mx[s]);
    if ((threadIdx.x & 63) == 0) red[s][threadIdx.x >> 6] = r;
  }
  __syncthreads();
  if (threadIdx.x < 3) {   // one atomic per block and tensor: tens of thousands of same-address atomics cost more than the pass itself
    const float r = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
    atomicMax(amax + threadIdx.x, __float_as_uint(r));
  }
}

__global__ void fp8_scales_kernel(const unsigned* __restrict__ amax, float* __restrict__ scales) {
  if (threadIdx.x < 3) {
    const float a = __uint_as_float(amax[threadIdx.x]);
    scales[threadIdx.x] = a > 0.f ? a / 448.0f : 1.0f;
  }
}

__device__ __forceinline__ uint2 quant8(uint4 u, float inv) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
  float f[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(w[j] << 16) * inv; f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u) * inv; }
  const long r = pack8_fp8(f);
  return make_uint2((unsigned)(unsigned long)r, (unsigned)((unsigned long)r >> 32));
}

// q8, k8: same [M][d] layout as the sources (row stride d bytes)
__global__ __launch_bounds__(256) void fp8_quant_qk_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, long qs, long ks, long M, int d,
                                                          const float* __restrict__ scales, uint8_t* __restrict__ q8, uint8_t* __restrict__ k8) {
  const int per_row = d / 8;
  const float iq = 1.0f / scales[0], ik = 1.0f / scales[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M * per_row; i += (long)gridDim.x * 256) {
    const long row = i / per_row;
    const int c8 = (int)(i % per_row) * 8;
    *reinterpret_cast<uint2*>(q8 + row * d + c8) = quant8(*reinterpret_cast<const uint4*>(q + row * qs + c8), iq);
    *reinterpret_cast<uint2*>(k8 + row * d + c8) = quant8(*reinterpret_cast<const uint4*>(k + row * ks + c8), ik);
  }
}

// v8t[bh][dd][Lp]: block = (bh, 64-key tile); keys of each 16-chunk stored as 0-3, 8-11, 4-7, 12-15 (groups of 4 stay together); keys past L are
// zero.  A thread takes 4 consecutive keys x 8 head dims, quantises, transposes the 4 x 8 bytes in registers and writes 8 dwords (4 keys of one dim).
template <int D>
__global__ __launch_bounds__(256) void fp8_quant_vt_kernel(const bf16_t* __restrict__ v, long vs, int B, int H, int L, int Lp, const float* __restrict__ scales,
                                                          uint8_t* __restrict__ v8t) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[D][64 + 4];
  const int ntile = Lp / 64;
  const int bh = blockIdx.x / ntile, t = blockIdx.x % ntile;
  const int b = bh / H, h = bh % H;
  const float iv = 1.0f / scales[2];
  constexpr int VPR = D / 8;   // 16-byte vectors per key row
  for (int i = threadIdx.x; i < 16 * VPR; i += 256) {
    const int kg = i / VPR, c8 = (i % VPR) * 8;   // key group (4 keys), first head dim
    uint32_t w[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int l = t * 64 + kg * 4 + kk;
      uint2 r = make_uint2(0u, 0u);
      if (l < L) r = quant8(*reinterpret_cast<const uint4*>(v + ((long)b * L + l) * vs + h * D + c8), iv);
      w[kk][0] = r.x; w[kk][1] = r.y;
    }
    const int g = kg & 3;                                              // group inside its 16-key chunk: 0, 1, 2, 3 -> stored 0, 2, 1, 3
    const int pos = (kg & ~3) * 4 + (((g & 1) << 1) | (g >> 1)) * 4;   // byte position of the group inside the 64-key row
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sh = 8 * (e & 3);
      const uint32_t o = ((w[0][e >> 2] >> sh) & 0xffu) | (((w[1][e >> 2] >> sh) & 0xffu) << 8) | (((w[2][e >> 2] >> sh) & 0xffu) << 16) | (((w[3][e >> 2] >> sh) & 0xffu) << 24);
      *reinterpret_cast<uint32_t*>(&tile[c8 + e][pos]) = o;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D * 4; i += 256) {
    const int dd = i / 4, w16 = (i % 4) * 16;
    uint4 o;
    o.x = *reinterpret_cast<const uint32_t*>(&tile[dd][w16]); o.y = *reinterpret_cast<const uint32_t*>(&tile[dd][w16 + 4]);
    o.z = *reinterpret_cast<const uint32_t*>(&tile[dd][w16 + 8]); o.w = *reinterpret_cast<const uint32_t*>(&tile[dd][w16 + 12]);
    *reinterpret_cast<uint4*>(v8t + ((long)bh * D + dd) * Lp + t * 64 + w16) = o;
  }
}

}  // namespace

extern "C" int udm_attention_quantize_fp8(const void* q, const void* k, const void* v, void* q8, void* k8, void* v8t, float* scales, uint32_t* amax_ws, int64_t B,
                                          int64_t H, int64_t L, int64_t D, int64_t q_stride, int64_t k_stride, int64_t v_stride, hipStream_t stream) {
  UDM_CHECK_ARG(q && k && v && q8 && k8 && v8t && scales && amax_ws, "udm_attention_quantize_fp8: null pointer");
  UDM_CHECK_ARG(B > 0 && H > 0 && L > 0 && (D == 64 || D == 128), "udm_attention_quantize_fp8: bad shape (head_dim 64 or 128)");
  UDM_CHECK_ARG(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0, "udm_attention_quantize_fp8: row strides must be multiples of 8 elements");
  const long M = B * L;
  const int d = (int)(H * D), Lp = (int)((L + 63) / 64 * 64);
  if (hipMemsetAsync(amax_ws, 0, 3 * sizeof(uint32_t), stream) != hipSuccess) { udm_set_error("udm_attention_quantize_fp8: memset failed"); return 1; }
  const int grid = (int)std::min<long>((M * (d / 8) + 255) / 256, 2048);
  hipLaunchKernelGGL(fp8_amax_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (long)q_stride, (long)k_stride,
                     (long)v_stride, M, d, amax_ws);
  hipLaunchKernelGGL(fp8_scales_kernel, dim3(1), dim3(64), 0, stream, amax_ws, scales);
  hipLaunchKernelGGL(fp8_quant_qk_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (long)q_stride, (long)k_stride, M, d, scales,
                     (uint8_t*)q8, (uint8_t*)k8);
  const dim3 gv((unsigned)(B * H * (Lp / 64)));
  if (D == 128) hipLaunchKernelGGL(fp8_quant_vt_kernel<128>, gv, dim3(256), 0, stream, (const bf16_t*)v, (long)v_stride, (int)B, (int)H, (int)L, Lp, scales, (uint8_t*)v8t);
  else hipLaunchKernelGGL(fp8_quant_vt_kernel<64>, gv, dim3(256), 0, stream, (const bf16_t*)v, (long)v_stride, (int)B, (int)H, (int)L, Lp, scales, (uint8_t*)v8t);
  UDM_CHECK_LAUNCH("udm_attention_quantize_fp8");
  return 0;
}

extern "C" int udm_attention_fwd_fp8(const void* q8, const void* k8, const void* v8t, const float* scales, void* o, float* lse, const int64_t* sample_ids,
                                     const int32_t* doc_ranges, int64_t B, int64_t H, int64_t L, int64_t D, int64_t o_stride, hipStream_t stream) {
  UDM_CHECK_ARG(q8 && k8 && v8t && scales && o && lse, "udm_attention_fwd_fp8: null pointer");
  UDM_CHECK_ARG(B > 0 && H > 0 && L > 0 && (D == 64 || D == 128), "udm_attention_fwd_fp8: bad shape (head_dim 64 or 128)");
  UDM_CHECK_ARG(o_stride % 4 == 0 && (sample_ids || !doc_ranges), "udm_attention_fwd_fp8: bad o_stride / doc_ranges without sample_ids");
  UDM_CHECK_ARG(B * H * ((L + 127) / 128) < (1LL << 31), "udm_attention_fwd_fp8: grid too large");
  Fp8Args a{};
  a.q8 = (const uint8_t*)q8; a.k8 = (const uint8_t*)k8; a.v8t = (const uint8_t*)v8t; a.scales = scales; a.out = (bf16_t*)o; a.lse = lse;
  a.sample_ids = sample_ids; a.doc_ranges = doc_ranges; a.out_stride = o_stride;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.Lp = (int)((L + 63) / 64 * 64);
  a.scale_log2 = 1.4426950408889634f / sqrtf((float)D);
  const dim3 grid((unsigned)(((L + BQ8 - 1) / BQ8) * H * B));
  if (D == 128) {
    if (sample_ids) hipLaunchKernelGGL((attn_fwd_fp8_kernel<128, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_fp8_kernel<128, false>), grid, dim3(256), 0, stream, a);
  } else {
    if (sample_ids) hipLaunchKernelGGL((attn_fwd_fp8_kernel<64, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_fp8_kernel<64, false>), grid, dim3(256), 0, stream, a);
  }
  UDM_CHECK_LAUNCH("udm_attention_fwd_fp8");
  return 0;
}
