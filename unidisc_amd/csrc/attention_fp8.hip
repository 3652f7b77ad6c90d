// fp8 (OCP e4m3) attention forward on gfx950 through the block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 matrix rate).
//
// BASELINE config E ("interleaved long context, fp8 MFMA attention path"); the reference has no fp8 path: the parity target is this
// repository's own bf16 kernel (attention.hip) under a stated tolerance.  Structure of the bf16 forward - scores computed transposed (S^T = K Q^T) so that
// a lane owns one query column, lane-local online softmax, P^T fed to the second matrix product straight from registers - with these differences:
//
//   * One MFMA contracts 64 elements: S^T of a 64-key tile is 2 x (D / 64) instructions, O^T += V^T P^T is D / 32 instructions over ALL 64 keys of the
//     tile (the lane's 32 score registers, converted to e4m3, ARE the 32-byte B operand: byte f*16 + r = key f*32 + (r&3) + 8 (r>>2) + 4 hi).
//   * Scales are powers of two (E8M0 bytes) applied by the instruction itself: one per (row, head) for q and k (written by the qk-norm + rope kernel,
//     udm_qknorm_rope_fwd_fp8, which also leaves the DEQUANTISED values in the bf16 `qkr` the backward reads), one per 64-key tile and head for v, and
//     the constant 2^-8 that undoes the 2^8 the probabilities are scaled by before their conversion (p <= 2^0.75: the reference exponent is lazy within e4m3's head-room).
//     No scale arithmetic in the softmax.
//   * Operands: qk8 [B*L][2d] bytes (q | k), K tiles staged by LDS-DMA as [64 keys][D bytes]; v8t [B*H][Lp/64][D][64] bytes - per tile V^T with the
//     keys of a row in the order the P operand holds them (udm_attention_quantize_v_fp8) - one contiguous 64 D-byte piece per tile.
//   * Backward rule (dit.py): the bf16 backward kernels run on the dequantised q, k (bit-identical scores: products of e4m3 values are exact in the
//     fp32 accumulators of either instruction), the forward's log-sum-exp, and v.
#include "attention_common.h"
#include "fp8_common.h"

#include <algorithm>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;   // 32 bytes: A / B operand of the 32x32x64 f8f6f4 MFMA

constexpr int BQ8 = 128, BKV8 = 64;
constexpr int P_E8 = 127 - 8;   // probabilities are converted as 2^8 p

struct Fp8Args {
  const uint8_t* qk8;       // [B*L][2 d]: q bytes | k bytes
  const uint8_t* qk_e8;     // [B*L][2 Hp], Hp = H rounded up to 4: E8M0 scales of the q heads | of the k heads
  const uint8_t* v8t;       // [B*H][Lp/64][D][64]
  const int* v_e8;          // [B*H][Lp/64]
  bf16_t* out; float* lse;
  const int64_t* sample_ids; const int* doc_ranges;
  long out_stride;
  int B, H, L;
  float scale_log2;
};

__device__ __forceinline__ i32x8_t lds_frag32(const char* base, int off0, int off1) {
  const uint4 a = *reinterpret_cast<const uint4*>(base + off0), b = *reinterpret_cast<const uint4*>(base + off1);
  i32x8_t r = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
  return r;
}

// A key tile costs half the matrix time of the bf16 kernel's, so one tile of look-ahead no longer covers the memory latency: the stages form a RING of
// NST = 4 tiles (K8 | V8T | key scales: 16.25 KB per stage at D = 128 - the bf16 kernel's LDS footprint), refills run three tiles ahead and the top of a
// tile waits with a COUNTED vmcnt for its own pieces only.  That needs every vector-memory instruction of the loop to be one of the inline-asm pieces (the
// compiler's own waits would drain the ring): the per-key E8M0 scales therefore travel through LDS as well (one 4-byte piece per key: the dword of the row's
// scale bytes that holds this head's), the V scale of a tile is a scalar load.
template <int D, bool HAS_SID>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(Fp8Args a) {
  constexpr int KC = D / 64, DB = D / 32, NST = 4;
  constexpr int KT = BKV8 * D, VT = D * BKV8, ST_BYTES = KT + VT + 512;          // bytes per K8 / V8T tile, per stage (+ 4 x 32 scale dwords)
  constexpr int PIECES = 2 * (D / 64) + 1;                                       // LDS-DMA instructions per wave and tile
  // Byte geometry = bf16 tiles of half the width, so the LDS-DMA stager and the XOR swizzle of the bf16 kernels are reused as is:
  // K8 [64 keys][D bytes] = [64][D/2 bf16], V8T [D rows][64 bytes] = [D][32 bf16].
  using StgK = DmaStager<D / 2, BKV8>;
  __shared__ __attribute__((aligned(16))) char smem[NST * ST_BYTES + 2 * BKV8 * 8];   // NST x (K | V | scales) | sidk[2][64]
  long* sid_s = reinterpret_cast<long*>(smem + NST * ST_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);   // all tiles of one (b, h) on one XCD (see attention_common.h)
  const int b = bh / a.H, h = bh % a.H;
  const int qi = tile_x * BQ8 + wave * 32 + l31;
  const bool q_ok = qi < a.L;
  const long rowbase = (long)b * a.L;
  const int d = a.H * D, H2 = 2 * ((a.H + 3) & ~3);

  // Q^T operand: bytes [c*64 + hi*32, +32) of the lane's query row; its E8M0 scale
  i32x8_t qf[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const uint8_t* p = a.qk8 + (rowbase + qi) * (2L * d) + h * D + c * 64 + hi * 32;
    const uint4 u0 = q_ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0), u1 = q_ok ? *reinterpret_cast<const uint4*>(p + 16) : make_uint4(0, 0, 0, 0);
    qf[c] = i32x8_t{(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w, (int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
  }
  int qse = q_ok ? (int)a.qk_e8[(rowbase + qi) * H2 + h] : 127;
  // (as in the bf16 kernel: the waits for these global loads must land before the loop, whose DMA refills are inline asm the compiler does not count)
#pragma unroll
  for (int c = 0; c < KC; ++c) asm volatile("" : "+v"(qf[c]));
  asm volatile("" : "+v"(qse));

  f32x16_t oT[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;     // lsum accumulates 2^8 p
  const float c = a.scale_log2;

  const int nkv = (a.L + BKV8 - 1) / BKV8;
  const bf16_t* kbase = reinterpret_cast<const bf16_t*>(a.qk8 + rowbase * (2L * d) + d + h * D);     // row stride 2 d bytes = d "bf16"
  const uint8_t* vbase = a.v8t + (long)bh * nkv * VT;
  const uint8_t* kebase = a.qk_e8 + rowbase * H2 + H2 / 2 + (h & ~3);   // the dword of a row's k-scale bytes that holds head h
  const int ksh = (h & 3) * 8;
  const int* vebase = a.v_e8 + (long)bh * nkv;
  const long kstride = d;
  DmaPlan<D / 2, BKV8> plank;
  DmaPlan<32, D> planv;
  plank.init(kstride, wave, lane);
  planv.init(32, wave, lane);
  int t_begin = 0, t_end = nkv, blk_id = -1;
  bool doc_pure = false;
  if (HAS_SID) { const DocSpan sp = doc_tile_span(a.doc_ranges, b, a.L, tile_x, nkv); t_begin = sp.t_begin; t_end = sp.t_end; blk_id = sp.blk_id;
                 doc_pure = sp.pure && sp.lo % BKV8 == 0 && (sp.hi % BKV8 == 0 || sp.hi == a.L); }
  // all PIECES of key tile t into ring stage t % NST (every wave: its share of K8 and V8T, and the scale dwords of 16 keys)
  auto issue_tile = [&](int t) {
    char* stg = smem + (t & (NST - 1)) * ST_BYTES;
    const int kv0 = t * BKV8;
    if (kv0 + BKV8 <= a.L) plank.issue_full(kbase + (long)kv0 * kstride, stg, wave);
    else StgK::issue(kbase, kstride, kv0, a.L, stg, wave, lane);
    planv.issue_full(reinterpret_cast<const bf16_t*>(vbase + (long)t * VT), stg + KT, wave);
    // scale dwords: wave w, lane i < 16 -> the dword of key w*16 + i's k-scale bytes that holds head h, at dword w*32 + i; lane 16 -> this tile's V scale
    // (every wave fetches it into its own dword w*32 + 16; dword 16 is the one read).  Through LDS, not a scalar load: the compiler will not take a
    // loop-variant load from a pointer that may alias the kernel's stores through the scalar cache, and a vector load of its own would drain the ring.
    if (lane <= 16) dma4_asm(lane < 16 ? (const void*)(kebase + (long)min(kv0 + wave * 16 + lane, a.L - 1) * H2) : (const void*)(vebase + t), stg + KT + VT + wave * 128);
  };
  for (int j = 0; j < NST - 1; ++j)
    if (t_begin + j < t_end) issue_tile(t_begin + j);
  // one key tile; STC >= 0: the ring stage as a compile-time constant (fragment addresses = hoisted register + immediate), -1: taken from t (the few
  // tiles before / after the 4-tile steady state); IDS = false is the walk of a block whose whole key span is its own document (no sample-id code at all)
  auto tile = [&](auto st_c, int t, auto ids_c) {
    constexpr int STC = decltype(st_c)::value;
    constexpr bool IDS = HAS_SID && decltype(ids_c)::value;
    const int ST = STC >= 0 ? STC : (t & (NST - 1));
    const int kv0 = t * BKV8;
    const char* Ks = smem + ST * ST_BYTES;
    const char* Vs = Ks + KT;
    const unsigned* Es = reinterpret_cast<const unsigned*>(Ks + KT + VT);
    const long* sidk = sid_s + (t & 1) * BKV8;
    const bool id_test = IDS && doc_pair_needs_mask(a.doc_ranges, b, a.L, t, blk_id);   // block-uniform
    if (IDS && id_test && tid < BKV8) sid_s[(t & 1) * BKV8 + tid] = (kv0 + tid < a.L) ? a.sample_ids[rowbase + kv0 + tid] : -2;
    // this wave's pieces of tile t have landed once at most the pieces of the tiles issued after it are outstanding
    const int ahead = min(t_end - 1 - t, NST - 2);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // ... and everybody's; all waves are done with tile t-1, whose stage is the one tile t + NST - 1 goes to
    if (t + NST - 1 < t_end) issue_tile(t + NST - 1);
    const int vse = (int)Es[16];
    // S^T = K Q^T : [64 keys] x [32 queries per wave], 2 x KC instructions of 64-deep contraction
    f32x16_t sT[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[f][r] = 0.f;
    int kse[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) kse[f] = (int)((Es[(2 * f + (l31 >> 4)) * 32 + (l31 & 15)] >> ksh) & 0xffu);
#pragma unroll
    for (int cc = 0; cc < KC; ++cc)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int row = f * 32 + l31;
        const i32x8_t kf = lds_frag32(Ks, tile_off<D / 2>(row, cc * 4 + hi * 2), tile_off<D / 2>(row, cc * 4 + hi * 2 + 1));
        sT[f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[cc], sT[f], 0, 0, 0, kse[f], 0, qse);
      }
    if (id_test || kv0 + BKV8 > a.L) {
      long sid_q = 0;
      if (IDS && id_test) sid_q = q_ok ? a.sample_ids[rowbase + qi] : -1;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          bool ok = kv0 + kl < a.L;
          if (IDS) ok = ok && (!id_test || attn_pair_ok(sid_q, sidk[kl]));
          if (!ok) sT[f][r] = -INFINITY;
        }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sT[f][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // Lazy reference (as in the bf16 kernel, with the head-room e4m3 leaves): m is the reference exponent of this query and only moves when some query of
    // the wave saw a score more than 0.75 (log2 units) above its reference - then 2^8 p <= 2^8.75 = 431 < 448 still converts without saturation.  With the
    // exact running maximum the wave rescaled O^T (64 multiplies per lane) in most tiles: one of its 32 queries almost always sets a new record.
    // (explicit fused multiply-adds: the decision and the exponents must not depend on how each instantiation of this body happens to be contracted -
    // walks with and without the per-element id code are compared bit for bit)
    if (__builtin_amdgcn_ballot_w64(q_ok && (mloc * c > __builtin_fmaf(m, c, 0.75f))) != 0) {
      const float m_new = fmaxf(m, mloc);
      const float alpha = __builtin_amdgcn_exp2f((m - ((m_new == -INFINITY) ? 0.f : m_new)) * c);
      lsum *= alpha;
      m = m_new;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[i][r] *= alpha;
    }
    const float mc8 = (m == -INFINITY) ? -8.0f : __builtin_fmaf(m, c, -8.0f);
    float psum = 0.f;
    i32x8_t pb;
#pragma unroll
    for (int w = 0; w < 8; ++w) {   // dword w = registers 4 w .. 4 w + 3 of (sT[0] | sT[1]): byte f*16 + r
      float pv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pv[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[w >> 2][4 * (w & 3) + j], c, -mc8));
        psum += pv[j];
      }
      int u = 0;
      u = __builtin_amdgcn_cvt_pk_fp8_f32(pv[0], pv[1], u, false);
      u = __builtin_amdgcn_cvt_pk_fp8_f32(pv[2], pv[3], u, true);
      pb[w] = u;
    }
    lsum += psum;
    // O^T += V^T P^T over the 64 keys of the tile
#pragma unroll
    for (int i = 0; i < DB; ++i) {
      const int row = i * 32 + l31;
      const i32x8_t vf = lds_frag32(Vs, tile_off<32>(row, hi * 2), tile_off<32>(row, hi * 2 + 1));
      oT[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pb, oT[i], 0, 0, 0, vse, 0, P_E8);
    }
  };
#define UDM_WALK8(IDS_C)                                                                                     \
  {                                                                                                           \
    int t = t_begin;                                                                                          \
    for (; t < t_end && (t & (NST - 1)); ++t) tile(std::integral_constant<int, -1>{}, t, IDS_C);               \
    for (; t + NST - 1 < t_end; t += NST) {                                                                   \
      tile(std::integral_constant<int, 0>{}, t, IDS_C);                                                       \
      tile(std::integral_constant<int, 1>{}, t + 1, IDS_C);                                                   \
      tile(std::integral_constant<int, 2>{}, t + 2, IDS_C);                                                   \
      tile(std::integral_constant<int, 3>{}, t + 3, IDS_C);                                                   \
    }                                                                                                         \
    for (; t < t_end; ++t) tile(std::integral_constant<int, -1>{}, t, IDS_C);                                 \
  }
  if (HAS_SID && !doc_pure) UDM_WALK8(std::true_type{}) else UDM_WALK8(std::false_type{})
#undef UDM_WALK8
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);     // = 2^8 sum p
  const float inv = ltot > 0.f ? 256.0f / ltot : 0.f;
  if (q_ok && hi == 0) a.lse[((long)b * a.H + h) * a.L + qi] = ltot > 0.f ? __builtin_fmaf(m, c, log2f(ltot) - 8.0f) : INFINITY;
  if constexpr (D == 128) {
    if (a.out_stride % 8 == 0) {   // block-uniform: whole-row stores through the (now idle) ring
      __syncthreads();
      const int q0 = tile_x * BQ8 + wave * 32;
      store_rows_via_lds_d128(smem + wave * 8192, oT, inv, a.out + (rowbase + q0) * a.out_stride + h * D, a.out_stride, a.L - q0, lane);
      return;
    }
  }
  if (q_ok) {
    bf16_t* op = a.out + (rowbase + qi) * a.out_stride + h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = i * 32 + 8 * rg + 4 * hi;
        *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(oT[i][rg * 4] * inv, oT[i][rg * 4 + 1] * inv), pack2bf(oT[i][rg * 4 + 2] * inv, oT[i][rg * 4 + 3] * inv));
      }
  }
}

// ------------------------------------------------------------------------------------------------ quantisation
// q, k in place: qkr [M][2d] bf16 (normalised + rotated) -> qk8 bytes, one E8M0 scale per (row, head) and part, qkr <- the dequantised values.
// Generic form (any d, D in {64, 128}); at d = 2048 the qk-norm + rope kernel does this itself while the row is in registers (rowops.hip).
// A group of D / 8 lanes owns one (row, part, head): 8 values per lane.
template <int D>
__global__ __launch_bounds__(256) void fp8_quant_qk_kernel(bf16_t* __restrict__ qkr, uint8_t* __restrict__ qk8, uint8_t* __restrict__ qk_e8, long M, int H) {
  constexpr int LPG = D / 8;
  const long groups = M * 2 * H;
  const int d = H * D, Hp = (H + 3) & ~3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups * LPG; i += (long)gridDim.x * 256) {
    const long g = i / LPG;                  // (row, part * H + head)
    const int j = (int)(i % LPG);
    const long row = g / (2 * H);
    const int ph = (int)(g % (2 * H));
    const long off = row * 2L * d + (long)ph * D + j * 8;
    const uint4 u = *reinterpret_cast<const uint4*>(qkr + off);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    float v[8];
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
      amax = fmaxf(amax, fmaxf(fabsf(v[2 * e]), fabsf(v[2 * e + 1])));
    }
#pragma unroll
    for (int o = LPG / 2; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const int e8 = udm::e8m0_for_amax(amax);
    const uint2 q8 = udm::quant8_e4m3(v, e8);
    *reinterpret_cast<uint2*>(qk8 + off) = q8;
    *reinterpret_cast<uint4*>(qkr + off) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    if (j == 0) qk_e8[row * 2 * Hp + (ph / H) * Hp + ph % H] = (uint8_t)e8;
  }
}

// v -> v8t[bh][tile][D][64] + one E8M0 scale per (bh, tile).  Block = (bh, 64-key tile).  A thread takes 4 consecutive keys x 8 head dims (D = 128: all 256
// threads, one item each), the block agrees on the tile's amax, then the 4 x 8 bytes are transposed in registers into 8 dwords (4 keys of one dim) at byte
// position hi*32 + f*16 + 4*(k'>>3) of the dim's row: the order in which a lane of the attention kernel holds its probabilities
// (key = f*32 + (r&3) + 8 (r>>2) + 4 hi at byte f*16 + r of half hi).  Keys past L are zero.
template <int D>
__global__ __launch_bounds__(256) void fp8_quant_vt_kernel(const bf16_t* __restrict__ v, long vs, int H, int L, int ntile, uint8_t* __restrict__ v8t,
                                                          int* __restrict__ v_e8) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[D][64 + 16];
  __shared__ float red[4];
  const int bh = blockIdx.x / ntile, t = blockIdx.x % ntile;
  const int b = bh / H, h = bh % H;
  constexpr int VPR = D / 8;   // 16-byte vectors per key row
  const int i = threadIdx.x;
  const bool act = i < 16 * VPR;
  const int kg = i / VPR, c8 = (i % VPR) * 8;   // key group (4 keys), first head dim
  float x[4][8];
  float amax = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int l = t * 64 + kg * 4 + kk;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (act && l < L) u = *reinterpret_cast<const uint4*>(v + ((long)b * L + l) * vs + h * D + c8);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[kk][2 * e] = __uint_as_float(w[e] << 16); x[kk][2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
      amax = fmaxf(amax, fmaxf(fabsf(x[kk][2 * e]), fabsf(x[kk][2 * e + 1])));
    }
  }
  amax = udm::wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int e8 = udm::e8m0_for_amax(amax);
  if (threadIdx.x == 0) v_e8[(long)bh * ntile + t] = e8;
  if (act) {
    const float inv = udm::e8m0_inv_scale(e8);
    uint32_t w[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = x[kk][e] * inv;
      const uint2 r = udm::pack8_e4m3(s);
      w[kk][0] = r.x; w[kk][1] = r.y;
    }
    const int pos = (kg & 1) * 32 + (kg >> 3) * 16 + ((kg & 7) >> 1) * 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sh = 8 * (e & 3);
      const uint32_t o = ((w[0][e >> 2] >> sh) & 0xffu) | (((w[1][e >> 2] >> sh) & 0xffu) << 8) | (((w[2][e >> 2] >> sh) & 0xffu) << 16) | (((w[3][e >> 2] >> sh) & 0xffu) << 24);
      *reinterpret_cast<uint32_t*>(&tile[c8 + e][pos]) = o;
    }
  }
  __syncthreads();
  uint8_t* dst = v8t + ((long)bh * ntile + t) * (D * 64);
  for (int j = threadIdx.x; j < D * 4; j += 256) {
    const int dd = j / 4, w16 = (j % 4) * 16;
    *reinterpret_cast<uint4*>(dst + dd * 64 + w16) = *reinterpret_cast<const uint4*>(&tile[dd][w16]);
  }
}

}  // namespace

extern "C" int udm_attention_quantize_qk_fp8(void* qkr, void* qk8, uint8_t* qk_e8, int64_t M, int64_t d, int64_t D, hipStream_t stream) {
  UDM_CHECK_ARG(qkr && qk8 && qk_e8, "udm_attention_quantize_qk_fp8: null pointer");
  UDM_CHECK_ARG(M > 0 && (D == 64 || D == 128) && d % D == 0, "udm_attention_quantize_qk_fp8: bad shape (head_dim 64 or 128)");
  const int H = (int)(d / D);
  const long lanes = M * 2 * H * (D / 8);
  const int grid = (int)std::min<long>((lanes + 255) / 256, 4096);
  if (D == 128) hipLaunchKernelGGL(fp8_quant_qk_kernel<128>, dim3(grid), dim3(256), 0, stream, (bf16_t*)qkr, (uint8_t*)qk8, qk_e8, (long)M, H);
  else hipLaunchKernelGGL(fp8_quant_qk_kernel<64>, dim3(grid), dim3(256), 0, stream, (bf16_t*)qkr, (uint8_t*)qk8, qk_e8, (long)M, H);
  UDM_CHECK_LAUNCH("udm_attention_quantize_qk_fp8");
  return 0;
}

extern "C" int udm_attention_quantize_v_fp8(const void* v, int64_t v_stride, void* v8t, int32_t* v_e8, int64_t B, int64_t H, int64_t L, int64_t D, hipStream_t stream) {
  UDM_CHECK_ARG(v && v8t && v_e8, "udm_attention_quantize_v_fp8: null pointer");
  UDM_CHECK_ARG(B > 0 && H > 0 && L > 0 && (D == 64 || D == 128) && v_stride % 8 == 0, "udm_attention_quantize_v_fp8: bad shape (head_dim 64 or 128, row stride a multiple of 8)");
  const int ntile = (int)((L + 63) / 64);
  UDM_CHECK_ARG(B * H * ntile < (1LL << 31), "udm_attention_quantize_v_fp8: grid too large");
  const dim3 grid((unsigned)(B * H * ntile));
  if (D == 128) hipLaunchKernelGGL(fp8_quant_vt_kernel<128>, grid, dim3(256), 0, stream, (const bf16_t*)v, (long)v_stride, (int)H, (int)L, ntile, (uint8_t*)v8t, v_e8);
  else hipLaunchKernelGGL(fp8_quant_vt_kernel<64>, grid, dim3(256), 0, stream, (const bf16_t*)v, (long)v_stride, (int)H, (int)L, ntile, (uint8_t*)v8t, v_e8);
  UDM_CHECK_LAUNCH("udm_attention_quantize_v_fp8");
  return 0;
}

extern "C" int udm_attention_fwd_fp8(const void* qk8, const uint8_t* qk_e8, const void* v8t, const int32_t* v_e8, void* o, float* lse, const int64_t* sample_ids,
                                     const int32_t* doc_ranges, int64_t B, int64_t H, int64_t L, int64_t D, int64_t o_stride, hipStream_t stream) {
  UDM_CHECK_ARG(qk8 && qk_e8 && v8t && v_e8 && o && lse, "udm_attention_fwd_fp8: null pointer");
  UDM_CHECK_ARG(B > 0 && H > 0 && L > 0 && (D == 64 || D == 128), "udm_attention_fwd_fp8: bad shape (head_dim 64 or 128)");
  UDM_CHECK_ARG(o_stride % 4 == 0 && (sample_ids || !doc_ranges), "udm_attention_fwd_fp8: bad o_stride / doc_ranges without sample_ids");
  UDM_CHECK_ARG(B * H * ((L + 127) / 128) < (1LL << 31), "udm_attention_fwd_fp8: grid too large");
  UDM_CHECK_ARG(((uintptr_t)qk8 & 15) == 0 && ((uintptr_t)v8t & 15) == 0, "udm_attention_fwd_fp8: qk8 / v8t must be 16-byte aligned");
  Fp8Args a{};
  a.qk8 = (const uint8_t*)qk8; a.qk_e8 = qk_e8; a.v8t = (const uint8_t*)v8t; a.v_e8 = v_e8; a.out = (bf16_t*)o; a.lse = lse;
  a.sample_ids = sample_ids; a.doc_ranges = doc_ranges; a.out_stride = o_stride;
  a.B = (int)B; a.H = (int)H; a.L = (int)L;
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
