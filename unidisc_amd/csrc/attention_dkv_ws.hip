// Attention backward, dK/dV, head dim 128, no document mask or document-pure key blocks: WAVE-SPECIALISED kernel, two waves per SIMD with different roles.
// (reference: the backward of flash_attn_qkvpacked_func / SDPA, models/dit.py:826-829, :843)
//
// Why: both accumulators dK^T, dV^T (128 registers) plus the K/V operands (64) only fit ONE wave per SIMD when one wave does the
// whole step, and a cycle timeline (s_memtime stamps, UDM_DKV_TIMELINE=1 below) showed that kernel bound by that single wave's
// instruction issue -- not by the matrix pipe, LDS or memory.  Two waves per SIMD cannot both hold the accumulators, so the two
// waves that share a SIMD take different roles for the same 32 keys of the block:
//
//     score wave (waves 0..3), iteration j:   X(j+1)  s, dp <- Q K^T, dO V^T      16 MFMAs back to back, K/V fragments in registers
//                                             Y(j)    p = exp2(s c - lse), ds = p (dp - delta), packed to bf16 and handed over
//                                                     through an LDS exchange buffer (4 KiB per step, lane-linear ds_write_b128)
//     accum wave (waves 4..7), iteration j:   refill  LDS-DMA of step j+4 (Q, dO, lse, delta) -- ALL refills are issued here
//                                             Z(j-1)  dV^T += dO^T P, dK^T += Q^T dS    16 MFMAs, no VALU work at all
//
// (a workgroup's waves are placed on SIMDs cyclically, so wave w and wave w+4 share a SIMD).  The score wave's MFMA burst runs while
// its partner issues DMA; its softmax VALU stream runs under the partner's MFMAs.  The query axis is walked in 32-row steps through
// a 6-stage LDS ring (prefetch distance 3).  ONE workgroup barrier per step orders everything:
//   - ring: step j+1 has landed (accum waves wait on their DMA before the barrier); the stage refilled after barrier j held step
//     j-2, last read by Z(j-2) in iteration j-1;
//   - exchange (double-buffered on j & 1): written by Y(j) in iteration j, read by Z(j) in iteration j+1, rewritten by Y(j+2).
// Rows past L are clamped by the DMA and neutralised by writing lse = +inf for them (p = ds = 0); keys past L only pollute
// accumulator columns that are never stored.
//
// Measured (B8 H16 L1280 D128, MI355X): 0.27 ms vs 0.39 ms for the single-role kernel.  Per 32-query step the timeline reads
// ~2150 cycles: score wave = 16 MFMAs in ~1050 (the partner's MFMAs share the pipe) + ~700 of VALU; 32 MFMAs run at ~45 cycles each.
#include "attention_common.h"

#include <stdio.h>
#include <stdlib.h>

namespace {
namespace dkvw {
constexpr int NST = 6, PD = 3, SUB = 32, D = 128, KS = 8, DB = 4;
constexpr int TQ = SUB * D * 2, STG = 2 * TQ;     // one step: Q tile | dO tile
constexpr int LD_OFF = NST * STG;                 // lse/delta ring: [NST][lse 32 f32 | delta 32 f32]
constexpr int XC_OFF = LD_OFF + NST * 256;        // exchange: [2 parities][4 key groups][pb0 | pb1 | dsb0 | dsb1][64 lanes x 16 B]
constexpr int LDS_BYTES = XC_OFF + 2 * 4 * 4096;
constexpr int DMA_PER_SUB = 5;                    // per ACCUM wave (score waves issue none): two Q pieces, two dO pieces, lse|delta
constexpr int TL_FIRST = 12, TL_ITERS = 6, TL_TAGS = 8;   // timeline: iterations recorded, stamps per iteration

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

struct Ctx {
  const char* qbase; const char* dobase;          // byte pointers at this (batch, head)
  const float* ldp;                               // per-lane lse (lanes 0..31) / delta (lanes 32..63) pointer at query row (lane & 31)
  long qstep, ostep;                              // bytes per 32-row step
  const bf16_t* qb16; const bf16_t* ob16; long q_stride, do_stride;   // ragged-step path
  int L, nsub, wave, lane;
  float c;
  float lse_pad;                                  // what a query row past L gets as its (possibly negated) lse: +inf, or -inf in the pre-scaled form - p = 0 either way
  uint32_t offq[2], offo[2];                      // accum wave's DMA pieces: byte offset of (row, swizzled slot) inside a step
  uint32_t xoff[8];                               // LDS byte addresses (stage 0): row fragments, k-step m
  uint32_t zoff1[4], zoff2[4];                    // LDS byte addresses (stage 0): transposing reads, d block i, rows +0 / +8
  uint32_t ldoff, xcoff;                          // LDS byte addresses: lse/delta ring (+4 hi rows), this key group's exchange slot
  unsigned long long* tl;                         // TIMELINE: cycle stamps of block 0, [wave][iteration][tag]
};

template <bool TIMELINE>
__device__ __forceinline__ void stamp(const Ctx& x, int j, int tag) {
  if (TIMELINE) {
    if (x.tl && j >= TL_FIRST && j < TL_FIRST + TL_ITERS) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (x.lane == 0) x.tl[(x.wave * TL_ITERS + (j - TL_FIRST)) * TL_TAGS + tag] = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// (The refills are inline asm - dma16_asm / dma4_asm - because the compiler puts s_waitcnt vmcnt(0) in front of the transposing reads of the
// Z section when it knows of an LDS-DMA in flight: the ring's prefetch depth was being waited away every step.)
// Refill of one step, issued by the four accum waves (an LDS-DMA piece costs its issuer ~100 cycles; the accum stream has that slack
// while the score wave is in its MFMA burst).  Full steps: wave-uniform base + precomputed 32-bit lane offset, no vector address math.
__device__ __forceinline__ void issue_sub(const Ctx& x, char* smem, int sub, int stage) {
  const int kw = x.wave & 3;
  char* st = smem + stage * STG + kw * 2048;
  char* ldst = smem + LD_OFF + stage * 256;
  if ((sub + 1) * SUB <= x.L) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      dma16_asm(x.qbase + sub * x.qstep + (size_t)x.offq[jj], st + jj * 1024);
      dma16_asm(x.dobase + sub * x.ostep + (size_t)x.offo[jj], st + TQ + jj * 1024);
    }
    // lse (lanes 0..31) | delta (lanes 32..63); every accum wave issues the same copy so that all of them carry identical vmcnt bookkeeping
    dma4_asm(x.ldp + sub * SUB, ldst);
  } else {  // ragged last step: clamp rows to L-1 (finite data; the lse = +inf fix-up zeroes their probabilities)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int row = (kw * 2 + jj) * 4 + x.lane / 16;
      const int slot = (x.lane % 16) ^ swz<D>(row);
      const int grow = min(sub * SUB + row, x.L - 1);
      dma16_asm(x.qb16 + (long)grow * x.q_stride + slot * 8, st + jj * 1024);
      dma16_asm(x.ob16 + (long)grow * x.do_stride + slot * 8, st + TQ + jj * 1024);
    }
    const int qrow = min(sub * SUB + (x.lane & 31), x.L - 1);
    dma4_asm(x.ldp - (x.lane & 31) + qrow, ldst);
  }
}

// top of iteration j for every wave: step j+1 has landed, everybody is done with iteration j-1.  lgkmcnt(0): this wave's LDS reads and
// exchange writes of iteration j-1 are complete before anybody may overwrite / read them, wherever the compiler put the consumers.
template <bool TIMELINE, int ROLE>
__device__ __forceinline__ void top(const Ctx& x, char* smem, int j, int jm) {
  stamp<TIMELINE>(x, j, 0);
  __builtin_amdgcn_sched_barrier(0);
  if (ROLE == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // score waves have no DMA in flight
  else if (j + PD < x.nsub) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_SUB * (PD - 1)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  stamp<TIMELINE>(x, j, 1);
  if (ROLE == 1 && j + 1 + PD < x.nsub) issue_sub(x, smem, j + 1 + PD, (jm + 1 + PD) % NST);
  if (ROLE == 0 && x.wave == 0 && j + 1 < x.nsub && (j + 2) * SUB > x.L && x.lane >= x.L - (j + 1) * SUB && x.lane < 32)   // ragged last step (read at iteration j+1)
    *reinterpret_cast<float*>(smem + LD_OFF + ((jm + 1) % NST) * 256 + x.lane * 4) = x.lse_pad;
  stamp<TIMELINE>(x, j, 2);
}

// score wave, iteration j.  JM = j mod NST as a compile-time constant (ring stages and exchange parity become immediate offsets),
// or -1: derive them at run time (head / tail iterations).
// PRE (q pre-scaled by log2(e) / sqrt(D): UDM_ATTN_Q_PRESCALED, the engine's form): the ring holds -lse | -delta (negated copies the dQ kernel writes) and the
// score chains of X(j+1) START from them - the C operand of their first MFMA - so that the accumulators are s - lse and dp - delta as they leave the matrix
// pipe: Y(j) is exp2 + multiply + pack per score instead of fma + subtract + exp2 + multiply + pack (the score wave is bound by its instruction issue).
template <bool TIMELINE, int JM, bool HX, bool HY, bool PRE>
__device__ __forceinline__ void score_step(const Ctx& x, char* smem, int j, const bf16x8_t (&kf)[KS], const bf16x8_t (&vf)[KS], const f32x16_t& s_in,
                                           const f32x16_t& dp_in, f32x16_t& s_out, f32x16_t& dp_out) {
  const int jm = (JM >= 0) ? JM : ((j + NST) % NST);
  top<TIMELINE, 0>(x, smem, j, jm);
  const uint32_t qx = ((jm + 1) % NST) * STG, ldy = x.ldoff + jm * 256;
  f32x4_t l4[4], d4[4];
  if (HY && !PRE) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      l4[rg] = lds_ld<f32x4_t>(ldy + 32 * rg);
      d4[rg] = lds_ld<f32x4_t>(ldy + 128 + 32 * rg);
    }
  }
  if (HX) {
    // X(j+1): all 16 MFMAs back to back, row fragments read four k-steps ahead (issue order pinned with sched_barrier: left to itself
    // the compiler reads each fragment right before its MFMA and eats the LDS latency every time)
    bf16x8_t xq[8], xo[8];
    auto x_read = [&](int m) { xq[m] = lds_ld<bf16x8_t>(x.xoff[m] + qx); xo[m] = lds_ld<bf16x8_t>(x.xoff[m] + qx + TQ); };
    f32x16_t s, dp;
    if (PRE) {   // -lse | -delta of step j+1 (landed with its Q / dO rows)
      const uint32_t ldx = x.ldoff + ((jm + 1) % NST) * 256;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4_t nl = lds_ld<f32x4_t>(ldx + 32 * rg), nd = lds_ld<f32x4_t>(ldx + 128 + 32 * rg);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[4 * rg + e] = nl[e]; dp[4 * rg + e] = nd[e]; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) x_read(m);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m + 4 < 8) x_read(m + 4);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xq[m], kf[m], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xo[m], vf[m], dp, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    s_out = s; dp_out = dp;
  }
  stamp<TIMELINE>(x, j, 3);
  if (HY) {
    // Y(j) on the s/dp of the previous iteration, batched by operation (element-serial order is a chain of dependent VALU latencies)
    float t[16], u[16], p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = PRE ? s_in[r] : s_in[r] * x.c - l4[r >> 2][r & 3];
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = PRE ? dp_in[r] : dp_in[r] - d4[r >> 2][r & 3];
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(t[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = p[r] * u[r];
    const uint32_t xc = x.xcoff + ((JM >= 0) ? (JM & 1) : (j & 1)) * (4 * 4096);
    UDM_LDS bf16x8_t* dst = reinterpret_cast<UDM_LDS bf16x8_t*>((size_t)xc);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {   // accumulator registers 8 c2 .. 8 c2 + 7 are the 8 k-slots of the second MFMA's B operand (attention.hip header)
      const float* pp = &p[8 * c2];
      const float* dd = &ds[8 * c2];
      const u32x4_t pv = {pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3]), pack2bf(pp[4], pp[5]), pack2bf(pp[6], pp[7])};
      const u32x4_t dv = {pack2bf(dd[0], dd[1]), pack2bf(dd[2], dd[3]), pack2bf(dd[4], dd[5]), pack2bf(dd[6], dd[7])};
      dst[c2 * 64] = __builtin_bit_cast(bf16x8_t, pv);
      dst[128 + c2 * 64] = __builtin_bit_cast(bf16x8_t, dv);
    }
  }
  stamp<TIMELINE>(x, j, 4);
}

// accum wave, iteration j:  refill (in top), then Z(j-1): operands of step j-1 from the exchange buffer, Q^T / dO^T by transposing reads
template <bool TIMELINE, int JM, bool HZ>
__device__ __forceinline__ void accum_step(const Ctx& x, char* smem, int j, f32x16_t (&dkT)[DB], f32x16_t (&dvT)[DB]) {
  const int jm = (JM >= 0) ? JM : ((j + NST) % NST);
  top<TIMELINE, 1>(x, smem, j, jm);
  if (!HZ) return;
  const uint32_t qz = ((jm + NST - 1) % NST) * STG;
  const uint32_t xc = x.xcoff + ((JM >= 0) ? (((JM + NST - 1) % NST) & 1) : ((j - 1) & 1)) * (4 * 4096);
  bf16x8_t pk[4], zq[8], zo[8];
  auto z_read = [&](int m) {
    const uint32_t i = m & 3, o = qz + (m >> 2) * (16 * 2 * D);
    s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)(x.zoff1[i] + o + TQ));
    s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)(x.zoff2[i] + o + TQ));
    s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)(x.zoff1[i] + o));
    s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)(x.zoff2[i] + o));
    zo[m] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
    zq[m] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
  };
#pragma unroll
  for (int f = 0; f < 4; ++f) pk[f] = lds_ld<bf16x8_t>(xc + f * 1024);
#pragma unroll
  for (int m = 0; m < 4; ++m) z_read(m);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < 8; ++m) {   // m: d block i = m & 3, 16-query chunk c2 = m >> 2
    if (m + 4 < 8) z_read(m + 4);
    dvT[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zo[m], pk[m >> 2], dvT[m & 3], 0, 0, 0);
    dkT[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zq[m], pk[2 + (m >> 2)], dkT[m & 3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  stamp<TIMELINE>(x, j, 3);
}
}  // namespace dkvw

template <bool TIMELINE, bool PRE = false>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_ws_kernel(AttnArgs a) {
  using namespace dkvw;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, kw = wave & 3;   // role 0: score wave, 1: accum wave; both own keys kw*32 .. kw*32+31 of the block
  // 1-D grid, tile-major (id = tile * (B*H) + bh): all key blocks of one (b, h) land on the same XCD and share its Q / dO through that L2
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);
  const int b = bh / a.H, h = bh % a.H;
  const int ki = tile_x * 128 + kw * 32 + l31;
  const bool k_ok = ki < a.L;
  const long rowbase = (long)b * a.L;
  const long sbase = ((long)b * a.H + h) * a.L;
  // Packed documents (a.doc_ranges, launched by attention.hip::launch_bwd beside the single-role kernel): this kernel takes the key blocks that lie
  // inside ONE document whose rows are exactly the positions [lo, hi) - then no (query, key) pair of the walk needs an id test, and the walk is this
  // kernel's ordinary one over the `hi - lo` queries starting at `lo`.  Every other key block returns here and is computed by the other kernel.
  int q_lo = 0, q_n = a.L;
  if (a.doc_ranges != nullptr) {
    const DocSpan sp = doc_tile_span(a.doc_ranges, b, a.L, tile_x, (a.L + 63) / 64);
    if (!sp.pure) return;   // block-uniform
    q_lo = sp.lo;
    q_n = sp.hi - sp.lo;
  }

  Ctx x;
  x.qb16 = a.q + (rowbase + q_lo) * a.q_stride + h * D; x.ob16 = a.dout + (rowbase + q_lo) * a.do_stride + h * D;
  x.qbase = reinterpret_cast<const char*>(x.qb16); x.dobase = reinterpret_cast<const char*>(x.ob16);
  x.q_stride = a.q_stride; x.do_stride = a.do_stride;
  x.qstep = (long)SUB * a.q_stride * 2; x.ostep = (long)SUB * a.do_stride * 2;
  // PRE: the negated copies sit behind delta in the caller's scratch: delta | -lse | -delta, B H L floats each (written by the dQ kernel)
  const long plane = (long)a.B * a.H * a.L;
  x.ldp = (PRE ? (lane < 32 ? a.delta + plane : a.delta + 2 * plane) : (lane < 32 ? a.lse : a.delta)) + sbase + q_lo + l31;
  x.lse_pad = PRE ? -INFINITY : INFINITY;
  x.L = q_n; x.nsub = (q_n + SUB - 1) / SUB; x.wave = wave; x.lane = lane; x.c = a.scale_log2;
  {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int row = (kw * 2 + jj) * 4 + lane / 16, slot = (lane % 16) ^ swz<D>(row);
      x.offq[jj] = (uint32_t)((row * a.q_stride + slot * 8) * 2);
      x.offo[jj] = (uint32_t)((row * a.do_stride + slot * 8) * 2);
    }
    const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;   // folded into every LDS address once, here
    const int g1 = (lane >> 4) & 1, p = lane & 15;
#pragma unroll
    for (int m = 0; m < 8; ++m) x.xoff[m] = lds0 + tile_off<D>(l31, m * 2 + hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // same gather as lds_frag_T<D, true> (attention_common.h)
      const int r = 4 * hi + (p >> 2), col = i * 32 + g1 * 16 + (p & 3) * 4;
      x.zoff1[i] = lds0 + tile_off<D>(r, col >> 3) + (col & 7) * 2;
      x.zoff2[i] = lds0 + tile_off<D>(r + 8, col >> 3) + (col & 7) * 2;
    }
    x.ldoff = lds0 + LD_OFF + 16 * hi;
    x.xcoff = lds0 + XC_OFF + kw * 4096 + lane * 16;
    x.tl = TIMELINE && blockIdx.x == 0 ? reinterpret_cast<unsigned long long*>(const_cast<bf16_t*>(a.o)) : nullptr;
  }
  const int nsub = x.nsub;

  // Both roles run iterations j = -1 .. nsub (one barrier each).  Steady iterations are unrolled over the ring period.
  if (role == 0) {
    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = load_frag_global(a.k + (rowbase + ki) * a.k_stride + h * D + ks * 16 + hi * 8, k_ok);
      vf[ks] = load_frag_global(a.v + (rowbase + ki) * a.v_stride + h * D + ks * 16 + hi * 8, k_ok);
    }
    f32x16_t sA, dpA, sB, dpB;   // ping-pong: odd iterations read A / write B, even iterations read B / write A
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[r] = 0.f; dpA[r] = 0.f; sB[r] = 0.f; dpB[r] = 0.f; }
    score_step<TIMELINE, -1, true, false, PRE>(x, smem, -1, kf, vf, sA, dpA, sB, dpB);   // j = -1: X(0)
    int j = 0;
#define UDM_WS_EVEN(JM) score_step<TIMELINE, JM, true, true, PRE>(x, smem, j, kf, vf, sB, dpB, sA, dpA)
#define UDM_WS_ODD(JM) score_step<TIMELINE, JM, true, true, PRE>(x, smem, j, kf, vf, sA, dpA, sB, dpB)
    for (; j + 5 <= nsub - 2;) {
      UDM_WS_EVEN(0); ++j; UDM_WS_ODD(1); ++j; UDM_WS_EVEN(2); ++j; UDM_WS_ODD(3); ++j; UDM_WS_EVEN(4); ++j; UDM_WS_ODD(5); ++j;
    }
    if (j <= nsub - 2) { UDM_WS_EVEN(0); ++j; }
    if (j <= nsub - 2) { UDM_WS_ODD(1); ++j; }
    if (j <= nsub - 2) { UDM_WS_EVEN(2); ++j; }
    if (j <= nsub - 2) { UDM_WS_ODD(3); ++j; }
    if (j <= nsub - 2) { UDM_WS_EVEN(4); ++j; }
#undef UDM_WS_EVEN
#undef UDM_WS_ODD
    // j = nsub-1: Y only;  j = nsub: barrier only
    if (j & 1) score_step<TIMELINE, -1, false, true, PRE>(x, smem, j, kf, vf, sA, dpA, sB, dpB);
    else score_step<TIMELINE, -1, false, true, PRE>(x, smem, j, kf, vf, sB, dpB, sA, dpA);
    score_step<TIMELINE, -1, false, false, PRE>(x, smem, j + 1, kf, vf, sA, dpA, sB, dpB);
  } else {
    f32x16_t dkT[DB], dvT[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkT[i][r] = 0.f; dvT[i][r] = 0.f; }
#pragma unroll
    for (int s0 = 0; s0 < PD; ++s0)
      if (s0 < nsub) issue_sub(x, smem, s0, s0);
    accum_step<TIMELINE, -1, false>(x, smem, -1, dkT, dvT);
    accum_step<TIMELINE, 0, false>(x, smem, 0, dkT, dvT);
    int j = 1;
#define UDM_WS_Z(JM) accum_step<TIMELINE, JM, true>(x, smem, j, dkT, dvT)
    for (; j + 5 <= nsub;) {
      UDM_WS_Z(1); ++j; UDM_WS_Z(2); ++j; UDM_WS_Z(3); ++j; UDM_WS_Z(4); ++j; UDM_WS_Z(5); ++j; UDM_WS_Z(0); ++j;
    }
    if (j <= nsub) { UDM_WS_Z(1); ++j; }
    if (j <= nsub) { UDM_WS_Z(2); ++j; }
    if (j <= nsub) { UDM_WS_Z(3); ++j; }
    if (j <= nsub) { UDM_WS_Z(4); ++j; }
    if (j <= nsub) { UDM_WS_Z(5); ++j; }
#undef UDM_WS_Z
    if (a.out2_stride % 8 == 0 && a.out3_stride % 8 == 0) {
      // whole-row stores through this key group's two exchange slots (free: the last P / dS hand-over has been consumed by this very wave)
      const int k0 = ki - l31;   // first key of this wave's 32
      char* stg = smem + XC_OFF + (wave & 3) * 4096;
      store_rows_via_lds_d128(stg, dkT, a.scale, a.out2 + (rowbase + k0) * a.out2_stride + h * D, a.out2_stride, a.L - k0, lane, 4 * 4096);
      store_rows_via_lds_d128(stg, dvT, 1.0f, a.out3 + (rowbase + k0) * a.out3_stride + h * D, a.out3_stride, a.L - k0, lane, 4 * 4096);
    } else if (k_ok) {
      bf16_t* kp = a.out2 + (rowbase + ki) * a.out2_stride + h * D;
      bf16_t* vp = a.out3 + (rowbase + ki) * a.out3_stride + h * D;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int d0 = i * 32 + 8 * rg + 4 * hi;
          *reinterpret_cast<uint2*>(kp + d0) = make_uint2(pack2bf(dkT[i][rg * 4] * a.scale, dkT[i][rg * 4 + 1] * a.scale),
                                                          pack2bf(dkT[i][rg * 4 + 2] * a.scale, dkT[i][rg * 4 + 3] * a.scale));
          *reinterpret_cast<uint2*>(vp + d0) = make_uint2(pack2bf(dvT[i][rg * 4], dvT[i][rg * 4 + 1]), pack2bf(dvT[i][rg * 4 + 2], dvT[i][rg * 4 + 3]));
        }
    }
  }
}
}  // namespace

void udm_launch_attn_bwd_dkv_ws(const void* args, hipStream_t stream) {
  using namespace dkvw;
  const AttnArgs& a = *static_cast<const AttnArgs*>(args);
  const dim3 grid(((a.L + 127) / 128) * a.H * a.B);
  static bool once = false;
  static int timeline = 0;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_ws_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_ws_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (const char* e = getenv("UDM_DKV_TIMELINE")) timeline = atoi(e);
    once = true;
  }
  if (timeline > 0 && --timeline == 0) {
    // developer hook: the N-th call (UDM_DKV_TIMELINE=N) runs the instrumented build and prints block 0's cycle stamps to stderr:
    // per wave and iteration  [0] enter  [1] barrier released  [2] refill issued  [3] MFMA burst done  [4] (score wave) hand-off done
    constexpr int N = 8 * TL_ITERS * TL_TAGS;
    unsigned long long* buf = nullptr;
    if (hipMalloc(&buf, N * sizeof(unsigned long long)) == hipSuccess) {
      (void)hipMemsetAsync(buf, 0, N * sizeof(unsigned long long), stream);
      AttnArgs a2 = a;
      a2.o = reinterpret_cast<const bf16_t*>(buf);   // `o` is not read by this kernel
      hipLaunchKernelGGL(attn_bwd_dkv_ws_kernel<true>, grid, dim3(512), LDS_BYTES, stream, a2);
      (void)hipStreamSynchronize(stream);
      static unsigned long long h[N];
      (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w)
        for (int it = 0; it < TL_ITERS; ++it) {
          fprintf(stderr, "TL wave %d j %d:", w, it + TL_FIRST);
          for (int t = 0; t < 5; ++t) {
            const unsigned long long v = h[(w * TL_ITERS + it) * TL_TAGS + t];
            fprintf(stderr, " %lld", v ? (long long)(v - h[0]) : -1LL);
          }
          fprintf(stderr, "\n");
        }
      (void)hipFree(buf);
      return;
    }
  }
  static const int pre_on = [] { const char* e = getenv("UDM_DKV_PRE"); return e ? atoi(e) : 1; }();   // A/B switch: 0 = the plain arithmetic also for pre-scaled q
  if (a.q_prescaled && pre_on) {   // the engine's form: score chains start from -lse / -delta (see score_step)
    static bool once_p = false;
    if (!once_p) { (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_ws_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); once_p = true; }
    hipLaunchKernelGGL((attn_bwd_dkv_ws_kernel<false, true>), grid, dim3(512), LDS_BYTES, stream, a);
    return;
  }
  hipLaunchKernelGGL(attn_bwd_dkv_ws_kernel<false>, grid, dim3(512), LDS_BYTES, stream, a);
}

// Note on the dQ half: the same role split was built and measured for dQ (score wave: S, dP, softmax; accum wave: refills + dQ MFMAs).
// It is correct but no faster than the single-role kernel (185 vs 188 us): its timeline shows the score wave as the critical path
// (2 x 16 MFMAs at ~47 cycles each + ~1000 cycles of VALU, 32 v_exp_f32 alone are 512) while the accum wave idles ~900 cycles per
// tile, and moving dP to the accum wave would need the probabilities exchanged in fp32.  The 8 LDS-DMA pieces of a tile cost their
// issuer ~1000 cycles (126 each) in that timeline.
