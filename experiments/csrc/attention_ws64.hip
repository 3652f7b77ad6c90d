// EXPERIMENT (libunidisc_exp.so, scripts/ only - measured slower than the kernels the product uses): attention forward, head dim 128,
// no document mask, L a multiple of 128: WAVE-SPECIALISED kernel, 64 queries per wave pair.
// (reference: flash_attn_qkvpacked_func / SDPA, models/dit.py:826-829, :843)
//
// Why: a cycle timeline of the one-wave-per-SIMD kernel (attention_w64.hip) shows it bound by ONE wave's instruction issue - about 540
// instructions per 64-key tile at ~5.4 cycles each against 64 MFMAs x 32 cycles - and two such waves do not fit a SIMD's registers.  So the two
// waves of a SIMD take roles for the same 64 queries (as attention_dkv_ws.hip does for dK / dV):
//
//     score wave (waves 0..3)   S(t+1) = K(t+1) Q^T (32 MFMAs) interleaved with the softmax of tile t: exp2 / row sums / bf16 packing, P(t)
//                               handed over through an LDS exchange buffer (lane-linear 16-byte writes: the accumulator registers of a lane are
//                               the B-operand fragment of the same lane); then the running maximum of S(t+1) and the lazy-rescale decision
//     PV wave    (waves 4..7)   all LDS-DMA refills; O^T += V(t-1)^T P(t-1)^T (32 MFMAs) one tile behind, rescaling O^T when the score wave flagged
//                               tile t-1; at the end normalises by the row sums it gets from the score wave and stores O through LDS as whole rows
//
// (wave w and wave w + 4 share a SIMD).  Each stream is short enough (about 330 and 110 instructions per tile) that the matrix pipe, not
// instruction issue, paces the SIMD.  One workgroup barrier per tile orders everything:
//   - K ring (3 stages): K(t+3) is issued in iteration t into the stage K(t) left (last read in iteration t-1) and must have landed at barrier t+2;
//     V ring (2 stages): V(t) is issued in iteration t (first, so it is the OLDER group) into the stage V(t-2) left and must have landed at
//     barrier t+1: the PV waves wait with vmcnt(4) - everything but the four newest pieces, which are K(t+3)'s;
//   - exchange (double-buffered on t & 1): P(t) written in iteration t, read in iteration t+1, rewritten in iteration t+2;
//   - rescale record of tile t (flag + per-lane factors): written by the score wave at the START of iteration t, read in iteration t+1.
// The arithmetic is that of attn_fwd_kernel step for step (same k order, same rescale decisions per 32-query block, same summation order): O is
// bit-identical to the other two forward kernels.
#include "attention_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {
namespace ws64 {
constexpr int D = 128, KS = 8, DB = 4, BKV = 64, BQW = 256;
constexpr int TB = BKV * D * 2;          // one K or V tile, 16 KiB
constexpr int NKST = 3, NVST = 2;
constexpr int V_OFF = NKST * TB;
constexpr int X_OFF = (NKST + NVST) * TB;                 // exchange: [4 pairs][2 parities][8 fragments][64 lanes x 16 B]
constexpr int AL_OFF = X_OFF + 4 * 2 * 8192;              // rescale factors: [2 parities][4 pairs][2 blocks][64 lanes] f32
constexpr int FL_OFF = AL_OFF + 2 * 4 * 2 * 256;          // rescale flags:   [2 parities][4 pairs] u32 (bit q = block q rescales)
constexpr int LF_OFF = FL_OFF + 64;                       // 1 / row sum:     [4 pairs][2 blocks][64 lanes] f32
constexpr int LDS_BYTES = LF_OFF + 4 * 2 * 256;
constexpr int AHEAD = 3, NFR = 4;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// gap marker: nothing is scheduled across it, memory operations keep their side of it
__device__ __forceinline__ void sb() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma_piece(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// The score wave's VALU work is inline asm (volatile statements keep their order; pure arithmetic does not), one statement per MFMA gap:
// the MFMA (VGPR form) and one softmax pair - two scores -> p = exp2(s c - mc), row-sum update, one packed bf16 pair.  v_exp_f32 results are
// read one instruction later at the earliest (the transcendental-use wait state).
#define UDM_PAIR_ASM                                                                                                                      \
  "v_fma_f32 %0, %6, %8, -%9\n\tv_fma_f32 %1, %7, %8, -%9\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %2, %2, %1\n\t" \
  "v_cvt_pk_bf16_f32 %3, %0, %1"
template <bool ZERO>
__device__ __forceinline__ void mfma_pair(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b, float s0, float s1, float c, float mc, float& psum, uint32_t& pw) {
  float t0, t1;
  if (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %4, %5, %10, 0\n\t" UDM_PAIR_ASM
                 : "=&v"(t0), "=&v"(t1), "+v"(psum), "=&v"(pw), "=&v"(d) : "v"(a), "v"(s0), "v"(s1), "s"(c), "v"(mc), "v"(b));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %4, %5, %10, %4\n\t" UDM_PAIR_ASM
                 : "=&v"(t0), "=&v"(t1), "+v"(psum), "=&v"(pw), "+v"(d) : "v"(a), "v"(s0), "v"(s1), "s"(c), "v"(mc), "v"(b));
}
__device__ __forceinline__ void pair_only(float s0, float s1, float c, float mc, float& psum, uint32_t& pw) {
  float t0, t1;
  f32x16_t d_unused;
  bf16x8_t a_unused, b_unused;
  (void)d_unused; (void)a_unused; (void)b_unused;
  asm volatile(
      "v_fma_f32 %0, %4, %6, -%7\n\tv_fma_f32 %1, %5, %6, -%7\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %0\n\tv_add_f32 %2, %2, %1\n\t"
      "v_cvt_pk_bf16_f32 %3, %0, %1"
      : "=&v"(t0), "=&v"(t1), "+v"(psum), "=&v"(pw) : "v"(s0), "v"(s1), "s"(c), "v"(mc));
}
template <bool ZERO>
__device__ __forceinline__ void mfma_v(f32x16_t& d, const bf16x8_t& a, const bf16x8_t& b) {
  if (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void max2(float& mxa, float a0, float a1, float& mxb, float b0, float b1) {
  asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %4, %5" : "+v"(mxa), "+v"(mxb) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}

template <bool TL>
__global__ __launch_bounds__(512, 2) void attn_fwd_ws64_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3;
  // tile-major 1-D grid as in attention.hip: all query blocks of one (b, h) run on one XCD (block id % 8) and share K / V through its L2
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);
  const int b = bh / a.H, h = bh % a.H;
  const long rowbase = (long)b * a.L;
  const int L = a.L;
  const int nkv = L / BKV;                      // even, >= 2 (dispatch condition)
  const int q0w = tile_x * BQW + pair * 64;     // first query of this wave pair
  const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;
  const uint32_t xbase = lds0 + X_OFF + pair * 16384 + lane * 16;          // + parity * 8192 + fragment * 1024
  const uint32_t albase = lds0 + AL_OFF + pair * 512 + lane * 4;           // + parity * 2048 + block * 256
  const uint32_t flbase = lds0 + FL_OFF + pair * 4;                        // + parity * 16
  const uint32_t lfbase = lds0 + LF_OFF + pair * 512 + lane * 4;           // + block * 256

  // TL: cycle stamps (s_memtime) of block 0, [wave 0..7][64 tags], written by lane 0
  auto stamp = [&](int tag) {
    if (TL) {
      if (blockIdx.x == 0 && a.timeline && tag < 64) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tm = __builtin_amdgcn_s_memtime();
        if (lane == 0) a.timeline[wave * 64 + tag] = tm;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  stamp(0);
  if (wave < 4) {
    // =================================================================================================== score wave
    const float c = a.scale_log2;
    bf16x8_t qf[2][KS];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int qi = q0w + q * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[q][ks] = load_frag_global(a.q + (rowbase + qi) * a.q_stride + h * D + ks * 16 + hi * 8, qi < L);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[q][ks]));   // the wait for these loads sits here
    // Reference exponent per query.  Query rows past L start at +inf: they never ask for a move (their own columns turn into NaN when a
    // neighbour does - lane-local and never stored), so the loop needs no validity mask.
    float m[2], mc[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f}, alpha[2] = {1.f, 1.f}, psum[2];
    unsigned flag = 0;   // rescale flags of the tile whose record is written next
#pragma unroll
    for (int q = 0; q < 2; ++q) m[q] = (q0w + q * 32 + l31 < L) ? -INFINITY : INFINITY;
    const uint32_t kl = (uint32_t)tile_off<D>(l31, hi);   // K fragment (ks, f) at kl ^ (ks << 5), + f * 8192 (attention_w64.hip)
    auto kread = [&](uint32_t kb, int j) { return lds_ld<bf16x8_t>((kb ^ (uint32_t)((j >> 1) << 5)) + (j & 1) * 8192); };
    // running maximum of S, lazy-rescale decision (attention.hip: the reference exponent only moves when some query of the 32-query block saw
    // a score more than 2^8 above it) and its effect on this wave's state; the factors go to the PV wave with the next record
    auto maxima_and_move = [&](const f32x16_t (&S)[2][2]) {
      float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
      for (int k = 0; k < 16; ++k) max2(mx[0], S[0][k >> 3][(k & 7) * 2], S[0][k >> 3][(k & 7) * 2 + 1], mx[1], S[1][k >> 3][(k & 7) * 2], S[1][k >> 3][(k & 7) * 2 + 1]);
      flag = 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float mloc = fmaxf(mx[q], __shfl_xor(mx[q], 32, 64));
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mloc * c > m[q] * c + 8.0f);
        alpha[q] = 1.f;
        if (bal != 0) {
          const float m_new = fmaxf(m[q], mloc);
          alpha[q] = __builtin_amdgcn_exp2f((m[q] - ((m_new == -INFINITY) ? 0.f : m_new)) * c);
          lsum[q] *= alpha[q];
          m[q] = m_new;
          flag |= 1u << q;
        }
        mc[q] = (m[q] == -INFINITY) ? 0.f : m[q] * c;
      }
    };

    f32x16_t S0[2][2], S1[2][2];
    bf16x8_t kfr[NFR];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // K(0), K(1), K(2) have landed (PV waves waited for them)
    {
      const uint32_t kb = lds0 + (0 % NKST) * TB + kl;
      static_for<0, 16>([&](auto j_) {
        constexpr int j = decltype(j_)::value, ks = j >> 1, f = j & 1;
        const bf16x8_t kf = kread(kb, j);
        mfma_v<ks == 0>(S0[0][f], kf, qf[0][ks]);
        mfma_v<ks == 0>(S0[1][f], kf, qf[1][ks]);
      });
      asm volatile("s_nop 15" ::: "memory");   // asm MFMA results -> first VALU reader: wait states the compiler does not know it owes
      maxima_and_move(S0);
      flag = 0;   // nothing to rescale yet: O^T and the row sums are zero
    }

    // one tile: S(t+1) -> Sn (when HAS_NEXT) under the softmax of tile t (scores in Sc), P(t) to the exchange buffer
    auto body = [&](auto has_next_t, f32x16_t (&Sc)[2][2], f32x16_t (&Sn)[2][2], int t) {
      constexpr bool HAS_NEXT = decltype(has_next_t)::value;
      sb();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // K(t+1) is there; the exchange / record slots of parity t are free
      sb();
      stamp(4 + 2 * t);
      const int par = t & 1;
      // rescale record of tile t
      if (flag) {
        if (flag & 1) asm volatile("ds_write_b32 %0, %1" ::"v"(albase + par * 2048), "v"(alpha[0]) : "memory");
        if (flag & 2) asm volatile("ds_write_b32 %0, %1 offset:256" ::"v"(albase + par * 2048), "v"(alpha[1]) : "memory");
      }
      if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(flbase + par * 16), "v"(flag) : "memory");
      uint32_t kb = lds0 + ((t + 1) % NKST) * TB + kl;
      uint32_t xw = xbase + par * 8192;
      asm volatile("" : "+v"(kb), "+v"(xw));
      psum[0] = 0.f;
      psum[1] = 0.f;
      uint32_t Pw[4];
      if (HAS_NEXT) {
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) kfr[j] = kread(kb, j);
      }
      sb();
      static_for<0, 32>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value, j = mi >> 1, ks = mi >> 2, f = (mi >> 1) & 1, q = mi & 1;
        constexpr int uq = mi >> 4, ucc = (mi >> 2) & 3, ue = mi & 3, uf = ucc >> 1, ur = 8 * (ucc & 1) + 2 * ue;   // softmax unit mi of tile t
        if constexpr (HAS_NEXT) mfma_pair<ks == 0>(Sn[q][f], kfr[j % NFR], qf[q][ks], Sc[uq][uf][ur], Sc[uq][uf][ur + 1], c, mc[uq], psum[uq], Pw[ue]);
        else pair_only(Sc[uq][uf][ur], Sc[uq][uf][ur + 1], c, mc[uq], psum[uq], Pw[ue]);
        sb();
        if constexpr (HAS_NEXT && q == 1 && j + AHEAD < 16) kfr[(j + AHEAD) % NFR] = kread(kb, j + AHEAD);
        if constexpr (ue == 3) {   // fragment (block uq, 16-key chunk ucc) of P^T is complete
          asm volatile("ds_write_b128 %0, %1 offset:%c2" ::"v"(xw), "v"(u32x4_t{Pw[0], Pw[1], Pw[2], Pw[3]}), "i"((uq * 4 + ucc) * 1024) : "memory");
        }
        sb();
      });
      lsum[0] += psum[0];
      lsum[1] += psum[1];
      stamp(5 + 2 * t);
      if (HAS_NEXT) {
        asm volatile("s_nop 15" ::: "memory");
        maxima_and_move(Sn);
      }
      sb();
    };
    {
      const std::true_type T{};
      const std::false_type F{};
      int t = 0;
      for (; t + 2 < nkv; t += 2) {
        body(T, S0, S1, t);
        body(T, S1, S0, t + 1);
      }
      body(T, S0, S1, t);
      body(F, S1, S0, t + 1);
    }
    // row sums -> the PV wave; log-sum-exp -> global
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float ltot = lsum[q] + __shfl_xor(lsum[q], 32, 64);
      const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
      asm volatile("ds_write_b32 %0, %1 offset:%c2" ::"v"(lfbase), "v"(inv), "i"(q * 256) : "memory");
      const int qi = q0w + q * 32 + l31;
      if (qi < L && hi == 0) a.lse[((long)b * a.H + h) * L + qi] = ltot > 0.f ? __builtin_fmaf(m[q], c, log2f(ltot)) : INFINITY;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier nkv
  } else {
    // =================================================================================================== PV wave
    const long ks_ = a.k_stride, vs_ = a.v_stride;
    const bf16_t* kbase = a.k + rowbase * ks_ + h * D;
    const bf16_t* vbase = a.v + rowbase * vs_ + h * D;
    const long ktile_step = (long)BKV * ks_, vtile_step = (long)BKV * vs_;
    DmaPlan<D, BKV> plank, planv;
    plank.init(ks_, pair, lane);
    planv.init(vs_, pair, lane);
    auto refill = [&](const DmaPlan<D, BKV>& plan, const bf16_t* src, uint32_t dst) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dma_piece(plan.off[j], src, dst + j * 1024);
    };
    refill(plank, kbase, lds0 + 0 * TB + pair * 4096);
    refill(plank, kbase + ktile_step, lds0 + 1 * TB + pair * 4096);
    refill(plank, kbase + min(2, nkv - 1) * ktile_step, lds0 + 2 * TB + pair * 4096);
    f32x16_t oT[2][DB];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[q][i][r] = 0.f;
    // V^T fragment (cc, i), two transposing reads: v1 ^ (i << 6) + cc * 4096 and v2 ^ (i << 6) + cc * 4096 (attention_w64.hip)
    uint32_t v1, v2;
    {
      const int g1 = (lane >> 4) & 1, p = lane & 15;
      const int row = 4 * hi + (p >> 2), col = g1 * 16 + (p & 3) * 4;
      v1 = (uint32_t)(tile_off<D>(row, col >> 3) + (col & 7) * 2);
      v2 = (uint32_t)(tile_off<D>(row + 8, col >> 3) + (col & 7) * 2);
    }
    auto vread = [&](uint32_t vb1, uint32_t vb2, int j) {
      const int cc = j >> 2, i = j & 3;
      s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)((vb1 ^ (uint32_t)(i << 6)) + cc * 4096));
      s16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)((vb2 ^ (uint32_t)(i << 6)) + cc * 4096));
      s16x8_t r = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8_t, r);
    };
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // the prologue barrier: K(0), K(1), K(2) are there
    for (int t = 0; t <= nkv; ++t) {
      sb();
      // V(t-1) (issued first in iteration t-1) and K(t+1) have landed for this wave: all but the four newest pieces, K(t+2)'s
      if (t < nkv) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      sb();
      stamp(4 + 3 * t);
      if (t < nkv) {
        refill(planv, vbase + t * vtile_step, lds0 + V_OFF + (t & 1) * TB + pair * 4096);                                   // V(t)   -> the stage V(t-2) left
        refill(plank, kbase + min(t + 3, nkv - 1) * ktile_step, lds0 + ((t + 3) % NKST) * TB + pair * 4096);               // K(t+3) -> the stage K(t) left
      }
      stamp(5 + 3 * t);
      if (t >= 1) {
        const int j = t - 1, par = j & 1;
        // rescale record of tile j
        const unsigned fl = __builtin_amdgcn_readfirstlane(lds_ld<uint32_t>(flbase + par * 16));
        if (fl) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (fl & (1u << q)) {
              const float al = lds_ld<float>(albase + par * 2048 + q * 256);
#pragma unroll
              for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[q][i][r] *= al;
            }
        }
        uint32_t vb1 = lds0 + V_OFF + par * TB + v1, vb2 = lds0 + V_OFF + par * TB + v2, xr = xbase + par * 8192;
        asm volatile("" : "+v"(vb1), "+v"(vb2), "+v"(xr));
        bf16x8_t pfr[2][4], vfr[NFR];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) pfr[q][cc] = lds_ld<bf16x8_t>(xr + (q * 4 + cc) * 1024);
#pragma unroll
        for (int jj = 0; jj < AHEAD; ++jj) vfr[jj] = vread(vb1, vb2, jj);
        sb();
        stamp(6 + 3 * t);
#pragma unroll
        for (int mi = 0; mi < 32; ++mi) {
          const int jf = mi >> 1, cc = mi >> 3, i = (mi >> 1) & 3, q = mi & 1;
          oT[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[jf % NFR], pfr[q][cc], oT[q][i], 0, 0, 0);
          sb();
          if (q == 1 && jf + AHEAD < 16) vfr[(jf + AHEAD) % NFR] = vread(vb1, vb2, jf + AHEAD);
          sb();
        }
      }
    }
    // ---- epilogue: O^T / l -> bf16 rows through this pair's 16 KiB of the exchange buffer (free: P(nkv-1) has been consumed by this very
    // wave), then whole-row global stores.  The row sums were written by the score wave before the last barrier.
    char* Ow = smem + X_OFF + pair * 16384;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float inv = lds_ld<float>(lfbase + q * 256);
      const int row = q * 32 + l31;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int slot = i * 4 + rg;
          *reinterpret_cast<uint2*>(Ow + row * 256 + ((slot ^ (row & 15)) << 4) + hi * 8) =
              make_uint2(pack2bf(oT[q][i][rg * 4] * inv, oT[q][i][rg * 4 + 1] * inv), pack2bf(oT[q][i][rg * 4 + 2] * inv, oT[q][i][rg * 4 + 3] * inv));
        }
    }
    // (each wave reads back only what it wrote: no barrier needed, the compiler orders this wave's LDS writes before its reads)
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int row = p * 4 + (lane >> 4), slot = lane & 15;
      const uint4 v = *reinterpret_cast<const uint4*>(Ow + row * 256 + ((slot ^ (row & 15)) << 4));
      if (q0w + row < L) *reinterpret_cast<uint4*>(a.out + (rowbase + q0w + row) * a.out_stride + h * D + slot * 8) = v;
    }
  }
}
}  // namespace ws64
}  // namespace

// forward at head dim 128, no document mask, L % 128 == 0, o_stride % 8 == 0 (called from attention_w64.hip's launcher)
void udm_launch_attn_fwd_ws64(const void* args, hipStream_t stream) {
  using namespace ws64;
  const AttnArgs& a = *reinterpret_cast<const AttnArgs*>(args);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_ws64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_fwd_ws64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    once = true;
  }
  dim3 grid(((a.L + BQW - 1) / BQW) * a.H * a.B);
  if (a.timeline) hipLaunchKernelGGL(attn_fwd_ws64_kernel<true>, grid, dim3(512), LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(attn_fwd_ws64_kernel<false>, grid, dim3(512), LDS_BYTES, stream, a);
}

// C entry of the experiments library (scripts/bench_attn_w64.py, scripts/attn_ws64_timeline.py): same arguments as udm_attention_fwd without the mask
extern "C" int udm_exp_attention_fwd_ws64(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t H, int64_t L, int64_t q_stride,
                                          int64_t k_stride, int64_t v_stride, int64_t o_stride, uint64_t* timeline, hipStream_t stream) {
  UDM_CHECK_ARG(q && k && v && o && lse && B > 0 && H > 0 && L >= 128 && L % 128 == 0 && o_stride % 8 == 0, "udm_exp_attention_fwd_ws64: bad arguments");
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse;
  a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride; a.out_stride = o_stride;
  a.B = (int)B; a.H = (int)H; a.L = (int)L;
  a.scale = 1.0f / sqrtf(128.0f);
  a.scale_log2 = a.scale * 1.4426950408889634f;
  a.timeline = reinterpret_cast<unsigned long long*>(timeline);
  udm_launch_attn_fwd_ws64(&a, stream);
  UDM_CHECK_LAUNCH("udm_exp_attention_fwd_ws64");
  return 0;
}
