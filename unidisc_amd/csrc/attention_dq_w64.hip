// Attention backward, dQ, head dim 128, no document mask, L a multiple of 128: ONE WAVE PER SIMD, 64 queries per wave.
// (reference: the backward of flash_attn_qkvpacked_func / SDPA, models/dit.py:826-829, :843)
//
// Same register plan as the forward of attention_w64.hip.  A workgroup is four waves, one per SIMD, each owning two 32-query blocks (a, b):
//
//     accumulator file   dQ^T of both blocks (128, compiler-allocated) | Q fragments a[128:191] and dO fragments a[192:255] (named here: only ever
//                        MFMA B operands; see the ownership note in attention_w64.hip)
//     arch VGPRs         S^T and dP^T of two 32-key half tiles (2 x 64: the score MFMAs are inline asm in VGPR form), dS^T (16), fragments, addresses
//
// so every K / V fragment read from LDS feeds two MFMAs.  The key axis is walked in 32-key HALF tiles h (two per 64-key LDS tile), software-pipelined:
//
//     X(h)   32 MFMAs  S(h+1) = K Q^T, dP(h+1) = V dO^T          fillers: p = exp2(s c - lse), ds = p (dp - delta), bf16 packing of half h; K / V
//                                                                         fragment reads of half h+1
//     Y(h)   16 MFMAs  dQ^T += K(h)^T dS(h)^T                    fillers: transposing K reads, the refill of tile T+2 (LDS-DMA), first fragments of X(h+1)
//
// One workgroup barrier per 64-key tile; K and V tiles go through three-stage LDS rings (tile T+1 is read from the second half of tile T on, tile
// T+2 is refilled into the stage tile T-1 left).  The arithmetic is that of attn_bwd_dq_kernel step for step; delta = rowsum(dO o O) is computed
// and stored here as there (the dK/dV kernel reads it).  dQ leaves through LDS as whole rows.
#include "attention_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {
namespace dq64 {
constexpr int D = 128, KS = 8, DB = 4, BKV = 64, BQW = 256;
constexpr int TB = BKV * D * 2;          // one K or V tile, 16 KiB
constexpr int NST = 3;
constexpr int V_OFF = NST * TB;
constexpr int LDS_BYTES = 2 * NST * TB;

#define UDM_DQ_ACC_CLOBBERS                                                                                                                       \
  "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", \
      "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162",    \
      "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179",    \
      "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196",    \
      "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213",    \
      "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230",    \
      "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247",    \
      "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
constexpr int qreg(int q, int ks) { return 128 + (q * KS + ks) * 4; }
constexpr int doreg(int q, int ks) { return 192 + (q * KS + ks) * 4; }
template <int R>
__device__ __forceinline__ void to_acc(const bf16x8_t& v) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
               ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3)
               : UDM_DQ_ACC_CLOBBERS);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// gap marker: nothing is scheduled across it, memory operations keep their side of it, and the named accumulator registers stay ours
__device__ __forceinline__ void sb() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory", UDM_DQ_ACC_CLOBBERS);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma_piece(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// score MFMA in VGPR form, B operand = the named accumulator quad a[R:R+3]
template <int R, bool ZERO>
__device__ __forceinline__ void mfma_acc(f32x16_t& d, const bf16x8_t& a) {
  if (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(R), "i"(R + 3));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(R), "i"(R + 3));
}
// ... and, in the same statement, one unit of the softmax backward: two (s, dp) pairs -> ds = exp2(s c - lse) (dp - delta), packed to bf16.
// (volatile asm keeps its place in the instruction order, pure arithmetic does not; v_exp_f32 results are read three instructions later)
#define UDM_DS_PAIR_ASM(T0, T1, U0, U1, PW, S0, S1, P0, P1, C, LSE, DEL)                                                                       \
  "v_fma_f32 " T0 ", " S0 ", " C ", -" LSE "\n\tv_fma_f32 " T1 ", " S1 ", " C ", -" LSE "\n\tv_exp_f32 " T0 ", " T0 "\n\tv_exp_f32 " T1 ", " T1 "\n\t" \
  "v_sub_f32 " U0 ", " P0 ", " DEL "\n\tv_sub_f32 " U1 ", " P1 ", " DEL "\n\tv_mul_f32 " T0 ", " T0 ", " U0 "\n\tv_mul_f32 " T1 ", " T1 ", " U1 "\n\t"   \
  "v_cvt_pk_bf16_f32 " PW ", " T0 ", " T1
template <int R, bool ZERO>
__device__ __forceinline__ void mfma_acc_ds(f32x16_t& d, const bf16x8_t& a, float s0, float s1, float p0, float p1, float c, float lse, float del, uint32_t& pw) {
  float t0, t1, u0, u1;
  if (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %5, %6, a[%c14:%c15], 0\n\t" UDM_DS_PAIR_ASM("%0", "%1", "%2", "%3", "%4", "%7", "%8", "%9", "%10", "%11", "%12", "%13")
                 : "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1), "=&v"(pw), "=&v"(d)
                 : "v"(a), "v"(s0), "v"(s1), "v"(p0), "v"(p1), "s"(c), "v"(lse), "v"(del), "i"(R), "i"(R + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %5, %6, a[%c14:%c15], %5\n\t" UDM_DS_PAIR_ASM("%0", "%1", "%2", "%3", "%4", "%7", "%8", "%9", "%10", "%11", "%12", "%13")
                 : "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1), "=&v"(pw), "+v"(d)
                 : "v"(a), "v"(s0), "v"(s1), "v"(p0), "v"(p1), "s"(c), "v"(lse), "v"(del), "i"(R), "i"(R + 3));
}
__device__ __forceinline__ void ds_pair(float s0, float s1, float p0, float p1, float c, float lse, float del, uint32_t& pw) {
  float t0, t1, u0, u1;
  asm volatile(UDM_DS_PAIR_ASM("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "%8", "%9", "%10", "%11")
               : "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1), "=&v"(pw)
               : "v"(s0), "v"(s1), "v"(p0), "v"(p1), "s"(c), "v"(lse), "v"(del));
}

template <bool TL>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_w64_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K0 | K1 | K2 | V0 | V1 | V2
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tile-major 1-D grid as in attention.hip: all query blocks of one (b, h) run on one XCD (block id % 8) and share K / V through its L2
  int bh, tile_x;
  attn_block_to_work(blockIdx.x, a.B * a.H, bh, tile_x);
  const int b = bh / a.H, h = bh % a.H;
  const long rowbase = (long)b * a.L;
  const float c = a.scale_log2;
  const int L = a.L;
  const long ks_ = a.k_stride, vs_ = a.v_stride;   // row strides of K and V (the engine reads K from the roped [M, 2d] buffer, V from qkv [M, 3d])
  const int q0w = tile_x * BQW + wave * 64;    // first query of this wave
  const int nkv = L / BKV;                     // even, >= 2 (dispatch condition)
  const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;
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

  const bf16_t* kbase = a.k + rowbase * ks_ + h * D;
  const bf16_t* vbase = a.v + rowbase * vs_ + h * D;
  DmaPlan<D, BKV> plank, planv;
  plank.init(ks_, wave, lane);
  planv.init(vs_, wave, lane);
  const long ktile_step = (long)BKV * ks_, vtile_step = (long)BKV * vs_;
  auto refill = [&](int T) {   // K and V tile T into stage T % 3, this wave's four pieces of each
    const uint32_t dst = lds0 + (T % NST) * TB + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_piece(plank.off[j], kbase + T * ktile_step, dst + j * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_piece(planv.off[j], vbase + T * vtile_step, dst + V_OFF + j * 1024);
  };
  refill(0);
  refill(1);

  // Q, dO fragments -> named accumulator registers; lse; delta = rowsum(dO o O), stored for the dK/dV kernel (attn_bwd_dq_kernel does the same)
  float lse_q[2], delta_q[2];
  {
    bf16x8_t qf[2][KS], dof[2][KS];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int qi = q0w + q * 32 + l31;
      const bool ok = qi < L;
      float dl = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qf[q][ks] = load_frag_global(a.q + (rowbase + qi) * a.q_stride + h * D + ks * 16 + hi * 8, ok);
        dof[q][ks] = load_frag_global(a.dout + (rowbase + qi) * a.do_stride + h * D + ks * 16 + hi * 8, ok);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t of = load_frag_global(a.o + (rowbase + qi) * a.o_stride + h * D + ks * 16 + hi * 8, ok);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)of[e] * (float)dof[q][ks][e];
      }
      dl += __shfl_xor(dl, 32, 64);
      const long sidx = ((long)b * a.H + h) * L + qi;
      lse_q[q] = ok ? a.lse[sidx] : INFINITY;
      if (ok && hi == 0) const_cast<float*>(a.delta)[sidx] = dl;
      delta_q[q] = dl;
    }
    static_for<0, 2 * KS>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      to_acc<qreg(i / KS, i % KS)>(qf[i / KS][i % KS]);
      to_acc<doreg(i / KS, i % KS)>(dof[i / KS][i % KS]);
    });
  }

  f32x16_t dqT[2][DB];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqT[q][i][r] = 0.f;

  // Per-lane LDS byte offsets (as in attention_w64.hip).  Row fragment (ks) of half f: (base ^ (ks << 5)) + f * 8192, base = stage + kl.
  // K^T fragment (chunk cc, d block i), two transposing reads: (stage + t1) ^ (i << 6) + cc * 4096 and (stage + t2) ^ (i << 6) + cc * 4096.
  const uint32_t kl = (uint32_t)tile_off<D>(l31, hi);
  uint32_t t1, t2;
  {
    const int g1 = (lane >> 4) & 1, p = lane & 15;
    const int row = 4 * hi + (p >> 2), col = g1 * 16 + (p & 3) * 4;
    t1 = (uint32_t)(tile_off<D>(row, col >> 3) + (col & 7) * 2);
    t2 = (uint32_t)(tile_off<D>(row + 8, col >> 3) + (col & 7) * 2);
  }
  auto rowfrag = [&](uint32_t base, int ks, int f) { return lds_ld<bf16x8_t>((base ^ (uint32_t)(ks << 5)) + f * 8192); };
  auto ktfrag = [&](uint32_t b1, uint32_t b2, int cc, int i) {
    s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)((b1 ^ (uint32_t)(i << 6)) + cc * 4096));
    s16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)((b2 ^ (uint32_t)(i << 6)) + cc * 4096));
    s16x8_t r = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
  };

  f32x16_t SA[2], PA[2], SB[2], PB[2];   // S^T / dP^T of the even and the odd half tile, per query block
  bf16x8_t kfr[3], vfr[3], tfr[4];
  uint32_t dsw[2][2][4];                 // bf16 dS^T of the current half: [block][16-key chunk][packed pair]
  auto dsfrag = [&](int q, int c2) {
    uint4 u = make_uint4(dsw[q][c2][0], dsw[q][c2][1], dsw[q][c2][2], dsw[q][c2][3]);
    return __builtin_bit_cast(bf16x8_t, u);
  };

  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // tiles 0 and 1 have landed everywhere
  stamp(1);
  {   // half 0 (tile 0, f = 0), not pipelined
    const uint32_t kb = lds0 + kl, vb = lds0 + V_OFF + kl;
    static_for<0, KS>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      const bf16x8_t kf = rowfrag(kb, ks, 0), vf = rowfrag(vb, ks, 0);
      mfma_acc<qreg(0, ks), ks == 0>(SA[0], kf);
      mfma_acc<qreg(1, ks), ks == 0>(SA[1], kf);
      mfma_acc<doreg(0, ks), ks == 0>(PA[0], vf);
      mfma_acc<doreg(1, ks), ks == 0>(PA[1], vf);
    });
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j) { kfr[j] = rowfrag(kb, j, 1); vfr[j] = rowfrag(vb, j, 1); }   // first fragments of X(0): half 1 = tile 0, f = 1
  }

  // X(h): scores of the next half (tile Tn, row half fn) into (Sn, Pn) under the softmax backward of the current half (Sc, Pc) -> dsw
  auto phase_x = [&](auto has_next_t, f32x16_t (&Sc)[2], f32x16_t (&Pc)[2], f32x16_t (&Sn)[2], f32x16_t (&Pn)[2], uint32_t kb, uint32_t vb, int fn) {
    constexpr bool HAS_NEXT = decltype(has_next_t)::value;
    static_for<0, 32>([&](auto mi_) {
      constexpr int mi = decltype(mi_)::value, ks = mi >> 2, sub = mi & 3, q = sub & 1;
      constexpr bool has_u = (mi & 1) == 0;                                                       // unit u = mi / 2 of the current half
      constexpr int u = mi >> 1, uq = u >> 3, uc2 = (u >> 2) & 1, ue = u & 3, ur = 8 * uc2 + 2 * ue;
      if constexpr (HAS_NEXT) {
        if constexpr (sub < 2) {
          if constexpr (has_u) mfma_acc_ds<qreg(q, ks), ks == 0>(Sn[q], kfr[ks % 3], Sc[uq][ur], Sc[uq][ur + 1], Pc[uq][ur], Pc[uq][ur + 1], c, lse_q[uq], delta_q[uq], dsw[uq][uc2][ue]);
          else mfma_acc<qreg(q, ks), ks == 0>(Sn[q], kfr[ks % 3]);
        } else {
          if constexpr (has_u) mfma_acc_ds<doreg(q, ks), ks == 0>(Pn[q], vfr[ks % 3], Sc[uq][ur], Sc[uq][ur + 1], Pc[uq][ur], Pc[uq][ur + 1], c, lse_q[uq], delta_q[uq], dsw[uq][uc2][ue]);
          else mfma_acc<doreg(q, ks), ks == 0>(Pn[q], vfr[ks % 3]);
        }
      } else {
        if constexpr (has_u) ds_pair(Sc[uq][ur], Sc[uq][ur + 1], Pc[uq][ur], Pc[uq][ur + 1], c, lse_q[uq], delta_q[uq], dsw[uq][uc2][ue]);
      }
      sb();
      // fragments two k-steps ahead (K after the K pair, V after the V pair)
      if constexpr (HAS_NEXT && sub == 1 && ks + 2 < KS) kfr[(ks + 2) % 3] = rowfrag(kb, ks + 2, fn);
      if constexpr (HAS_NEXT && sub == 3 && ks + 2 < KS) vfr[(ks + 2) % 3] = rowfrag(vb, ks + 2, fn);
      sb();
    });
  };
  // Y(h): dQ^T += K^T dS^T over the two 16-key chunks of the half (chunks cc0, cc0 + 1 of the tile whose transposing-read bases are b1, b2)
  auto phase_y = [&](uint32_t b1, uint32_t b2, int cc0, auto&& filler) {
#pragma unroll
    for (int j = 0; j < 2; ++j) tfr[j] = ktfrag(b1, b2, cc0 + (j >> 2), j & 3);
    sb();
#pragma unroll
    for (int mi = 0; mi < 16; ++mi) {
      const int jf = mi >> 1, c2 = mi >> 3, i = (mi >> 1) & 3, q = mi & 1;
      dqT[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfr[jf % 4], dsfrag(q, c2), dqT[q][i], 0, 0, 0);
      sb();
      if (q == 1 && jf + 2 < 8) tfr[(jf + 2) % 4] = ktfrag(b1, b2, cc0 + ((jf + 2) >> 2), (jf + 2) & 3);
      filler(mi);
      sb();
    }
  };

  auto tile = [&](auto last_t, int T) {
    constexpr bool LAST = decltype(last_t)::value;
    sb();
    // tile T+1 (issued early in tile T-1) has landed for this wave; after the barrier: for all waves, and all are done with tile T-1
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    sb();
    stamp(2 + 2 * T);
    const int st = (T % NST) * TB, stn = ((T + 1) % NST) * TB;
    uint32_t kb = lds0 + st + kl, vb = lds0 + st + V_OFF + kl, kbn = lds0 + stn + kl, vbn = lds0 + stn + V_OFF + kl;
    uint32_t b1 = lds0 + st + t1, b2 = lds0 + st + t2;
    asm volatile("" : "+v"(kb), "+v"(vb), "+v"(kbn), "+v"(vbn), "+v"(b1), "+v"(b2));   // per-tile values: keeps XOR-ed addresses out of loop-invariant registers
    const bf16_t* ksrc = kbase + min(T + 2, nkv - 1) * ktile_step;   // tile T+2 -> the stage tile T-1 left (past the end: a harmless re-fetch)
    const bf16_t* vsrc = vbase + min(T + 2, nkv - 1) * vtile_step;
    const uint32_t dst = lds0 + ((T + 2) % NST) * TB + wave * 4096;
    // even half (rows 0..31 of tile T): scores of the odd half under its softmax backward, then its dQ update (with the refill of tile T+2: it
    // has the whole odd half to land)
    phase_x(std::true_type{}, SA, PA, SB, PB, kb, vb, 1);
    phase_y(b1, b2, 0, [&](int mi) {
      if (mi >= 1 && mi < 5) dma_piece(plank.off[mi - 1], ksrc, dst + (mi - 1) * 1024);
      if (mi >= 5 && mi < 9) dma_piece(planv.off[mi - 5], vsrc, dst + V_OFF + (mi - 5) * 1024);
      if (!LAST) {   // first fragments of the next X: tile T+1, rows 0..31
        if (mi == 12) kfr[0] = rowfrag(kbn, 0, 0);
        if (mi == 13) vfr[0] = rowfrag(vbn, 0, 0);
        if (mi == 14) kfr[1] = rowfrag(kbn, 1, 0);
        if (mi == 15) vfr[1] = rowfrag(vbn, 1, 0);
      }
    });
    stamp(3 + 2 * T);
    // odd half: scores of the even half of tile T+1 under its softmax backward, then its dQ update
    phase_x(std::integral_constant<bool, !LAST>{}, SB, PB, SA, PA, kbn, vbn, 0);
    phase_y(b1, b2, 2, [&](int mi) {
      if (!LAST) {   // first fragments of the next X: tile T+1, rows 32..63
        if (mi == 12) kfr[0] = rowfrag(kbn, 0, 1);
        if (mi == 13) vfr[0] = rowfrag(vbn, 0, 1);
        if (mi == 14) kfr[1] = rowfrag(kbn, 1, 1);
        if (mi == 15) vfr[1] = rowfrag(vbn, 1, 1);
      }
    });
  };
  {
    int T = 0;
    for (; T + 1 < nkv; ++T) tile(std::false_type{}, T);
    tile(std::true_type{}, T);
  }

  // ---- epilogue: scale dQ^T -> bf16 rows through this wave's 16 KiB of the (idle) LDS stages, then whole-row global stores
  stamp(62);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  char* Ow = smem + wave * (64 * D * 2);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = q * 32 + l31;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int slot = i * 4 + rg;
        *reinterpret_cast<uint2*>(Ow + row * 256 + ((slot ^ (row & 15)) << 4) + hi * 8) =
            make_uint2(pack2bf(dqT[q][i][rg * 4] * a.scale, dqT[q][i][rg * 4 + 1] * a.scale), pack2bf(dqT[q][i][rg * 4 + 2] * a.scale, dqT[q][i][rg * 4 + 3] * a.scale));
      }
  }
  // (each wave reads back only what it wrote: no barrier needed, the compiler orders this wave's LDS writes before its reads)
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int row = p * 4 + (lane >> 4), slot = lane & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(Ow + row * 256 + ((slot ^ (row & 15)) << 4));
    if (q0w + row < L) *reinterpret_cast<uint4*>(a.out + (rowbase + q0w + row) * a.out_stride + h * D + slot * 8) = v;
  }
  stamp(63);
}
}  // namespace dq64
}  // namespace

// dQ at head dim 128, no document mask, L % 128 == 0, dq_stride % 8 == 0 (called from attention.hip's launch_bwd)
void udm_launch_attn_bwd_dq_w64(const void* args, hipStream_t stream, unsigned long long* timeline) {
  using namespace dq64;
  AttnArgs a = *reinterpret_cast<const AttnArgs*>(args);
  a.timeline = timeline;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_w64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_w64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    once = true;
  }
  dim3 grid(((a.L + BQW - 1) / BQW) * a.H * a.B);
  if (timeline) hipLaunchKernelGGL(attn_bwd_dq_w64_kernel<true>, grid, dim3(256), LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(attn_bwd_dq_w64_kernel<false>, grid, dim3(256), LDS_BYTES, stream, a);
}
