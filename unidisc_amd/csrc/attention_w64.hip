// Attention forward, head dim 128, no document mask: ONE WAVE PER SIMD, 64 queries per wave.
// (reference: flash_attn_qkvpacked_func / SDPA, models/dit.py:826-829, :843)
//
// Why: the 8-wave kernel of attention.hip reads 1 KiB of K / V fragments from LDS per MFMA (32 queries per wave) and its two waves per SIMD
// share 512 registers, so nothing of a tile can be carried across a phase; it sits at 0.8 PF (0.31 of the MFMA peak).  Here a workgroup is
// four waves, one per SIMD, each owning TWO 32-query blocks (a, b) and the whole 512-register file:
//
//     accumulator file (AGPRs)   O^T of both blocks (128) and the Q fragments of both blocks (64, only ever MFMA B operands)
//     arch VGPRs                 S^T of tile t and of tile t+1 (2 x 64: the score MFMAs are inline asm in VGPR form, because the softmax reads
//                                them with the VALU), P (32), K / V fragments in flight, addresses
//
// so that every K / V fragment read from LDS feeds two MFMAs (512 B per MFMA) and the softmax of tile t runs under the MFMAs of its
// neighbours.  The tile loop is software-pipelined by hand, one workgroup barrier per 64-key tile:
//
//     phase A(t)   32 MFMAs  S(t+1) = K(t+1) Q^T          fillers: exp2 / row sums / bf16 packing of tile t (block a, first chunk of b),
//                                                                  K(t+1) fragment reads three fragments ahead, V(t) refill of the NEXT tile
//     phase B(t)   32 MFMAs  O^T += V(t)^T P(t)^T          fillers: rest of block b's exponentials (each 16-key chunk one chunk ahead of its
//                                                                  MFMAs), V(t) transposing reads, running maximum of S(t+1), K(t+3) refill,
//                                                                  the first K(t+2) fragments
//
// and the issue order is pinned with sched_barrier around every MFMA (about five single-issue instructions fit under one MFMA).  K tiles go
// through a three-stage LDS ring (a tile is readable one barrier before its phase, so its first fragments are requested under the tail of the
// previous phase), V tiles through two stages, both filled by LDS-DMA.  The arithmetic is that of attn_fwd_kernel step for step (same k order,
// same lazy-rescale decisions per 32-query block, same summation order): the two kernels give bit-identical O and LSE.
// O leaves through the (then idle) LDS stages so that global stores are whole 256-byte rows.
#include "attention_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {
namespace w64 {
constexpr int D = 128, KS = 8, DB = 4, BKV = 64, BQW = 256;
constexpr int TB = BKV * D * 2;          // one K or V tile, 16 KiB
constexpr int NKST = 3, NVST = 2;
constexpr int LDS_BYTES = (NKST + NVST) * TB;
constexpr int AHEAD = 3;                 // fragments requested ahead of their MFMAs
constexpr int NFR = 4;                   // fragment registers in rotation (AHEAD + 1)

// Register ownership.  The compiler will not keep a value that is only ever an inline-asm "a" operand in the accumulator file (it parks it in VGPRs /
// scratch and copies it into a temporary AGPR quad in front of every statement), so the Q fragments live in accumulator registers this file NAMES:
// a[192:255], block q / k-step ks at a[QREG(q, ks) : +3].  They are written once per block (q_to_acc) and read by the score MFMAs.  Every gap
// marker (sb) lists all 64 as clobbered, so the compiler cannot hold anything of its own in them across any gap of the loop (its accumulators O^T
// and its spill slots go to a[0:191]); nothing but these statements touches them.  Audit after edits: no compiler v_accvgpr_* on a192+ in the .s.
#define UDM_QACC_CLOBBERS                                                                                                                       \
  "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", \
      "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226",    \
      "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243",    \
      "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
constexpr int QREG0 = 192;
constexpr int qreg(int q, int ks) { return QREG0 + (q * KS + ks) * 4; }
template <int R>
__device__ __forceinline__ void q_to_acc(const bf16x8_t& v) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
               ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3)
               : UDM_QACC_CLOBBERS);
}
// score MFMAs in VGPR form (the compiler puts builtin MFMA accumulators into the accumulator file under a 512-register budget); B operand = Q at a[R:R+3]
template <int R, bool ZERO>
__device__ __forceinline__ void mfma_sq(f32x16_t& d, const bf16x8_t& a) {
  if (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(R), "i"(R + 3));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(R), "i"(R + 3));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// The softmax VALU work is inline asm statement by statement: pure arithmetic has no place in the compiler's instruction order (the DAG scheduler
// bunches it wherever register pressure looks best), volatile asm statements keep theirs.
__device__ __forceinline__ void max3(float& mx, float b, float c) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(b), "v"(c)); }
// two scores -> p = exp2(s c - mc), row-sum update, one packed bf16 pair.  (v_exp_f32 results are read one instruction later at the earliest:
// the transcendental-use wait state the compiler would otherwise insert)
__device__ __forceinline__ void softmax_pair(float s0, float s1, float c, float mc, float& psum, uint32_t& pw) {
  float t0, t1;
  asm volatile(
      "v_fma_f32 %0, %4, %6, -%7\n\t"
      "v_fma_f32 %1, %5, %6, -%7\n\t"
      "v_exp_f32 %0, %0\n\t"
      "v_exp_f32 %1, %1\n\t"
      "v_add_f32 %2, %2, %0\n\t"
      "v_add_f32 %2, %2, %1\n\t"
      "v_cvt_pk_bf16_f32 %3, %0, %1"
      : "=&v"(t0), "=&v"(t1), "+v"(psum), "=v"(pw)
      : "v"(s0), "v"(s1), "s"(c), "v"(mc));
}
#define UDM_SOFTMAX_PAIR_ASM(T0, T1, PS, PW, S0, S1, C, MC) \
  "v_fma_f32 " T0 ", " S0 ", " C ", -" MC "\n\tv_fma_f32 " T1 ", " S1 ", " C ", -" MC "\n\tv_exp_f32 " T0 ", " T0 "\n\tv_exp_f32 " T1 ", " T1 "\n\t" \
  "v_add_f32 " PS ", " PS ", " T0 "\n\tv_add_f32 " PS ", " PS ", " T1 "\n\tv_cvt_pk_bf16_f32 " PW ", " T0 ", " T1
// score MFMA and one softmax pair in ONE statement (the compiler pads a wait state between consecutive asm statements)
template <int R, bool ZERO>
__device__ __forceinline__ void mfma_sq_pair(f32x16_t& d, const bf16x8_t& a, float s0, float s1, float c, float mc, float& psum, uint32_t& pw) {
  float t0, t1;
  if (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %4, %5, a[%c10:%c11], 0\n\t" UDM_SOFTMAX_PAIR_ASM("%0", "%1", "%2", "%3", "%6", "%7", "%8", "%9")
                 : "=&v"(t0), "=&v"(t1), "+v"(psum), "=&v"(pw), "=&v"(d)
                 : "v"(a), "v"(s0), "v"(s1), "s"(c), "v"(mc), "i"(R), "i"(R + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %4, %5, a[%c10:%c11], %4\n\t" UDM_SOFTMAX_PAIR_ASM("%0", "%1", "%2", "%3", "%6", "%7", "%8", "%9")
                 : "=&v"(t0), "=&v"(t1), "+v"(psum), "=&v"(pw), "+v"(d)
                 : "v"(a), "v"(s0), "v"(s1), "s"(c), "v"(mc), "i"(R), "i"(R + 3));
}
// one softmax pair and two running-maximum updates (independent chains) in one statement
__device__ __forceinline__ void pair_max2(float s0, float s1, float c, float mc, float& psum, uint32_t& pw, float& mxa, float a0, float a1, float& mxb, float b0, float b1) {
  float t0, t1;
  asm volatile(UDM_SOFTMAX_PAIR_ASM("%0", "%1", "%2", "%3", "%6", "%7", "%8", "%9") "\n\tv_max3_f32 %4, %4, %10, %11\n\tv_max3_f32 %5, %5, %12, %13"
               : "=&v"(t0), "=&v"(t1), "+v"(psum), "=&v"(pw), "+v"(mxa), "+v"(mxb)
               : "v"(s0), "v"(s1), "s"(c), "v"(mc), "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}
__device__ __forceinline__ void max2(float& mxa, float a0, float a1, float& mxb, float b0, float b1) {
  asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %4, %5" : "+v"(mxa), "+v"(mxb) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}
__device__ __forceinline__ void pack_pair(float s0, float s1, uint32_t& pw) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw) : "v"(s0), "v"(s1)); }
__device__ __forceinline__ void dma_piece(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// gap marker: nothing is scheduled across it, memory operations keep their side of it, and the Q registers stay ours (see above)
__device__ __forceinline__ void sb() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory", UDM_QACC_CLOBBERS);
  __builtin_amdgcn_sched_barrier(0);
}

template <int ABL>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K0 | K1 | K2 | V0 | V1
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

  {
    bf16x8_t qf[2][KS];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int qi = q0w + q * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[q][ks] = load_frag_global(a.q + (rowbase + qi) * a.q_stride + h * D + ks * 16 + hi * 8, qi < L);
    }
    static_for<0, 2 * KS>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      q_to_acc<qreg(i / KS, i % KS)>(qf[i / KS][i % KS]);
    });
  }

  const bf16_t* kbase = a.k + rowbase * ks_ + h * D;
  const bf16_t* vbase = a.v + rowbase * vs_ + h * D;
  using Stg = DmaStager<D, BKV>;
  DmaPlan<D, BKV> plank, planv;
  plank.init(ks_, wave, lane);
  planv.init(vs_, wave, lane);
  const int nkv = (L + BKV - 1) / BKV;
  const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;
  const long ktile_step = (long)BKV * ks_, vtile_step = (long)BKV * vs_;   // elements per 64-row tile
  // ABL & 4: cycle stamps (s_memtime) of blocks 0 and 300, [block][wave][64 tags], written by lane 0
  auto stamp = [&](int tag) {
    if (ABL & 4) {
      if ((blockIdx.x == 0 || blockIdx.x == 300) && a.timeline && tag < 64) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tm = __builtin_amdgcn_s_memtime();
        if (lane == 0) a.timeline[((blockIdx.x ? 1 : 0) * 4 + wave) * 64 + tag] = tm;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  stamp(0);

  auto k_stage = [&](int t) { return (t % NKST) * TB; };
  auto v_stage = [&](int t) { return (NKST + (t & 1)) * TB; };
  auto refill_all = [&](int which, int t) {   // any tile, also the ragged last one (rows clamped to L - 1)
    const bf16_t* base = which ? vbase : kbase;
    char* dst = smem + (which ? v_stage(t) : k_stage(t));
    if ((t + 1) * BKV <= L) (which ? planv : plank).issue_full(base + t * (which ? vtile_step : ktile_step), dst, wave);
    else Stg::issue(base, which ? vs_ : ks_, t * BKV, L, dst, wave, lane);
  };

  refill_all(0, 0);
  refill_all(1, 0);
  if (nkv > 1) refill_all(0, 1);
  if (nkv > 2) refill_all(0, 2);
  f32x16_t oT[2][DB];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oT[q][i][r] = 0.f;
  // Reference exponent per query.  Query rows past L start at +inf: they never ask for a move (their own columns turn into NaN when a
  // neighbour does - lane-local and never stored), so the loop needs no validity mask.
  float m[2], mc[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) m[q] = (q0w + q * 32 + l31 < L) ? -INFINITY : INFINITY;

  // Per-lane LDS byte offsets.  K fragment (ks, f): row f*32 + l31, slot (2 ks + hi) ^ swz(row) - the XOR splits, so the offset is
  // kl ^ (ks << 5), + f * 8192.  V^T fragment (cc, i), two transposing reads: v1 ^ (i << 6) + cc * 4096 and v2 ^ (i << 6) + cc * 4096.
  const uint32_t kl = (uint32_t)tile_off<D>(l31, hi);
  uint32_t v1, v2;
  {
    const int g1 = (lane >> 4) & 1, p = lane & 15;
    const int row = 4 * hi + (p >> 2), col = g1 * 16 + (p & 3) * 4;
    v1 = (uint32_t)(tile_off<D>(row, col >> 3) + (col & 7) * 2);
    v2 = (uint32_t)(tile_off<D>(row + 8, col >> 3) + (col & 7) * 2);
  }
  auto kread = [&](uint32_t kb, int j) { return lds_ld<bf16x8_t>((kb ^ (uint32_t)((j >> 1) << 5)) + (j & 1) * 8192); };
  auto vread = [&](uint32_t vb1, uint32_t vb2, int j) {
    const int cc = j >> 2, i = j & 3;
    s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)((vb1 ^ (uint32_t)(i << 6)) + cc * 4096));
    s16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)((vb2 ^ (uint32_t)(i << 6)) + cc * 4096));
    s16x8_t r = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
  };

  // running maximum of one block's 32 scores in this lane, op k of 16 (two scores each)
  auto max_op = [&](const f32x16_t (&S)[2][2], int q, int k, float& mx) {
    const int f = k >> 3, r = (k & 7) * 2;
    max3(mx, S[q][f][r], S[q][f][r + 1]);
  };
  // lazy-rescale decision for the tile whose per-lane maxima are mx[] (attention.hip: the reference exponent only moves when some query of the
  // 32-query block saw a score more than 2^8 above it)
  auto decide = [&](const float (&mx)[2], float (&mloc)[2], unsigned long long (&bal)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      mloc[q] = fmaxf(mx[q], __shfl_xor(mx[q], 32, 64));
      bal[q] = __builtin_amdgcn_ballot_w64(mloc[q] * c > m[q] * c + 8.0f);
    }
  };
  auto apply_move = [&](const float (&mloc)[2], const unsigned long long (&bal)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (bal[q] != 0) {
        const float m_new = fmaxf(m[q], mloc[q]);
        const float alpha = __builtin_amdgcn_exp2f((m[q] - ((m_new == -INFINITY) ? 0.f : m_new)) * c);
        lsum[q] *= alpha;
        m[q] = m_new;
        // (the empty asm statements pin O^T in the accumulator file on both sides of the multiply: without them the compiler copies all 128
        // accumulators to VGPRs BEFORE the branch and back after the join, on every tile)
        asm volatile("" : "+a"(oT[q][0]), "+a"(oT[q][1]), "+a"(oT[q][2]), "+a"(oT[q][3]));
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oT[q][i][r] *= alpha;
        asm volatile("" : "+a"(oT[q][0]), "+a"(oT[q][1]), "+a"(oT[q][2]), "+a"(oT[q][3]));
      }
      mc[q] = (m[q] == -INFINITY) ? 0.f : m[q] * c;
    }
  };

  uint32_t Pw[2][4][4];   // bf16 P^T, [block][16-key chunk][packed pair]
  float psum[2];
  // one unit of the softmax of the current tile: two scores of block q, chunk cc -> one packed pair (2 fma, 2 exp2, 2 adds, 1 pack)
  auto unit = [&](const f32x16_t (&S)[2][2], int q, int cc, int e2) {
    const int f = cc >> 1, r0 = 8 * (cc & 1) + 2 * e2;
    if (ABL & 1) pack_pair(S[q][f][r0], S[q][f][r0 + 1], Pw[q][cc][e2]);
    else softmax_pair(S[q][f][r0], S[q][f][r0 + 1], c, mc[q], psum[q], Pw[q][cc][e2]);
  };
  auto pfrag = [&](int q, int cc) {
    uint4 u = make_uint4(Pw[q][cc][0], Pw[q][cc][1], Pw[q][cc][2], Pw[q][cc][3]);
    return __builtin_bit_cast(bf16x8_t, u);
  };

  bf16x8_t kfr[NFR], vfr[NFR];

  // ---- prologue: S(0), its maximum and reference exponent; first fragments of K(1)
  f32x16_t S0[2][2], S1[2][2];
  stamp(1);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // K(0), V(0), K(1), K(2) have landed everywhere
  stamp(2);
  {
    const uint32_t kb = lds0 + k_stage(0) + kl;
    static_for<0, 16>([&](auto j_) {
      constexpr int j = decltype(j_)::value, ks = j >> 1, f = j & 1;
      const bf16x8_t kf = kread(kb, j);
      mfma_sq<qreg(0, ks), ks == 0>(S0[0][f], kf);
      mfma_sq<qreg(1, ks), ks == 0>(S0[1][f], kf);
    });
    asm volatile("s_nop 15" ::: "memory");   // asm MFMA results -> first VALU reader: wait states the compiler does not know it owes
    float mx[2] = {-INFINITY, -INFINITY}, mloc[2];
    unsigned long long bal[2];
#pragma unroll
    for (int k = 0; k < 32; ++k) max_op(S0, k >> 4, k & 15, mx[k >> 4]);
    decide(mx, mloc, bal);
    apply_move(mloc, bal);
    if (nkv > 1) {
      const uint32_t kb1 = lds0 + k_stage(1) + kl;
#pragma unroll
      for (int j = 0; j < AHEAD; ++j) kfr[j] = kread(kb1, j);
    }
  }

  // ---- one tile: phases A and B.  Sc = scores of tile t (softmax input), Sn = scores of tile t+1 (written here when HAS_NEXT).
  // Refills are unconditional single pieces and the body is one basic block (but for the rare rescale at its end).
  auto body = [&](auto has_next_t, f32x16_t (&Sc)[2][2], f32x16_t (&Sn)[2][2], int t) {
    constexpr bool HAS_NEXT = decltype(has_next_t)::value, STEADY = HAS_NEXT;
    sb();
    // V(t) and K(t+2), both issued early in tile t-1, have landed for this wave; after the barrier: for all waves, and all are done with V(t-1), K(t).
    // (Everything a tile reads after its barrier - V(t), and K(t+2) for the early fragment reads at its end - was issued one tile ago, so the wait
    // cannot be a counted one with these ring depths; the refills go out in the first half of phase A and have the rest of the tile to land.)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    sb();
    stamp(4 + 2 * t);
    uint32_t kb = lds0 + k_stage(t + 1) + kl, kb2 = lds0 + k_stage(t + 2) + kl;
    uint32_t vb1 = lds0 + v_stage(t) + v1, vb2 = lds0 + v_stage(t) + v2;
    asm volatile("" : "+v"(kb), "+v"(kb2), "+v"(vb1), "+v"(vb2));   // per-tile values: keeps the XOR-ed fragment addresses out of loop-invariant registers
    // V(t+1) -> the stage V(t-1) left, K(t+3) -> the stage K(t) left.  Past the last tile the steady body re-fetches the last tile into a
    // stage nobody reads any more: same instruction stream (and the same counted waits) for every tile of the block.
    const bf16_t* vsrc = vbase + min(t + 1, nkv - 1) * vtile_step;
    const bf16_t* ksrc = kbase + min(t + 3, nkv - 1) * ktile_step;
    const uint32_t vdst = lds0 + v_stage(t + 1) + wave * 4096, kdst = lds0 + k_stage(t + 3) + wave * 4096;
    psum[0] = 0.f;
    psum[1] = 0.f;
    // ---------------- phase A
    if (HAS_NEXT) {
      static_for<0, 32>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value, j = mi >> 1, ks = mi >> 2, f = (mi >> 1) & 1, q = mi & 1;
        // softmax units of tile t: block a (16 units) then chunk 0 of block b (4 units), 20 units over the 32 gaps; a gap's unit shares the
        // statement of its MFMA
        constexpr int u = (mi * 5 + 7) / 8;                     // the unit with (u * 8) / 5 == mi, if any
        constexpr bool has_u = u < 20 && (u * 8) / 5 == mi;
        if constexpr (has_u && !(ABL & 1) && !(ABL & 16)) {
          constexpr int uq = u < 16 ? 0 : 1, ucc = u < 16 ? (u >> 2) : 0, ue = u < 16 ? (u & 3) : u - 16, uf = ucc >> 1, ur = 8 * (ucc & 1) + 2 * ue;
          mfma_sq_pair<qreg(q, ks), ks == 0>(Sn[q][f], kfr[j % NFR], Sc[uq][uf][ur], Sc[uq][uf][ur + 1], c, mc[uq], psum[uq], Pw[uq][ucc][ue]);
        } else {
          if (!(ABL & 16)) mfma_sq<qreg(q, ks), ks == 0>(Sn[q][f], kfr[j % NFR]);
          if constexpr (has_u) { if (u < 16) unit(Sc, 0, u >> 2, u & 3); else unit(Sc, 1, 0, u - 16); }
        }
        sb();
        if (!(ABL & 8) && q == 1 && j + AHEAD < 16) kfr[(j + AHEAD) % NFR] = kread(kb, j + AHEAD);
        // refills, early in the tile: V(t+1) at gaps 1, 3, 5, 7 and K(t+3) at gaps 9, 11, 13, 15
        if (!(ABL & 2) && (mi & 1) && mi < 8) dma_piece(planv.off[mi >> 1], vsrc, vdst + (mi >> 1) * 1024);
        if (!(ABL & 2) && (mi & 1) && mi >= 8 && mi < 16) dma_piece(plank.off[(mi - 8) >> 1], ksrc, kdst + ((mi - 8) >> 1) * 1024);
        // first V(t) fragments for phase B
        if constexpr (mi >= 32 - AHEAD) { if (!(ABL & 8)) vfr[mi - (32 - AHEAD)] = vread(vb1, vb2, mi - (32 - AHEAD)); }
        sb();
      });
    } else {
#pragma unroll
      for (int u = 0; u < 20; ++u) { if (u < 16) unit(Sc, 0, u >> 2, u & 3); else unit(Sc, 1, 0, u - 16); }
#pragma unroll
      for (int j = 0; j < AHEAD; ++j) vfr[j] = vread(vb1, vb2, j);
      sb();
    }
    stamp(5 + 2 * t);
    // ---------------- phase B
    float mx[2] = {-INFINITY, -INFINITY}, mloc[2];
    unsigned long long bal[2] = {0, 0};
#pragma unroll
    for (int mi = 0; mi < 32; ++mi) {
      const int j = mi >> 1, cc = mi >> 3, i = (mi >> 1) & 3, q = mi & 1;
      if (!(ABL & 32)) oT[q][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[j % NFR], pfrag(q, cc), oT[q][i], 0, 0, 0);
      sb();
      if (!(ABL & 8) && q == 1 && j + AHEAD < 16) vfr[(j + AHEAD) % NFR] = vread(vb1, vb2, j + AHEAD);
      // block b, chunks 1..3: chunk cc+1 is finished under the MFMAs of chunk cc (units at even gaps 0..22); running maximum of S(t+1): two
      // updates (one per block) in each of the gaps 2..17; unit and updates of a gap in one statement
      {
        const bool has_u = mi < 24 && (mi & 1) == 0, has_m = HAS_NEXT && !(ABL & 64) && mi >= 2 && mi < 18;
        const int u = mi >> 1, ucc = 1 + (u >> 2), ue = u & 3, uf = ucc >> 1, ur = 8 * (ucc & 1) + 2 * ue;
        const int k = mi - 2, kf = k >> 3, kr = (k & 7) * 2;
        if (has_u && has_m && !(ABL & 1))
          pair_max2(Sc[1][uf][ur], Sc[1][uf][ur + 1], c, mc[1], psum[1], Pw[1][ucc][ue], mx[0], Sn[0][kf][kr], Sn[0][kf][kr + 1], mx[1], Sn[1][kf][kr], Sn[1][kf][kr + 1]);
        else {
          if (has_u) unit(Sc, 1, ucc, ue);
          if (has_m) max2(mx[0], Sn[0][kf][kr], Sn[0][kf][kr + 1], mx[1], Sn[1][kf][kr], Sn[1][kf][kr + 1]);
        }
      }
      if (mi == 23) { lsum[0] += psum[0]; lsum[1] += psum[1]; }
      if (HAS_NEXT && mi == 26) decide(mx, mloc, bal);
      // first fragments of K(t+2) (landed before this tile's barrier)
      if (STEADY && !(ABL & 8) && mi >= 32 - AHEAD) kfr[mi - (32 - AHEAD)] = kread(kb2, mi - (32 - AHEAD));
      sb();
    }
    if (HAS_NEXT) apply_move(mloc, bal);
    sb();
  };

  {
    const std::true_type T{};
    const std::false_type F{};
    // L is a multiple of 128 (dispatch condition): an even number of full tiles.  Every tile but the last runs the same instruction stream.
    int t = 0;
    for (; t + 2 < nkv; t += 2) {
      body(T, S0, S1, t);
      body(T, S1, S0, t + 1);
    }
    body(T, S0, S1, t);
    body(F, S1, S0, t + 1);
  }

  // ---- epilogue: O^T / l -> bf16 rows through this wave's 16 KiB of the (idle) LDS stages, then whole-row global stores
  stamp(60);
  __syncthreads();
  stamp(61);
  char* Ow = smem + wave * (64 * D * 2);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float ltot = lsum[q] + __shfl_xor(lsum[q], 32, 64);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    const int row = q * 32 + l31;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int slot = i * 4 + rg;
        *reinterpret_cast<uint2*>(Ow + row * 256 + ((slot ^ (row & 15)) << 4) + hi * 8) =
            make_uint2(pack2bf(oT[q][i][rg * 4] * inv, oT[q][i][rg * 4 + 1] * inv), pack2bf(oT[q][i][rg * 4 + 2] * inv, oT[q][i][rg * 4 + 3] * inv));
      }
    const int qi = q0w + row;
    if (qi < L && hi == 0) a.lse[((long)b * a.H + h) * L + qi] = ltot > 0.f ? __builtin_fmaf(m[q], c, log2f(ltot)) : INFINITY;
  }
  stamp(62);
  // (each wave reads back only what it wrote: no barrier needed, the compiler orders this wave's LDS writes before its reads)
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int row = p * 4 + (lane >> 4), slot = lane & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(Ow + row * 256 + ((slot ^ (row & 15)) << 4));
    if (q0w + row < L) *reinterpret_cast<uint4*>(a.out + (rowbase + q0w + row) * a.out_stride + h * D + slot * 8) = v;
  }
  stamp(63);
}

int g_enabled = -1;
unsigned long long* g_timeline = nullptr;
}  // namespace w64
}  // namespace

extern "C" int udm_attention_w64_timeline(uint64_t* buf) {   // diagnostics: device buffer of 2 x 4 x 64 cycle stamps filled by the next launches (null = off)
  w64::g_timeline = reinterpret_cast<unsigned long long*>(buf);
  return 0;
}
extern "C" int udm_attention_set_w64(int enable) {
  w64::g_enabled = enable ? 1 : 0;   // 0 = off, 1 = the one-wave-per-SIMD forward / dQ kernels
  return 0;
}

int udm_attn_w64_mode() {
  if (w64::g_enabled < 0) { const char* e = getenv("UDM_ATTN_W64"); w64::g_enabled = e ? atoi(e) : 0; }
  return w64::g_enabled;
}
unsigned long long* udm_attn_w64_timeline() { return w64::g_timeline; }

// forward at head dim 128 without a document mask (called from attention.hip's dispatch); returns false when the kernel is switched off
bool udm_launch_attn_fwd_w64(const void* args, hipStream_t stream) {
  using namespace w64;
  if (g_enabled < 0) { const char* e = getenv("UDM_ATTN_W64"); g_enabled = e ? atoi(e) : 0; }
  if (!g_enabled) return false;
  AttnArgs a = *reinterpret_cast<const AttnArgs*>(args);
  a.timeline = g_timeline;
  static const int abl = [] { const char* e = getenv("UDM_ATTN_W64_ABL"); return e ? atoi(e) : 0; }();   // timing-only ablations (wrong results)
  dim3 grid(((a.L + BQW - 1) / BQW) * a.H * a.B);
  const int v = (abl & ~4) | (g_timeline ? 4 : 0);
#define UDM_W64_CASE(V)                                                                                                          \
  case V: {                                                                                                                      \
    static bool once = false;                                                                                                    \
    if (!once) { (void)hipFuncSetAttribute((const void*)attn_fwd_w64_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); once = true; } \
    hipLaunchKernelGGL(attn_fwd_w64_kernel<V>, grid, dim3(256), LDS_BYTES, stream, a);                                           \
  } break;
  switch (v) {
    UDM_W64_CASE(0) UDM_W64_CASE(4)
#ifdef UDM_W64_ABLATIONS
    UDM_W64_CASE(1) UDM_W64_CASE(2) UDM_W64_CASE(5) UDM_W64_CASE(12) UDM_W64_CASE(13) UDM_W64_CASE(20) UDM_W64_CASE(36) UDM_W64_CASE(68) UDM_W64_CASE(52) UDM_W64_CASE(61) UDM_W64_CASE(125)
#endif
    default: udm_set_error("udm_attention_fwd: ablation %d not built", v); return true;
  }
#undef UDM_W64_CASE
  return true;
}
