// "Quad" bf16 MFMA GEMM for gfx950: ONE wave per SIMD.
//
//   (64 FM) x 256 x 64 block tile, FOUR waves as 2 x 2, each wave (32 FM) x 128 as FM x 4 v_mfma_f32_32x32x16_bf16 accumulators
//   (FM = 3 / 4 / 5: 192 / 256 / 320 accumulator registers out of the 512 a lone wave owns).
//
// Why: in the 8-wave kernels of gemm.hip a wave owns 160 x 64 or 128 x 64 outputs and needs one 1 KiB LDS fragment per
// 1.4 MFMAs (NT) or two transposing reads per fragment (TN: 12 LDS instructions per 8 MFMAs) - its own instruction stream (fragment
// reads + LDS-DMA issue + barrier hand-offs) is LONGER than its share of the matrix pipe, so the pipe idles (0.64 NT / 0.46-0.55 TN
// measured).  A 128-column wave tile needs (FM + 4) fragments per 4 FM MFMAs; with one wave per SIMD nothing but that wave's own
// stream hides latency, so the K loop is software-pipelined by hand and the issue order is pinned:
//   * fragments of k-step s+1 are read into the other register buffer in the shadow of the MFMAs of k-step s;
//   * the single barrier of a K tile sits BEFORE the last k-step's MFMAs (whose fragments are already in registers): the barrier wait,
//     the first fragment reads of the next tile and the LDS-DMA refills of the stage just retired all run under matrix work;
//   * refills (global_load_lds_dwordx4, 1 KiB per instruction, source-side swizzle so the LDS image stays lane-linear) are spread one
//     per DMA_GAP MFMAs over the following MFMAs instead of going out in a burst.
// TN = false: C[M,N] = A[M,K] B[N,K]^T, both operands K-contiguous (forward / dgrad): tiles are [rows][64 k] (128-byte rows, XOR-swizzled
//   16-byte k-slots, ds_read_b128 fragments) exactly as in gemm.hip.
// TN = true:  C[M,N] = A[K,M]^T B[K,N], both operands K-major (wgrad dW = dY^T X read in place): tiles are [64 k-rows][columns], MFMA
//   operands are gathered with ds_read_b64_tr_b16; the four k-rows one transposing read touches are spread over the 64 LDS banks by
//   XOR-ing the 64-byte column granule with the k-row (rows of 256 / 512 bytes) or with k-row >> 1 (rows of 384 / 640 bytes, which
//   alternate between bank offsets 0 and 128).
// Whole tiles only (M % (64 FM) == 0, N % 256 == 0, K % 64 == 0, K >= 128): the dispatcher in gemm.hip keeps the 8-wave kernels for
// everything else.  Split-K (TN, fp32 output) through a workspace like the 8-wave form.
#include "common.h"
#include "gemm_quad.h"
#include "gemm_loop_gen.h"
#include "../../include/unidisc_hip.h"

#include <stdlib.h>
#include <type_traits>

namespace {
using namespace udm;
constexpr int BK = 64, GROUP_M_DEFAULT = 8;

// one LDS-DMA piece: lane i of the wave lands at lds_dst + 16 i; source = wave-uniform 64-bit base (SGPR pair) + this lane's 32-bit byte offset
__device__ __forceinline__ void dma16(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ const char* uniform_ptr(const char* ptr) {   // pin a wave-uniform pointer into an SGPR pair
  const uint64_t u = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi32 << 32) | lo);
}

// ABL (timing-only ablations, WRONG results; env UDM_QUAD_ABL): bit 0 = no refills after the prologue, bit 1 = the boundary does not wait for the refills,
// bit 2 = no L2 touch-ahead
// MODE: 0 = NT (C = A B^T: both operands K-contiguous), 1 = TN (C = A^T B: both operands K-major, rows = contraction index: the wgrad form),
// 2 = NN (C = A B: A K-contiguous, B K-major - the dgrad form dX = dY W read from the forward's own W shadow, so no W^T shadow has to be cast).
// Every operand is staged and gathered by its own layout (TA / TB): NN is the NT kernel's A side next to the TN kernel's B side.
// RAGGED (NT / NN forms): the last tile row may hang over M - its operand rows are clamped to M - 1 at the LDS-DMA source and its output rows are not stored.  A row count
// that no whole tile height divides into one round (config E: M = 9216 = 48 x 192 -> 384 tiles, 1.5 rounds) then still runs as ONE round of 320-row tiles (29 x 8 = 232).
// ASMLOOP (NT form, whole tiles, an even number >= 4 of K tiles, no split-K): the prologue and the K loop below are replaced by ONE hand-scheduled asm statement
// (asmgen/gemm_loop.py: same LDS image, same k order per output - bit-identical accumulators; counted lgkmcnt waits per fragment, every instruction placed).
// ASMLOOP = 2: the same statement built on v_mfma_f32_16x16x32_bf16 (the matrix pipe sustains 12 % more on random operands in that form: the loop is power bound).
// A 32 x 32 accumulator block then holds four 16 x 16 sub-blocks of 4 registers (the epilogue's patch store maps them); sums of 32 products per instruction instead
// of 16: equal to the other loops up to fp32 rounding, not bit for bit.
template <int FM, int MODE, int EPI, bool OUT_F32, int DMA_GAP, int ABL = 0, bool RAGGED = false, int ASMLOOP = 0>
__global__ __launch_bounds__(256, 1) void gemm_quad_kernel(QuadArgs p0) {
  QuadArgs p = p0;   // (the paired launch redirects the operand fields of the blocks that belong to the second problem, once, before anything reads them)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool TA = MODE == 1, TB = MODE >= 1;
  constexpr int FN = 4, BM = 64 * FM, BN = 256;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int A_PW = A_BYTES / 1024 / 4, B_PW = B_BYTES / 1024 / 4, LOADS = A_PW + B_PW;   // 1 KiB pieces per wave per K tile
  constexpr int RB_A = TA ? BM * 2 : 128, RB_B = TB ? BN * 2 : 128;                           // LDS row bytes
  constexpr int NMF = FM * FN;                                                                 // MFMAs per k-step
  constexpr int NRD_A = TA ? 2 * FM : FM, NRD_B = TB ? 2 * FN : FN, NRD = NRD_A + NRD_B;       // fragment reads per k-step
  // LDS map: [A stage 0][A stage 1][B stage 0][B stage 1]
  constexpr int A0 = 0, B0 = 2 * A_BYTES;
  static_assert(2 * (A_BYTES + B_BYTES) <= 160 * 1024, "tile does not fit the LDS");

  const int S = p.splitk > 1 ? p.splitk : 1;
  const int nwg = p.tiles_m * p.tiles_n;
  int pid = xcd_remap(blockIdx.x, nwg * S);
  const int slice = pid % S;
  pid /= S;
  const int grp_rows = p.group_m > 0 ? p.group_m : GROUP_M_DEFAULT;
  const int per_group = grp_rows * p.tiles_n;
  const int grp = pid / per_group, first_m = grp * grp_rows;
  const int gsz = min(p.tiles_m - first_m, grp_rows);
  int tm = first_m + (pid % per_group) % gsz;
  int tn = (pid % per_group) / gsz;
  if (p.tiles_m_split > 0 && tm >= p.tiles_m_split) {   // block-uniform
    tm -= p.tiles_m_split;
    p.A = p.A2; p.B = p.B2; p.C = p.C2; p.lda = p.lda2; p.ldb = p.ldb2; p.ldc = p.ldc2;
  }
  if (MODE == 1 && p0.nprob > 0) {   // multi-problem launch (block-uniform, scalar): which problem does tile `pid` belong to, and where inside it
    // (selected by an unrolled compare chain on the kernel ARGUMENT: indexing a local copy with a run-time index would send the whole struct to private memory)
    int start = 0, end = p0.m_tile_start[1], tnn = p0.m_tiles_n[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      if (i < p0.nprob && pid >= p0.m_tile_start[i]) {
        start = p0.m_tile_start[i]; end = p0.m_tile_start[i + 1]; tnn = p0.m_tiles_n[i];
        p.A = p0.mA[i]; p.B = p0.mB[i]; p.C = p0.mC[i]; p.lda = p0.mlda[i]; p.ldb = p0.mldb[i]; p.ldc = p0.mldc[i];
      }
    }
    const int local = pid - start, rows_i = (end - start) / tnn;
    tm = local % rows_i;   // column-major inside a problem: neighbouring blocks share the B panel
    tn = local / rows_i;
  }
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;

  // ---- LDS-DMA sources: per piece a 32-bit lane offset from the operand's tile base; the base advances by one K tile per iteration ----
  uint32_t offa[A_PW], offb[B_PW];
  {
    const int lrow = lane >> 3, lslot = lane & 7;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
      if (!TA) {
        const int r = (wave * A_PW + j) * 8 + lrow;
        const int rs = RAGGED ? min(r, p.M - 1 - row0) : r;   // source row (the swizzle follows the LDS row r)
        offa[j] = (uint32_t)(((long)rs * p.lda + ((lslot ^ ((r >> 1) & 7)) << 3)) * 2);
      } else {
        const int off = (wave * A_PW + j) * 1024 + lane * 16;
        const int row = off / RB_A, within = off % RB_A;
        const int xr = (FM % 2 == 0) ? (row & 3) : ((row & 3) >> 1);
        const int col = (((within >> 6) ^ xr) << 5) + ((within & 63) >> 1);
        offa[j] = (uint32_t)(((long)row * p.lda + col) * 2);
      }
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
      if (!TB) {
        const int r = (wave * B_PW + j) * 8 + lrow;
        offb[j] = (uint32_t)(((long)r * p.ldb + ((lslot ^ ((r >> 1) & 7)) << 3)) * 2);
      } else {
        const int off = (wave * B_PW + j) * 1024 + lane * 16;
        const int row = off / RB_B, within = off % RB_B;
        const int col = (((within >> 6) ^ (row & 3)) << 5) + ((within & 63) >> 1);
        offb[j] = (uint32_t)(((long)row * p.ldb + col) * 2);
      }
    }
  }
  const int nk_all = p.K / BK;
  const int kt0 = (int)((long)nk_all * slice / S), kt1 = (int)((long)nk_all * (slice + 1) / S);
  const int nk = kt1 - kt0;
  const long kstep_a = TA ? (long)BK * p.lda * 2 : BK * 2, kstep_b = TB ? (long)BK * p.ldb * 2 : BK * 2;   // bytes per K tile
  // running tile bases (wave-uniform, SGPR pairs): ap / bp = the K tile being COMPUTED; refills read one or two tiles ahead of it
  const char* ap = uniform_ptr(reinterpret_cast<const char*>(TA ? p.A + row0 : p.A + (long)row0 * p.lda) + kt0 * kstep_a);
  const char* bp = uniform_ptr(reinterpret_cast<const char*>(TB ? p.B + col0 : p.B + (long)col0 * p.ldb) + kt0 * kstep_b);
  const uint32_t dsta = lds0 + A0 + wave * A_PW * 1024, dstb = lds0 + B0 + wave * B_PW * 1024;
  auto dma_piece = [&](int ahead, int stage, int j) {   // the tile `ahead` K tiles after the current one into `stage`; j < A_PW: A piece j, else B piece j - A_PW
    if ((ABL & 1) && ahead == 2) return;
    if (j < A_PW) dma16(ap + ahead * kstep_a, offa[j], dsta + stage * A_BYTES + j * 1024);
    else dma16(bp + ahead * kstep_b, offb[j - A_PW], dstb + stage * B_BYTES + (j - A_PW) * 1024);
  };

  // ---- fragment addresses ----
  bf16x8_t fa[2][FM], fb[2][FN];                      // K-contiguous operand: whole fragments (one ds_read_b128 each)
  s16x4_t ha[2][FM][2], hb[2][FN][2];                  // K-major operand: fragment halves (k +0..3 / +4..7), one transposing read each
  uint32_t fa_addr[TA ? FM : 1], fb_addr[TB ? FN : 1];
  const int p16 = lane & 15, g1 = (lane >> 4) & 1, rot = p16 >> 2;   // K-major gather: lane supplies k-row rot (+4), 4 columns of its 16-lane group
  if (!TA) {
    fa_addr[0] = lds0 + A0 + (wm * 32 * FM + l31) * 128;
  } else {
    const int xra = (FM % 2 == 0) ? rot : (rot >> 1);
#pragma unroll
    for (int i = 0; i < FM; ++i) fa_addr[i] = lds0 + A0 + (hi * 8 + rot) * RB_A + (((wm * FM + i) ^ xra) << 6) + g1 * 32 + (p16 & 3) * 8;
  }
  if (!TB) {
    fb_addr[0] = lds0 + B0 + (wn * 128 + l31) * 128;
  } else {
#pragma unroll
    for (int j = 0; j < FN; ++j) fb_addr[j] = lds0 + B0 + (hi * 8 + rot) * RB_B + (((wn * FN + j) ^ rot) << 6) + g1 * 32 + (p16 & 3) * 8;
  }
  const int sw = (l31 >> 1) & 7;
  // read n of k-step kk (stage st) into register buffer buf, in the order the MFMAs need them: the FN B fragments (every MFMA row uses all
  // of them), then the A fragments row by row.  NT: one ds_read_b128 per fragment.  TN: two transposing reads per fragment (k-rows +0..3, +4..7).
  auto read_one = [&](int st, int kk, int buf, int n) {
    const uint32_t so = (uint32_t)(((kk * 2 + hi) ^ sw) << 4);
    if (n < NRD_B) {   // the B fragments first
      if (!TB) fb[buf][n] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fb_addr[0] + st * B_BYTES + n * 4096 + so));
      else hb[buf][n >> 1][n & 1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)(fb_addr[TB ? (n >> 1) : 0] + st * B_BYTES + (kk * 16 + (n & 1) * 4) * RB_B));
    } else {
      const int m = n - NRD_B;
      if (!TA) fa[buf][m] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fa_addr[0] + st * A_BYTES + m * 4096 + so));
      else ha[buf][m >> 1][m & 1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(size_t)(fa_addr[TA ? (m >> 1) : 0] + st * A_BYTES + (kk * 16 + (m & 1) * 4) * RB_A));
    }
  };
  auto frag_a = [&](int buf, int i) -> bf16x8_t {
    if (!TA) return fa[buf][i];
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(ha[buf][i][0], ha[buf][i][1], 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto frag_b = [&](int buf, int j) -> bf16x8_t {
    if (!TB) return fb[buf][j];
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(hb[buf][j][0], hb[buf][j][1], 0, 1, 2, 3, 4, 5, 6, 7));
  };

  // Accumulators are zeroed BY the matrix pipe (MFMA of an opaque zero fragment onto the constant 0): hundreds of v_mov + copies into the
  // accumulator file would put that many live VGPRs here and make the allocator spill the loop's addresses and fragments.
  // The compiler puts every builtin MFMA's accumulator into the 256-register accumulator file (all or nothing per kernel), which holds four
  // rows of fragments.  A fifth row (FM = 5: 320-row tiles) lives in 64 ARCH VGPRs instead and is driven by inline-asm MFMAs in VGPR form;
  // their only readers are the next MFMA of the same chain (no wait states needed) and the epilogue (far behind the last one).
  constexpr int FMA = FM < 4 ? FM : 4;   // rows in the accumulator file
  f32x16_t acc[FM][FN];
  {
    bf16x8_t zf = {};
    asm volatile("" : "+v"(zf));
    const f32x16_t zc = {};
#pragma unroll
    for (int i = 0; i < FMA; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, zf, zc, 0, 0, 0);
#pragma unroll
    for (int i = FMA; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        acc[i][j] = zc;
        asm volatile("" : "+v"(acc[i][j]));
      }
  }

  // Tile boundary and refill schedule.  The boundary of tile t (barrier + waits) sits BND MFMAs into its k-step 3: that k-step's fragments
  // were requested a whole k-step earlier, so by then the lgkmcnt(0) costs nothing.  It retires stage t & 1; tile t + 2 goes there, one
  // piece per DMA_GAP MFMAs counted from the boundary: pieces [0, P1) under the rest of k-step 3, the others under k-steps 0-2 of tile
  // t + 1, all issued at least one k-step before the boundary of tile t + 1 waits for them.
  constexpr int BND = FN, K3S = NMF - BND;
  constexpr int P1 = ((K3S + DMA_GAP - 1) / DMA_GAP) < LOADS ? ((K3S + DMA_GAP - 1) / DMA_GAP) : LOADS;
  static_assert((LOADS - 1) * DMA_GAP < K3S + 2 * NMF + BND, "refills must be issued at least one k-step before the boundary that waits for them");

  if constexpr (ASMLOOP != 0) {
    static_assert(MODE == 0 && !RAGGED && ABL == 0 && (FM == 4 || FM == 5), "the asm K loop exists for the NT form with 256- / 320-row tiles");
    const uint32_t lda_b = (uint32_t)(p.lda * 2), ldb_b = (uint32_t)(p.ldb * 2), nk_u = (uint32_t)nk, tid_u = (uint32_t)tid;
#ifdef UDM_QUADLOOP_TIMELINE   // diagnostic build (make UDM_QUADLOOP=timeline): [block][wave]{loop cycles, cycles in the boundary's vmcnt wait, in its barrier, nk}
    const char* tlp = uniform_ptr(reinterpret_cast<const char*>(p.timeline) + ((size_t)blockIdx.x * 4 + wave) * 16);
#define UDM_QUADLOOP_TL_OPERAND "s"(tlp),
#else
#define UDM_QUADLOOP_TL_OPERAND
#endif
#define UDM_ACC16_ROWS_0_3                                                                                                                                  \
  "+{a[0:15]}"(acc[0][0]), "+{a[16:31]}"(acc[0][1]), "+{a[32:47]}"(acc[0][2]), "+{a[48:63]}"(acc[0][3]), "+{a[64:79]}"(acc[1][0]), "+{a[80:95]}"(acc[1][1]),       \
      "+{a[96:111]}"(acc[1][2]), "+{a[112:127]}"(acc[1][3]), "+{a[128:143]}"(acc[2][0]), "+{a[144:159]}"(acc[2][1]), "+{a[160:175]}"(acc[2][2]),                 \
      "+{a[176:191]}"(acc[2][3]), "+{a[192:207]}"(acc[3][0]), "+{a[208:223]}"(acc[3][1]), "+{a[224:239]}"(acc[3][2]), "+{a[240:255]}"(acc[3][3])
#ifdef UDM_QUADLOOP16_NT5_ASM   // (a build with `make UDM_QUADLOOP=mf16`: measured slower - 21 cycles per 16x16x32 MFMA against the pipe's 16.5 - and not shipped)
    if constexpr (ASMLOOP == 2 && FM == 5) {
      asm volatile(UDM_QUADLOOP16_NT5_ASM
                   : UDM_ACC16_ROWS_0_3, "+{v[192:207]}"(acc[FM - 1][0]), "+{v[208:223]}"(acc[FM - 1][1]), "+{v[224:239]}"(acc[FM - 1][2]), "+{v[240:255]}"(acc[FM - 1][3])
                   : "s"(ap), "s"(bp), "s"(lda_b), "s"(ldb_b), "s"(lds0), "s"(nk_u), UDM_QUADLOOP_TL_OPERAND "v"(tid_u)
                   : UDM_QUADLOOP16_NT5_CLOBBERS);
    } else if constexpr (ASMLOOP == 2) {
      asm volatile(UDM_QUADLOOP16_NT4_ASM
                   : UDM_ACC16_ROWS_0_3
                   : "s"(ap), "s"(bp), "s"(lda_b), "s"(ldb_b), "s"(lds0), "s"(nk_u), UDM_QUADLOOP_TL_OPERAND "v"(tid_u)
                   : UDM_QUADLOOP16_NT4_CLOBBERS);
    } else
#endif
    if constexpr (FM == 5) {
      asm volatile(UDM_QUADLOOP_NT5_ASM
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                     "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]),
                     "+v"(acc[FM - 1][0]), "+v"(acc[FM - 1][1]), "+v"(acc[FM - 1][2]), "+v"(acc[FM - 1][3])
                   : "s"(ap), "s"(bp), "s"(lda_b), "s"(ldb_b), "s"(lds0), "s"(nk_u), UDM_QUADLOOP_TL_OPERAND "v"(tid_u)
                   : UDM_QUADLOOP_NT5_CLOBBERS);
    } else {
      asm volatile(UDM_QUADLOOP_NT4_ASM
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                     "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3])
                   : "s"(ap), "s"(bp), "s"(lda_b), "s"(ldb_b), "s"(lds0), "s"(nk_u), UDM_QUADLOOP_TL_OPERAND "v"(tid_u)
                   : UDM_QUADLOOP_NT4_CLOBBERS);
    }
  } else {
  // ---- prologue: tile 0 and the first part of tile 1 in flight, tile 0 landed, its first fragments in registers ----
#pragma unroll
  for (int j = 0; j < LOADS; ++j) dma_piece(0, 0, j);
  if (nk > 1) {
#pragma unroll
    for (int j = 0; j < P1; ++j) dma_piece(1, 1, j);   // (ahead = 1, stage 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P1) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int n = 0; n < NRD; ++n) read_one(0, 0, 0, n);

  // One K tile.  The issue order is written out and pinned (sched_barrier around every MFMA): an in-order wave can only hide the issue
  // slots of reads and refills in the shadow of an MFMA that is already executing.
  auto tile_body = [&](int t, auto stage_c) {
    // stage_c: 0 / 1 = steady-state tile (the stage is a compile-time constant; a next tile and a refill always exist: no branches);
    // -1 = one of the last tiles (everything derived from t at run time)
    constexpr int STC = decltype(stage_c)::value;
    const int ST = STC >= 0 ? STC : (t & 1);
    const bool HAS_NEXT = STC >= 0 || t + 1 < nk, HAS_DMA = STC >= 0 || t + 2 < nk;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int buf = kk & 1;
      int rd = 0;
#pragma unroll
      for (int m = 0; m < NMF; ++m) {
        if (kk == 3 && m == BND) {
          // boundary: every fragment of this stage is in registers (lgkmcnt) and this wave's pieces of the next tile have landed (vmcnt);
          // after the barrier every wave's have, and the stage just read may be refilled
          __builtin_amdgcn_sched_barrier(0);
          if (ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const int i = m / FN, j = m % FN;
        if (i < FMA) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_a(buf, i), frag_b(buf, j), acc[i][j], 0, 0, 0);
        } else {
          const bf16x8_t av = frag_a(buf, i), bv = frag_b(buf, j);
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(av), "v"(bv));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 3) {   // first part of tile t + 2 into the stage this boundary retired; first fragments of tile t + 1
          if (m >= BND) {
            const int slot = m - BND;
            if (HAS_DMA && slot % DMA_GAP == 0 && slot / DMA_GAP < P1) dma_piece(2, ST, slot / DMA_GAP);
            const int upto = ((slot + 1) * NRD) / K3S;
#pragma unroll
            for (; rd < upto; ++rd)
              if (HAS_NEXT) read_one(ST ^ 1, 0, 0, rd);
          }
        } else {         // rest of tile t + 1 into the stage the previous boundary retired; fragments of the next k-step
          const int slot = K3S + kk * NMF + m;
          if (HAS_NEXT && slot % DMA_GAP == 0 && slot / DMA_GAP >= P1 && slot / DMA_GAP < LOADS) dma_piece(1, ST ^ 1, slot / DMA_GAP);
          const int upto = ((m + 1) * NRD) / NMF;
#pragma unroll
          for (; rd < upto; ++rd) read_one(ST, kk + 1, buf ^ 1, rd);
        }
      }
    }
    ap += kstep_a;
    bp += kstep_b;
    __builtin_amdgcn_sched_barrier(0);
  };
  int t = 0;
  for (; t + 3 < nk; t += 2) {
    tile_body(t, std::integral_constant<int, 0>{});
    tile_body(t + 1, std::integral_constant<int, 1>{});
  }
  for (; t < nk; ++t) tile_body(t, std::integral_constant<int, -1>{});
  }
  if (FM > FMA) asm volatile("s_nop 15" ::: "memory");   // inline-asm MFMA results -> first reader (wait states the compiler does not know it owes)
  __syncthreads();  // all LDS tile reads are done: the wave-private epilogue patches may overwrite stage memory

  // ---- epilogue: each 32 x 32 accumulator block goes through a wave-private 4 KiB LDS patch so a lane owns 4 consecutive columns of a row
  float* patch0 = reinterpret_cast<float*>(smem) + wave * 2048;
  const int er = lane >> 3, ec = (lane & 7) * 4;
  float bias4[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias4[j][e] = (EPI == UDM_EPI_BIAS || EPI == UDM_EPI_BIAS_GELU) ? p.bias[col0 + wn * 128 + j * 32 + ec + e] : 0.f;
  float* colsum = (EPI == UDM_EPI_DGELU) ? const_cast<float*>(p.bias) : nullptr;
  float csum[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) csum[j][e] = 0.f;
#pragma clang loop unroll(full)
  for (int i = 0; i < FM; ++i)
#pragma clang loop unroll(full)
    for (int j = 0; j < FN; ++j) {
      float* patch = patch0 + ((i * FN + j) & 1) * 1024;
#pragma clang loop unroll(full)
      for (int r = 0; r < 16; ++r) {
        if constexpr (ASMLOOP == 2)   // sub-block (r >> 3, (r >> 2) & 1), register r & 3 of a 16 x 16 tile: row 4 (lane >> 4) + (r & 3), column lane & 15
          patch[((r >> 3) * 16 + 4 * (lane >> 4) + (r & 3)) * 32 + ((r >> 2) & 1) * 16 + (lane & 15)] = acc[i][j][r];
        else patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[i][j][r];
      }
      const int gn = col0 + wn * 128 + j * 32 + ec;
      const int gm0 = row0 + wm * 32 * FM + i * 32 + er;
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(patch + (q * 8 + er) * 32 + ec);
      uint2 au[4];
      float4 cold[4];
      if (EPI == UDM_EPI_DGELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) au[q] = *reinterpret_cast<const uint2*>(p.aux + (long)(RAGGED ? min(gm0 + q * 8, p.M - 1) : gm0 + q * 8) * p.ldaux + gn);
      }
      if (OUT_F32 && p.beta != 0.f && S == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cold[q] = *reinterpret_cast<const float4*>(reinterpret_cast<float*>(p.C) + (long)(RAGGED ? min(gm0 + q * 8, p.M - 1) : gm0 + q * 8) * p.ldc + gn);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float x[4] = {v[q].x + bias4[j][0], v[q].y + bias4[j][1], v[q].z + bias4[j][2], v[q].w + bias4[j][3]};
        const long gm = gm0 + q * 8;
        if (RAGGED && gm >= p.M) continue;
        if (EPI == UDM_EPI_BIAS_GELU) {
          bf16_t pre[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { float dg; gelu_tanh_both(bf2f(f2bf(x[e])), x[e], dg); pre[e] = f2bf(dg); }
          *reinterpret_cast<uint2*>(p.aux + gm * p.ldaux + gn) = make_uint2((uint32_t)pre[0] | ((uint32_t)pre[1] << 16), (uint32_t)pre[2] | ((uint32_t)pre[3] << 16));
        }
        if (EPI == UDM_EPI_DGELU) {
          x[0] *= __uint_as_float(au[q].x << 16); x[1] *= __uint_as_float(au[q].x & 0xffff0000u);
          x[2] *= __uint_as_float(au[q].y << 16); x[3] *= __uint_as_float(au[q].y & 0xffff0000u);
          if (colsum) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[j][e] += OUT_F32 ? x[e] : bf2f(f2bf(x[e]));
          }
        }
        if (OUT_F32 && S > 1) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + slice * p.slice_stride + gm * p.ldc + gn) = make_float4(x[0], x[1], x[2], x[3]);
        } else if (OUT_F32) {
          if (p.beta != 0.f) { x[0] += p.beta * cold[q].x; x[1] += p.beta * cold[q].y; x[2] += p.beta * cold[q].z; x[3] += p.beta * cold[q].w; }
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + gm * p.ldc + gn) = make_float4(x[0], x[1], x[2], x[3]);
        } else {
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + gm * p.ldc + gn) = make_uint2(pack2bf(x[0], x[1]), pack2bf(x[2], x[3]));
        }
      }
    }
  if (EPI == UDM_EPI_DGELU && colsum) {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = csum[j][e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (er == 0) atomicAdd(colsum + col0 + wn * 128 + j * 32 + ec + e, v);
      }
  }
}

template <int FM, int MODE, int EPI, bool OUT_F32, bool RAGGED = false>
int launch_quad_t(const QuadArgs& a0, hipStream_t stream) {
  QuadArgs a = a0;
  constexpr int BM = 64 * FM;
  a.tiles_m = RAGGED ? (a.M + BM - 1) / BM : a.M / BM;
  a.tiles_n = a.N / 256;
  static const int env_gm = [] { const char* e = getenv("UDM_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();
  a.group_m = env_gm;
  const size_t lds = (size_t)2 * (BM + 256) * BK * 2;
  constexpr int GAP = 2;
  auto kern = gemm_quad_kernel<FM, MODE, EPI, OUT_F32, GAP, 0, RAGGED>;
  if constexpr (MODE == 0 && !RAGGED && (FM == 4 || FM == 5)) {
    static const int env_asm = [] { const char* e = getenv("UDM_QUAD_ASM"); return e ? atoi(e) : 1; }();
    const int nk = a.K / BK;
    const int mode = g_quad_asm < 0 ? env_asm : g_quad_asm;      // 1 = the 32x32x16 statement, 2 = the 16x16x32 statement
    if (mode && a.splitk <= 1 && nk >= 4 && nk % 2 == 0) {
      kern = gemm_quad_kernel<FM, MODE, EPI, OUT_F32, GAP, 0, RAGGED, 1>;
#ifdef UDM_QUADLOOP16_NT5_ASM
      if (mode == 2) kern = gemm_quad_kernel<FM, MODE, EPI, OUT_F32, GAP, 0, RAGGED, 2>;
#endif
    }
  }
  if constexpr (EPI == UDM_EPI_NONE && FM >= 4 && !RAGGED) {   // timing-only ablations of the plain kernels (scripts/bench_gemm_quad.py)
    static const int abl = [] { const char* e = getenv("UDM_QUAD_ABL"); return e ? atoi(e) : 0; }();
    if (abl == 1) kern = gemm_quad_kernel<FM, MODE, EPI, OUT_F32, GAP, 1>;
    if (abl == 2) kern = gemm_quad_kernel<FM, MODE, EPI, OUT_F32, GAP, 2>;
  }
  a.timeline = g_quad_timeline;
  static const void* attr_set = nullptr;
  if (attr_set != (const void*)kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = (const void*)kern;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n * (a.splitk > 1 ? a.splitk : 1)), dim3(256), lds, stream, a);
  UDM_CHECK_LAUNCH("udm_gemm (quad)");
  return 0;
}
}  // namespace

unsigned* g_quad_timeline = nullptr;   // diagnostic builds only (udm_debug_set "gemm_quad_timeline")
int g_quad_asm = -1;    // A/B switch (udm_debug_set "gemm_quad_asm"): -1 = env UDM_QUAD_ASM (default on), 0 = the C++ K loop everywhere
int g_quad_mode = -1;   // -1: read UDM_GEMM_QUAD (default 1); 0 = off, 1 = where it fills the chip, 2 = wherever the shape fits (tests)
int udm_quad_mode() {
  if (g_quad_mode < 0) {
    const char* e = getenv("UDM_GEMM_QUAD");
    g_quad_mode = e ? atoi(e) : 1;
  }
  return g_quad_mode;
}

bool udm_quad_tn_ok(long M, long N, long K, int* fm) {
  if (!udm_quad_mode() || K % 64 != 0 || K < 128 || N % 256 != 0) return false;
  // the tile height that fills whole rounds of the CUs (256, or what udm_gemm_set_cus leaves) best (ties: the larger wave tile)
  const long G = udm_gemm_cus_available();
  const int cands[2] = {4, 3};   // (FM = 5 spills in the TN form: 2 x 9 fragment halves beside 320 accumulators)
  double best = 1e30;
  int pick = 0;
  for (int c = 0; c < 2; ++c) {
    const int bm = 64 * cands[c];
    if (M % bm != 0) continue;
    const long tiles = (M / bm) * (N / 256);
    const long rounds = (tiles + G - 1) / G;
    const double t = (double)rounds * bm;
    if (t < best) { best = t; pick = cands[c]; }
  }
  if (!pick) return false;
  *fm = pick;
  return true;
}

int udm_quad_launch_tn_pair(const QuadArgs& a0, long M2, hipStream_t stream) {
  QuadArgs a = a0;
  a.tiles_m_split = a.M / 256;
  a.M = a.M + (int)M2;        // launch_quad_t derives the tile rows of BOTH problems from M
  return launch_quad_t<4, true, UDM_EPI_NONE, true>(a, stream);
}

int udm_quad_launch_tn_multi(const QuadArgs& a0, hipStream_t stream) {
  QuadArgs a = a0;
  a.M = a.m_tile_start[a.nprob] * 256;   // launch_quad_t derives the grid from M / 256 x N / 256: all tiles of all problems
  a.N = 256;
  a.tiles_m_split = 0;
  return launch_quad_t<4, true, UDM_EPI_NONE, true>(a, stream);
}

int udm_quad_launch_tn(const QuadArgs& a, int fm, hipStream_t stream) {
  switch (fm) {
    case 3: return launch_quad_t<3, true, UDM_EPI_NONE, true>(a, stream);
    case 4: return launch_quad_t<4, true, UDM_EPI_NONE, true>(a, stream);
    default: udm_set_error("udm_quad_launch_tn: bad tile"); return 2;
  }
}

bool udm_quad_nt_ok(long M, long N, long K, int* fm) {
  if (!udm_quad_mode() || K % 64 != 0 || K < 128 || N % 256 != 0) return false;
  const long G = udm_gemm_cus_available();
  const int cands[3] = {5, 4, 3};
  double best = 1e30;
  int pick = 0;
  for (int c = 0; c < 3; ++c) {
    const int bm = 64 * cands[c];
    if (M % bm != 0) continue;
    const long tiles = (M / bm) * (N / 256);
    const long rounds = (tiles + G - 1) / G;
    const double t = (double)rounds * bm;
    if (t < best) { best = t; pick = cands[c]; }
  }
  if (!pick) return false;
  *fm = pick;
  return true;
}

template <int FM>
static int launch_quad_nt_fm(const QuadArgs& a, int epilogue, int out_f32, hipStream_t stream) {
  if (out_f32) {
    if (epilogue != UDM_EPI_NONE) { udm_set_error("udm_quad_launch_nt: fp32 output only without an epilogue"); return 2; }
    return launch_quad_t<FM, false, UDM_EPI_NONE, true>(a, stream);
  }
  switch (epilogue) {
    case UDM_EPI_NONE: return launch_quad_t<FM, false, UDM_EPI_NONE, false>(a, stream);
    case UDM_EPI_BIAS: return launch_quad_t<FM, false, UDM_EPI_BIAS, false>(a, stream);
    case UDM_EPI_BIAS_GELU: return launch_quad_t<FM, false, UDM_EPI_BIAS_GELU, false>(a, stream);
    default: return launch_quad_t<FM, false, UDM_EPI_DGELU, false>(a, stream);
  }
}

int udm_quad_launch_nt(const QuadArgs& a, int fm, int epilogue, int out_f32, hipStream_t stream) {
  if (fm == -5) {   // 320-row tiles, ragged last tile row: plain / bias epilogue, bf16 output
    if (out_f32 || epilogue > UDM_EPI_BIAS) { udm_set_error("udm_quad_launch_nt: the ragged form has the plain and the bias epilogue only"); return 2; }
    return epilogue == UDM_EPI_BIAS ? launch_quad_t<5, false, UDM_EPI_BIAS, false, true>(a, stream) : launch_quad_t<5, false, UDM_EPI_NONE, false, true>(a, stream);
  }
  switch (fm) {
    case 3: return launch_quad_nt_fm<3>(a, epilogue, out_f32, stream);
    case 4: return launch_quad_nt_fm<4>(a, epilogue, out_f32, stream);
    case 5: return launch_quad_nt_fm<5>(a, epilogue, out_f32, stream);
    default: udm_set_error("udm_quad_launch_nt: bad tile"); return 2;
  }
}

// NN (dgrad) form: C[M, N] bf16 = A[M, K] B[K, N]; whole tiles, plain epilogue
// fm = 3 / 4 / 5: whole tiles of that height; fm = -5: 320-row tiles with a ragged last tile row, where that is fewer rounds x rows than any whole-tile height
bool udm_quad_nn_ok(long M, long N, long K, int* fm) {
  int whole = 0;
  const bool ok = udm_quad_nt_ok(M, N, K, &whole);
  const long G = udm_gemm_cus_available();
  double best = ok ? (double)(((M / (64 * whole)) * (N / 256) + G - 1) / G) * 64 * whole : 1e30;
  static const int env_ragged = [] { const char* e = getenv("UDM_QUAD_RAGGED"); return e ? atoi(e) : 1; }();   // diagnostics: 0 = whole tiles only
  if (env_ragged && udm_quad_mode() && K % 64 == 0 && K >= 128 && N % 256 == 0 && M > 320 && M % 320 != 0) {
    const long tiles = ((M + 319) / 320) * (N / 256);
    const double t = (double)((tiles + G - 1) / G) * 320 * 1.03;
    if (tiles >= 128 && t < best) { *fm = -5; return true; }
  }
  if (ok) *fm = whole;
  return ok;
}
int udm_quad_launch_nn_f32(const QuadArgs& a, int fm, hipStream_t stream) {   // fp32 output (split-K partial tiles)
  switch (fm) {
    case 3: return launch_quad_t<3, 2, UDM_EPI_NONE, true>(a, stream);
    case 4: return launch_quad_t<4, 2, UDM_EPI_NONE, true>(a, stream);
    case 5: return launch_quad_t<5, 2, UDM_EPI_NONE, true>(a, stream);
    default: udm_set_error("udm_quad_launch_nn_f32: bad tile"); return 2;
  }
}
int udm_quad_launch_nn(const QuadArgs& a, int fm, hipStream_t stream) {
  switch (fm) {
    case -5: return launch_quad_t<5, 2, UDM_EPI_NONE, false, true>(a, stream);
    case 3: return launch_quad_t<3, 2, UDM_EPI_NONE, false>(a, stream);
    case 4: return launch_quad_t<4, 2, UDM_EPI_NONE, false>(a, stream);
    case 5: return launch_quad_t<5, 2, UDM_EPI_NONE, false>(a, stream);
    default: udm_set_error("udm_quad_launch_nn: bad tile"); return 2;
  }
}
