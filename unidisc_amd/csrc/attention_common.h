// Shared device helpers of the attention kernels (attention.hip, attention_dkv_ws.hip): XOR-swizzled LDS tile layout, MFMA
// fragment gathers, LDS-DMA staging, the kernel argument block.  Everything has internal linkage (included per translation unit).
#pragma once
#include "common.h"
#include "../../include/unidisc_hip.h"

namespace {
using namespace udm;

template <int D>
__device__ __forceinline__ int swz(int row) {
  if (D == 128) return (row & 15) ^ ((row & 3) << 2);
  if (D == 64) { int x = row >> 1; return (x & 7) ^ ((x & 1) << 2); }
  return (row >> 2) & 3;  // D == 32
}
// byte offset of 16-byte slot `slot` of row `row` in a [rows][D] bf16 tile
template <int D>
__device__ __forceinline__ int tile_off(int row, int slot) { return row * (2 * D) + ((slot ^ swz<D>(row)) << 4); }

__device__ __forceinline__ bf16x8_t lds_frag(const char* base, int off) { return *reinterpret_cast<const bf16x8_t*>(base + off); }

// A-operand fragment of X^T for the permuted 16-row chunk starting at row r0 (rows = contraction index),
// 32 columns starting at c0:  element j of lane (col = lane&31, half h = lane>>5) is
//   X[r0 + 4h + j][c0 + col] (j<4),  X[r0 + 8 + 4h + (j-4)][c0 + col] (j>=4).
template <int D, bool USE_TR>
__device__ __forceinline__ bf16x8_t lds_frag_T(const char* tile, int r0, int c0, int lane) {
  const int hi = lane >> 5;
  if (USE_TR) {
    const int g1 = (lane >> 4) & 1, p = lane & 15;
    const int row = r0 + 4 * hi + (p >> 2);
    const int col = c0 + g1 * 16 + (p & 3) * 4;
    const int o1 = tile_off<D>(row, col >> 3) + (col & 7) * 2;
    const int o2 = tile_off<D>(row + 8, col >> 3) + (col & 7) * 2;
    s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(tile + o1));
    s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(tile + o2));
    s16x8_t r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
  } else {
    const int col = c0 + (lane & 31);
    s16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = r0 + (j < 4 ? 4 * hi + j : 8 + 4 * hi + (j - 4));
      r[j] = *reinterpret_cast<const short*>(tile + tile_off<D>(row, col >> 3) + (col & 7) * 2);
    }
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

// typed LDS load from a 32-bit LDS byte address (hoisted per-lane base + immediate offsets)
template <typename T>
__device__ __forceinline__ T lds_ld(uint32_t addr) { return *reinterpret_cast<UDM_LDS const T*>((size_t)addr); }

__device__ __forceinline__ bf16x8_t pack8(const float* p) {
  uint4 u = make_uint4(pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7]));
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ bf16x8_t load_frag_global(const bf16_t* p, bool ok) {
  uint4 u = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
  return __builtin_bit_cast(bf16x8_t, u);
}

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* dout;
  bf16_t* out;          // fwd: O;            bwd-dq: dQ
  bf16_t* out2;         // bwd-dkv: dK
  bf16_t* out3;         // bwd-dkv: dV
  float* lse;           // [B,H,L] log2-domain log-sum-exp of scaled scores
  const float* delta;   // [B,H,L] rowsum(dO * O)
  const int64_t* sample_ids;  // [B,L] or null
  const int* doc_ranges;      // [B, ceil(L/64), 8] or null: per 64-row tile {lo, hi, idmin, idmax, exact, -, -, -} (udm_attention_doc_ranges)
  int doc_pure_split;         // dK/dV with documents: 1 = document-pure key blocks are computed by the wave-specialised kernel, the rest by the other
  long q_stride, k_stride, v_stride, o_stride, do_stride, out_stride, out2_stride, out3_stride;
  int B, H, L;
  float scale_log2;     // log2(e) / sqrt(D)
  float scale;          // 1 / sqrt(D)
  unsigned long long* timeline;   // diagnostics (experiments library): cycle stamps of a few blocks, or null
  int exp;                        // experiment bits (udm_exp_flags), 0 in production
  int q_prescaled;                // UDM_ATTN_Q_PRESCALED: q carries log2(e) / sqrt(D) (scale_log2 = 1; the backward's `scale` = ln 2)
};

// Attention mask codes.  `sample_ids` holds one int64 per position:  bits 0-31 = sample id (signed; < 0 = padding), bits 32-39 = the KEY classes this
// position belongs to, bits 40-47 = the key classes this position may see as a QUERY.  A query sees a key iff both sample ids are equal and >= 0 and
// (query mask & key class) != 0.  Raw sample ids (document masks of packed batches) have zero high bits, which reads as "every class": the plain
// document mask.  Modality attention dropout (model.py:863-878, model_utils.py:721-731: text queries see text keys only / image queries see image
// keys only, per sample) sets key class 1 = text, 2 = image and the query mask accordingly - an asymmetric mask the id equality alone cannot express.
__device__ __forceinline__ bool attn_pair_ok(long code_q, long code_k) {
  const int iq = (int)code_q, ik = (int)code_k;
  unsigned qm = (unsigned)(code_q >> 40) & 0xffu, kb = (unsigned)(code_k >> 32) & 0xffu;
  qm = qm ? qm : 0xffu;
  kb = kb ? kb : 0xffu;
  return iq == ik && iq >= 0 && (qm & kb) != 0;
}

// 1-D grid -> ((b, h), tile).  Blocks go to XCD (block id % 8) in dispatch order.  All tiles of one (b, h) run on ONE XCD and share its K / V (or
// Q / dO) through that L2 - and an XCD works through its (b, h) pairs ONE AFTER THE OTHER (tiles of a pair adjacent in dispatch order): with the
// tile-major order used before, the ~64 blocks resident on an XCD spanned 16 pairs (10 MB of K / V against a 4 MB L2; PMC: the forward fetched
// 414 MB over the fabric for 126 MB of operands).  Needs B*H % 8 == 0; otherwise the plain tile-major order.
__device__ __forceinline__ void attn_block_to_work(int bid, int nbh, int& bh, int& tile) {
  const int ntiles = gridDim.x / nbh;
  if ((nbh & 7) == 0) {
    const int x = bid & 7, j = bid >> 3;
    bh = (j / ntiles) * 8 + x;
    tile = j % ntiles;
  } else {
    bh = bid % nbh;
    tile = bid / nbh;
  }
}

constexpr int DOC_STRIDE = 8;   // ints per tile in doc_ranges: {lo, hi, idmin, idmax, exact, 0, 0, 0}
// Document masks (packed samples): a 128-row block only has to walk the 64-row tiles of the other side that can hold one of its sample ids.
// `doc_ranges` (udm_attention_doc_ranges) gives, per 64-row tile, {lo, hi, idmin, idmax, exact}: the [lo, hi) span of positions whose id lies inside the
// tile's [idmin, idmax] of valid ids (lo = hi = 0 for a tile of padding only), and that id interval itself - with idmin = -1 when the tile holds
// any padding, so that idmin == idmax >= 0 means "every row of this tile belongs to the one document idmin".  The span is conservative for any id
// layout and exact for contiguous documents.  A tile pair that is uniform on both sides with the same id needs no per-element id test; every other
// pair keeps it, so neither skipping tiles nor skipping the test changes a result.
struct DocSpan {
  int t_begin, t_end;   // tiles of the other side to walk
  int blk_id;           // the one document all rows of this 128-row block belong to, or -1
  int lo, hi;           // the same span in positions
  bool pure;            // blk_id >= 0 and every position of [lo, hi) belongs to that document: no per-element test anywhere in the block's walk
};
__device__ __forceinline__ DocSpan doc_tile_span(const int* doc_ranges, int b, int L, int blk128, int ntiles) {
  DocSpan d{0, ntiles, -1, 0, L, false};
  if (doc_ranges == nullptr) return d;
  const int nT = (L + 63) / 64;
  int lo = L, hi = 0, id = -2, exact = 1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = blk128 * 2 + j;
    if (t < nT) {
      const int4 r = *reinterpret_cast<const int4*>(doc_ranges + ((long)b * nT + t) * DOC_STRIDE);
      if (r.y > r.x) { lo = min(lo, r.x); hi = max(hi, r.y); }
      const int tid_ = (r.z == r.w) ? r.z : -1;
      id = (id == -2 || id == tid_) ? tid_ : -1;
      exact &= doc_ranges[((long)b * nT + t) * DOC_STRIDE + 4];
    }
  }
  d.blk_id = __builtin_amdgcn_readfirstlane(id < 0 ? -1 : id);
  if (hi <= lo) { d.t_begin = 0; d.t_end = 0; d.lo = 0; d.hi = 0; return d; }
  d.lo = __builtin_amdgcn_readfirstlane(lo);
  d.hi = __builtin_amdgcn_readfirstlane(hi);
  d.t_begin = d.lo / 64;
  d.t_end = (d.hi + 63) / 64;
  d.pure = d.blk_id >= 0 && __builtin_amdgcn_readfirstlane(exact) != 0;
  return d;
}
// does the pair (this block, tile t of the other side) need the per-element id test?
__device__ __forceinline__ bool doc_pair_needs_mask(const int* doc_ranges, int b, int L, int t, int blk_id) {
  if (doc_ranges == nullptr || blk_id < 0) return true;
  const int nT = (L + 63) / 64;
  const int2 r = *reinterpret_cast<const int2*>(doc_ranges + ((long)b * nT + t) * DOC_STRIDE + 2);
  return !(r.x == blk_id && r.y == blk_id);
}

// stage a [ROWS][D] bf16 tile with LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  One wave
// instruction moves 1 KiB = 1024/(2D) rows; lane i lands at +16*i, so the XOR swizzle is applied on the per-lane SOURCE
// slot (the swizzle is an involution).  Rows past L are clamped to row L-1: finite data that every consumer masks out.
template <int D, int ROWS>
struct DmaStager {
  static constexpr int LPR = D / 8;               // lanes (16-byte slots) per row
  static constexpr int RPI = 64 / LPR;            // rows per wave instruction
  static constexpr int NINSTR = ROWS / RPI;       // instructions per tile
  static constexpr int PW = NINSTR / 4;           // per wave (4 waves per block)
  static_assert(NINSTR % 4 == 0 && PW >= 1, "tile must split evenly over 4 waves");
  static __device__ __forceinline__ void issue(const bf16_t* base, long stride, int row0, int L, char* tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int idx = wave * PW + j;
      const int row = idx * RPI + lane / LPR;
      const int slot = (lane % LPR) ^ swz<D>(row);
      const int grow = min(row0 + row, L - 1);
      // Issued as inline asm ON PURPOSE: the compiler tracks LDS-DMA it knows about and puts s_waitcnt vmcnt(0) in front of every LDS read
      // it cannot prove disjoint - which includes all ds_read_b64_tr_b16 - so the refill of the OTHER stage was waited for in the middle
      // of the tile it should overlap with.  The waits that matter are explicit (wait_all_vmem before the barrier that publishes a stage).
      const uint32_t dst = (uint32_t)(size_t)(UDM_LDS char*)(tile + idx * 1024);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(base + (long)grow * stride + slot * 8), "s"(dst) : "memory", "m0");
    }
  }
};
// Full tiles (every row < L) need no per-piece vector arithmetic: the per-lane byte offset of piece j inside a tile is fixed
// (row-in-tile * stride + swizzled slot), the tile's first row goes into a wave-uniform 64-bit base (SALU), and the DMA takes the
// (SGPR base + 32-bit VGPR offset) form.  The clamped general form above cost ~10 VALU instructions per piece, three of them quarter-rate
// integer multiplies - about 600 cycles per 64-key tile and wave in the forward / dQ loops.
template <int D, int ROWS>
struct DmaPlan {
  using Stg = DmaStager<D, ROWS>;
  uint32_t off[Stg::PW];
  __device__ __forceinline__ void init(long stride, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < Stg::PW; ++j) {
      const int idx = wave * Stg::PW + j;
      const int row = idx * Stg::RPI + lane / Stg::LPR;
      const int slot = (lane % Stg::LPR) ^ swz<D>(row);
      off[j] = (uint32_t)((row * stride + slot * 8) * 2);
    }
  }
  // tile_base: address of the tile's first row (wave-uniform); tile: LDS stage
  __device__ __forceinline__ void issue_full(const bf16_t* tile_base, char* tile, int wave) const {
    const uint64_t u = reinterpret_cast<uint64_t>(tile_base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    const uint64_t sb = ((uint64_t)hi32 << 32) | lo;
#pragma unroll
    for (int j = 0; j < Stg::PW; ++j) {
      const uint32_t dst = (uint32_t)(size_t)(UDM_LDS char*)(tile + (wave * Stg::PW + j) * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off[j]), "s"(sb), "s"(dst) : "memory", "m0");
    }
  }
};
// single LDS-DMA pieces as inline asm (same reason as in DmaStager): 16 or 4 bytes per lane, lane i lands at lds + 16 i / lds + 4 i
__device__ __forceinline__ void dma16_asm(const void* gptr, const char* lds) {
  const uint32_t dst = (uint32_t)(size_t)(UDM_LDS const char*)lds;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ void dma4_asm(const void* gptr, const char* lds) {
  const uint32_t dst = (uint32_t)(size_t)(UDM_LDS const char*)lds;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ void wait_all_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Epilogue of the transposed-accumulator kernels at head dim 128: a wave holds X^T for 32 rows (lane & 31 = row, registers walk the 128
// columns: acc[i][r] = column i*32 + 8*(r>>2) + 4*hi + (r&3)).  Per-lane stores of that layout are 8-byte pieces at a row stride (32 - 64 cache
// lines touched per store instruction; the store tail is issue-bound).  Instead: scale, pack to bf16, stage through 8 KiB of LDS owned by the wave
// (16-byte slots XOR-swizzled with the row: conflict-free for the 8-byte writes and the 16-byte reads) and store whole 256-byte rows, four rows
// per instruction.  `rows_valid` rows are stored (ragged last block).  The caller makes sure nobody still reads that LDS (barrier).
// (rows 16..31 are staged `half_stride` bytes behind rows 0..15: 4096 = one contiguous 8 KiB region)
__device__ __forceinline__ void store_rows_via_lds_d128(char* wave_lds, const f32x16_t (&acc)[4], float scale, bf16_t* out_row0, long stride, int rows_valid, int lane,
                                                        int half_stride = 4096) {
  const int row = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int slot = i * 4 + rg;
      *reinterpret_cast<uint2*>(wave_lds + (row >> 4) * half_stride + (row & 15) * 256 + ((slot ^ (row & 15)) << 4) + hi * 8) =
          make_uint2(pack2bf(acc[i][rg * 4] * scale, acc[i][rg * 4 + 1] * scale), pack2bf(acc[i][rg * 4 + 2] * scale, acc[i][rg * 4 + 3] * scale));
    }
  // (the wave reads back only what it wrote: the compiler orders this wave's LDS writes before its reads, no barrier needed)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = p * 4 + (lane >> 4), slot = lane & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(wave_lds + (r >> 4) * half_stride + (r & 15) * 256 + ((slot ^ (r & 15)) << 4));
    if (r < rows_valid) *reinterpret_cast<uint4*>(out_row0 + (long)r * stride + slot * 8) = v;
  }
}
}  // namespace

// dK/dV kernel for head dim 128 without a document mask (attention_dkv_ws.hip); grid = ceil(L / 128) * B * H blocks of 512 threads
void udm_launch_attn_bwd_dkv_ws(const void* args, hipStream_t stream);
bool udm_launch_attn_bwd_dq64(const void* args, hipStream_t stream);    // attention_dq64.hip: dQ (+ delta and the planes) at D = 128, no mask, L % 256 == 0, q pre-scaled (false = shape not taken)
bool udm_launch_attn_bwd_dkv64(const void* args, hipStream_t stream);   // attention_dkv64.hip: dK / dV at D = 128, no mask, L % 256 == 0, q pre-scaled (false = shape not taken)
bool udm_launch_attn_fwd64(const void* args, hipStream_t stream);      // attention_fwd64.hip: forward at D = 128, no mask, L % 256 == 0 (false = shape not taken)
