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
  long q_stride, k_stride, v_stride, o_stride, do_stride, out_stride, out2_stride, out3_stride;
  int B, H, L;
  float scale_log2;     // log2(e) / sqrt(D)
  float scale;          // 1 / sqrt(D)
};

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
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long)grow * stride + slot * 8),
                                       (UDM_LDS void*)(tile + idx * 1024), 16, 0, 0);
    }
  }
};
__device__ __forceinline__ void wait_all_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}  // namespace

// dK/dV kernel for head dim 128 without a document mask (attention_dkv_ws.hip); grid = ceil(L / 128) * B * H blocks of 512 threads
void udm_launch_attn_bwd_dkv_ws(const void* args, hipStream_t stream);
