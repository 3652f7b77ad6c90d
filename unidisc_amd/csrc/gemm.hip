// bf16 MFMA GEMM for gfx950:  C[M,N] (op)= A[M,K] · B[N,K]^T   ("NT": both operands K-contiguous)
//
// Replaces every nn.Linear on the hot path (reference models/dit.py:562,567,917-919,1068 — cuBLAS bf16
// GEMMs under autocast), the dgrad GEMMs (with a transposed bf16 weight shadow) and the wgrad GEMMs
// (with transposed activations, see transpose_bf16 below).
//
// Structure (round-1): 128x128x64 block tile, 256 threads = 4 waves in 2x2, each wave 64x64 as 2x2
// v_mfma_f32_32x32x16_bf16 fragments; register-staged global->LDS copy with the next tile's loads
// issued before the current tile's MFMAs (one barrier per K tile, two LDS buffers); XOR-swizzled LDS
// rows so every ds_read_b128 fragment read is bank-conflict free; fp32 accumulator tile staged
// through LDS so the epilogue (bias / GELU / GELU' / accumulate) runs on coalesced 16-byte rows.
// Tile order is XCD-aware (8 private L2s) with a grouped-M sweep.
#include "common.h"
#include <algorithm>
#include "gemm_quad.h"
#include "../../include/unidisc_hip.h"

namespace {
using namespace udm;

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int GROUP_M = 8;

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  bf16_t* aux;
  long lda, ldb, ldc, ldaux;
  int M, N, K;
  int tiles_m, tiles_n;
  float beta;
  int splitk;  // > 1: the K range is cut into `splitk` slices per tile and partial tiles are atomically added into an fp32 C
  long slice_stride;  // split-K with a workspace: slice s stores its partial tile (no atomics) at C + s * slice_stride; 0 = atomic form
  int group_m;        // big-tile kernels: row tiles per group of the tile order (0 = GROUP_M); diagnostics knob UDM_GEMM_GROUP_M
  int exp;            // experiment bits (udm_exp_flags), 0 in production
};

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A tile + B tile) = 64 KiB

  // ---- tile mapping: XCD-contiguous chunks, grouped along M for L2 reuse of the B panel ----
  const int nwg = p.tiles_m * p.tiles_n;
  int pid = xcd_remap(blockIdx.x, nwg);
  const int per_group = GROUP_M * p.tiles_n;
  const int group = pid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz;
  const int tn = (pid % per_group) / gsz;
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  // ---- global->register staging: thread owns rows ld_row + 32*i, 16-byte slot ld_slot ----
  const int ld_row = tid >> 3, ld_slot = tid & 7;
  const bf16_t* a_ptr[4];
  const bf16_t* b_ptr[4];
  bool a_ok[4], b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ra = row0 + ld_row + 32 * i, rb = col0 + ld_row + 32 * i;
    a_ok[i] = ra < p.M;
    b_ok[i] = rb < p.N;
    a_ptr[i] = p.A + (long)(a_ok[i] ? ra : 0) * p.lda + ld_slot * 8;
    b_ptr[i] = p.B + (long)(b_ok[i] ? rb : 0) * p.ldb + ld_slot * 8;
  }
  const int st_off = ld_row * 128 + ((ld_slot ^ ((ld_row >> 1) & 7)) << 4);  // + i*32*128

  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k = kt * BK;
    const bool kok = (k + ld_slot * 8) < p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = (a_ok[i] && kok) ? *reinterpret_cast<const uint4*>(a_ptr[i] + k) : make_uint4(0, 0, 0, 0);
      rb[i] = (b_ok[i] && kok) ? *reinterpret_cast<const uint4*>(b_ptr[i] + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    char* As = smem + buf * 2 * TILE_BYTES;
    char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(As + st_off + i * 32 * 128) = ra[i];
      *reinterpret_cast<uint4*>(Bs + st_off + i * 32 * 128) = rb[i];
    }
  };

  // ---- fragment read offsets (row = lane&31 within a 32-row fragment, k-slot = 2*kk + lane>>5) ----
  const int sw = (l31 >> 1) & 7;
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_off[i] = (wm * 64 + i * 32 + l31) * 128;
    b_off[i] = (wn * 64 + i * 32 + l31) * 128;
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);  // in flight under the MFMAs below
    const char* As = smem + cur * 2 * TILE_BYTES;
    const char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int so = ((kk * 2 + hi) ^ sw) << 4;
      bf16x8_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const bf16x8_t*>(As + a_off[i] + so);
        b[i] = *reinterpret_cast<const bf16x8_t*>(Bs + b_off[i] + so);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: accumulators -> LDS (fp32 [128][128]) -> coalesced rows ----
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        int n = wn * 64 + j * 32 + l31;
        Cs[m * BN + n] = acc[i][j][r];
      }
  __syncthreads();

  const bool n_vec_ok = (p.ldc % 4 == 0);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int idx = tid + it * NTHREADS;
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    const int gm = row0 + r, gn = col0 + c4;
    if (gm >= p.M || gn >= p.N) continue;
    float4 v = *reinterpret_cast<const float4*>(Cs + r * BN + c4);
    float x[4] = {v.x, v.y, v.z, v.w};
    const int nvalid = min(4, p.N - gn);
    if (EPI == UDM_EPI_BIAS || EPI == UDM_EPI_BIAS_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nvalid) x[e] += p.bias[gn + e];
    }
    if (EPI == UDM_EPI_BIAS_GELU) {
      // the pre-activation is rounded to bf16 (as the reference's mlp.0 output is) and GELU evaluated on it; aux receives bf16(gelu'(pre))
      bf16_t pre[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dg;
        gelu_tanh_both(bf2f(f2bf(x[e])), x[e], dg);
        pre[e] = f2bf(dg);
      }
      bf16_t* ap = p.aux + (long)gm * p.ldaux + gn;
      if (nvalid == 4 && (p.ldaux % 4 == 0)) {
        *reinterpret_cast<uint2*>(ap) = make_uint2((uint32_t)pre[0] | ((uint32_t)pre[1] << 16), (uint32_t)pre[2] | ((uint32_t)pre[3] << 16));
      } else {
        for (int e = 0; e < nvalid; ++e) ap[e] = pre[e];
      }
    }
    if (EPI == UDM_EPI_DGELU) {
      const bf16_t* ap = p.aux + (long)gm * p.ldaux + gn;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nvalid) {
          x[e] *= bf2f(ap[e]);
          if (p.bias) atomicAdd(const_cast<float*>(p.bias) + gn + e, OUT_F32 ? x[e] : bf2f(f2bf(x[e])));  // bias-gradient output (see header)
        }
    }
    if (OUT_F32) {
      float* cp = reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn;
      if (p.beta != 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nvalid) x[e] += p.beta * cp[e];
      }
      if (nvalid == 4 && n_vec_ok) {
        *reinterpret_cast<float4*>(cp) = make_float4(x[0], x[1], x[2], x[3]);
      } else {
        for (int e = 0; e < nvalid; ++e) cp[e] = x[e];
      }
    } else {
      bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (long)gm * p.ldc + gn;
      if (nvalid == 4 && n_vec_ok) {
        *reinterpret_cast<uint2*>(cp) = make_uint2(pack2bf(x[0], x[1]), pack2bf(x[2], x[3]));
      } else {
        for (int e = 0; e < nvalid; ++e) cp[e] = f2bf(x[e]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 transpose [R,C] -> [C,R] with optional fused column sums (bias gradient: db[c] += sum_r in[r,c]).
// 64x64 tiles through LDS; 16-byte global accesses on both sides.
// ---------------------------------------------------------------------------------------------
constexpr int TT = 64, TPAD = 8;
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C,
                                                            long ld_in, long ld_out, float* __restrict__ colsum) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[TT][TT + TPAD];
  const int r0 = blockIdx.y * TT, c0 = blockIdx.x * TT;
  const int tid = threadIdx.x;
  const int cs = (tid & 7) * 8;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = (tid >> 3) + pass * 32;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < R && c0 + cs < C) v = *reinterpret_cast<const uint4*>(in + (long)(r0 + r) * ld_in + c0 + cs);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[cs + k][r] = e[k];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = (tid >> 3) + pass * 32;
    const int rs = (tid & 7) * 8;
    uint4 v = *reinterpret_cast<const uint4*>(&tile[c][rs]);
    const bool ok = (c0 + c < C) && (r0 + rs < R);
    if (ok && out) *reinterpret_cast<uint4*>(out + (long)(c0 + c) * ld_out + r0 + rs) = v;
    if (colsum) {
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += bf2f(e[k]);  // rows beyond R were zero-filled
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if ((tid & 7) == 0 && c0 + c < C) atomicAdd(colsum + c0 + c, s);
    }
  }
}

// fp32 [R,C] -> bf16 [R,C] (ld_out) and/or bf16 transposed [C,R] (ld_t): the per-step weight shadow that
// autocast makes in the reference (cast of fp32 master weights), plus the K-major copy used by dgrad.
__device__ __forceinline__ void cast_transpose_tile(const float* __restrict__ in, bf16_t* __restrict__ out, bf16_t* __restrict__ out_t, int R, int C, long ld_in,
                                                    long ld_out, long ld_t, int r0, int c0, bf16_t (*tile)[TT + TPAD]) {
  const int tid = threadIdx.x;
  const int cs = (tid & 15) * 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = (tid >> 4) + pass * 16;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    const bool rok = r0 + r < R;
    if (rok) {
      const float* ip = in + (long)(r0 + r) * ld_in + c0 + cs;
      if (c0 + cs + 3 < C && (ld_in % 4 == 0)) {
        float4 v = *reinterpret_cast<const float4*>(ip);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      } else {
        for (int k = 0; k < 4; ++k)
          if (c0 + cs + k < C) x[k] = ip[k];
      }
    }
    bf16_t b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = f2bf(x[k]);
    if (out && rok) {
      bf16_t* op = out + (long)(r0 + r) * ld_out + c0 + cs;
      if (c0 + cs + 3 < C && (ld_out % 4 == 0)) {
        *reinterpret_cast<uint2*>(op) = make_uint2((uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16));
      } else {
        for (int k = 0; k < 4; ++k)
          if (c0 + cs + k < C) op[k] = b[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[cs + k][r] = b[k];
  }
  if (!out_t) return;   // (block-uniform)
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = (tid >> 3) + pass * 32;
    const int rs = (tid & 7) * 8;
    if (c0 + c >= C || r0 + rs >= R) continue;
    bf16_t* op = out_t + (long)(c0 + c) * ld_t + r0 + rs;
    if (r0 + rs + 7 < R && (ld_t % 8 == 0)) {
      *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(&tile[c][rs]);
    } else {
      for (int k = 0; k < 8; ++k)
        if (r0 + rs + k < R) op[k] = tile[c][rs + k];
    }
  }
}

__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, bf16_t* __restrict__ out_t,
                                                            int R, int C, long ld_in, long ld_out, long ld_t) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[TT][TT + TPAD];
  cast_transpose_tile(in, out, out_t, R, C, ld_in, ld_out, ld_t, blockIdx.y * TT, blockIdx.x * TT, tile);
}

// All weight casts of a forward in ONE launch (the reference's autocast casts every Linear weight once per forward, models/dit.py:1096-1109 under
// torch.autocast): 97 back-to-back launches of the single-matrix kernel spend a third of their time ramping up and draining (34 MB matrices
// run at 4.1 TB/s against 6.0 for 134 MB ones).  `jobs` is a device table sorted by first tile; a block finds its matrix by bisection.
struct CastJob {
  const float* in; bf16_t* out; bf16_t* out_t;
  long ld_in, ld_out, ld_t;
  int R, C, tile0, tiles_c;
};
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const CastJob* __restrict__ jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[TT][TT + TPAD];
  const int t = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
  }
  const CastJob j = jobs[lo];
  const int lt = t - j.tile0;
  cast_transpose_tile(j.in, j.out, j.out_t, j.R, j.C, j.ld_in, j.ld_out, j.ld_t, (lt / j.tiles_c) * TT, (lt % j.tiles_c) * TT, tile);
}


// =================================================================================================
// Large-shape path: BM x 256 x 64 tiles (BM = 192 / 256 / 320, chosen per shape so the tile count fills whole
// rounds of the 256 CUs), 8 waves as 2 x 4, operands staged by LDS-DMA (global_load_lds_dwordx4; the XOR swizzle is
// applied on the per-lane SOURCE address so the LDS image stays lane-linear), two 64-deep stages.
// Schedule: each K tile is four phases { 16-deep fragment ds_reads (+ a slice of the next tile's LDS-DMA) ;
// s_barrier ; MFMAs ; s_barrier } and waves 4-7 run ONE barrier behind waves 0-3.  A SIMD hosts wave w and
// wave w+4, so while one of them owns the matrix pipe its partner is in its read section: the pipe alternates
// between the two instead of idling while both sit at the same barrier.  Hazards:
//   RAW: a wave waits vmcnt(0) for its own LDS-DMA in the LAST phase of a tile before that phase's first barrier;
//        the first read of the new tile is at least one barrier later for both groups.
//   WAR: fragment reads are retired (lgkmcnt(0)) before each phase's first barrier, and a stage is re-staged
//        only from phase 0 of the following tile.
// Epilogue: per wave, each 32x32 accumulator fragment is transposed through a wave-private 4 KiB LDS patch so a
// lane owns 4 consecutive columns of a row (8/16-byte global accesses); bias / GELU / GELU' / beta are applied there.
// =================================================================================================
#ifndef UDM_EPI_WIDE
#define UDM_EPI_WIDE 1
#endif
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void glds16(const void* gptr, char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (UDM_LDS void*)lds_base, 16, 0, 0);
}

// LDS-DMA issued as inline asm (see the TN branch of the stagger kernel): lane i lands at lds + 16 i; `lds` must be wave-uniform
__device__ __forceinline__ void glds16_asm(const void* gptr, const char* lds) {
  const uint32_t dst = (uint32_t)(size_t)(UDM_LDS const char*)lds;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(dst) : "memory", "m0");
}

// TN = true: both operands are K-major ("A^T B": A is [K, M], B is [K, N], rows = contraction index) — the wgrad
// form dW = dY^T X read straight from the row-major activations.  Tiles are staged as [64 k-rows][columns] and the
// MFMA operands are gathered with ds_read_b64_tr_b16 transposing reads; rows are rotated by (k & 3) 64-byte
// granules (applied on the LDS-DMA source address) so the four k-rows of one transposing read hit different banks.
// PERSIST = true (NT form, whole tiles only, no split-K; grid = 256): a block walks its XCD's tiles instead of ending after one.  The last K
// iteration of a tile stages the FIRST K tile of the block's next output tile (the slot where a lone tile harmlessly re-stages itself), the
// epilogue's patches live in the stage that was consumed last, and the next main loop starts on data that is already in LDS: no block
// dispatch, no cold first load, and the epilogue's stores drain under the next tile's MFMAs.
template <int BMX, int EPI, bool OUT_F32, bool TN, int KKPP = (TN ? 2 : 1), bool PERSIST = false>  // KKPP = 16-deep k-steps per phase
__global__ __launch_bounds__(512) void gemm_nt_stagger_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BNX = 256, WGM = 2, WGN = 4, NWAVES = 8;
  constexpr int WM = BMX / WGM, WN = BNX / WGN, FM = WM / 32, FN = WN / 32;
  constexpr int A_BYTES = BMX * BK * 2, B_BYTES = BNX * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PW = A_BYTES / 1024 / NWAVES, B_PW = B_BYTES / 1024 / NWAVES, LOADS = A_PW + B_PW;
  // The transposing-read (TN) form issues twice as many LDS instructions per k-step, so its read section outlasts an 8-10 MFMA
  // section (measured: 40-50 % pipe utilisation vs 57 %); two k-steps per phase amortise the LDS latency there.
  constexpr int NPH = 4 / KKPP, ISSUE_PH = NPH > 2 ? 2 : 1, PER = (LOADS + ISSUE_PH - 1) / ISSUE_PH;
  static_assert(WM % 32 == 0 && (A_BYTES / 1024) % NWAVES == 0, "bad tile");

  static_assert(!PERSIST || !TN, "the persistent form is built for the NT kernels");
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int nwg = p.tiles_m * p.tiles_n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int late = __builtin_amdgcn_readfirstlane(wave >= NWAVES / 2 ? 1 : 0);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int grp_rows = p.group_m > 0 ? p.group_m : GROUP_M;
  const int per_group = grp_rows * p.tiles_n;
  int par = 0;           // PERSIST: stage that holds the first K tile of the current output tile
  bool primed = false;   // PERSIST: ... and it is already there (staged by the previous output tile's last K iteration)
  for (int wi = blockIdx.x;; wi += gridDim.x) {   // one pass unless PERSIST
  int pid = xcd_remap(wi, nwg * S);
  const int slice = pid % S;  // slices of one tile are neighbours in the remapped order: same XCD, shared operand panels
  pid /= S;
  const int grp = pid / per_group, first_m = grp * grp_rows;
  const int gsz = min(p.tiles_m - first_m, grp_rows);
  const int tm = first_m + (pid % per_group) % gsz, tn = (pid % per_group) / gsz;
  const int row0 = tm * BMX, col0 = tn * BNX;
  bool has_next = false;
  long next_da = 0, next_db = 0;   // element offsets from this tile's operand rows to the next tile's (whole tiles: no clamping in src[])
  if (PERSIST) {
    const int wn_i = wi + gridDim.x;
    has_next = wn_i < nwg;
    if (has_next) {
      const int pid2 = xcd_remap(wn_i, nwg);
      const int grp2 = pid2 / per_group, first2 = grp2 * grp_rows;
      const int gsz2 = min(p.tiles_m - first2, grp_rows);
      const int tm2 = first2 + (pid2 % per_group) % gsz2, tn2 = (pid2 % per_group) / gsz2;
      next_da = (long)(tm2 - tm) * BMX * p.lda;
      next_db = (long)(tn2 - tn) * BNX * p.ldb;
    }
  }
  const bf16_t* src[LOADS];
  int dst[LOADS];
  constexpr int RB_A = BMX * 2, RB_B = BNX * 2, NG_A = RB_A / 64, NG_B = RB_B / 64;  // TN: row bytes / 64-byte granules per k-row
  // TN granule rotation of k-row r (mod 4): the four k-rows of one transposing read must land in four different 64-byte bank groups of the
  // 256-byte LDS line.  Rows of a multiple of 256 bytes all start at the same bank: rotate by r.  Rows of 384 bytes (192-wide tile) alternate
  // between bank offsets 0 and 128, so r would collide whenever (granule + r) wraps past 6 (measured: bank conflicts on 23 % of the qkv
  // wgrad's LDS cycles); rotating by r >> 1 gives {g, g + 128 B, g + 1, g + 1 + 128 B} - distinct for every granule, wrap included.
#define ROT_A(r) ((NG_A % 4 == 0) ? (r) : ((r) >> 1))
#define ROT_B(r) ((NG_B % 4 == 0) ? (r) : ((r) >> 1))
  if (!TN) {
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
      const int r = (wave * A_PW + j) * 8 + lrow;
      src[j] = p.A + (long)min(row0 + r, p.M - 1) * p.lda + ((lslot ^ ((r >> 1) & 7)) << 3);
      dst[j] = (wave * A_PW + j) * 1024;
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
      const int r = (wave * B_PW + j) * 8 + lrow;
      src[A_PW + j] = p.B + (long)min(col0 + r, p.N - 1) * p.ldb + ((lslot ^ ((r >> 1) & 7)) << 3);
      dst[A_PW + j] = A_BYTES + (wave * B_PW + j) * 1024;
    }
  } else {
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
      const int off = (wave * A_PW + j) * 1024 + lane * 16;
      const int row = off / RB_A, within = off % RB_A;
      const int gran = (within / 64 + NG_A - ROT_A(row & 3)) % NG_A;
      // (ragged tiles: clamp to the last 8-column group that holds valid data - NOT to the row's end: A may be a column slice of a wider matrix, whose last k-row ends the allocation)
      const long col = min((long)row0 + (gran * 64 + within % 64) / 2, (long)((p.M - 1) / 8 * 8));
      src[j] = p.A + (long)row * p.lda + col;
      dst[j] = (wave * A_PW + j) * 1024;
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
      const int off = (wave * B_PW + j) * 1024 + lane * 16;
      const int row = off / RB_B, within = off % RB_B;
      const int gran = (within / 64 + NG_B - ROT_B(row & 3)) % NG_B;
      const long col = min((long)col0 + (gran * 64 + within % 64) / 2, (long)((p.N - 1) / 8 * 8));
      src[A_PW + j] = p.B + (long)row * p.ldb + col;
      dst[A_PW + j] = A_BYTES + (wave * B_PW + j) * 1024;
    }
  }
  const long kstep_a = TN ? (long)BK * p.lda : BK, kstep_b = TN ? (long)BK * p.ldb : BK;
  // TN fragment gather constants: 16-lane group g1 of the half-wave, lane p16 supplies k-row (p16 >> 2), columns 4*(p16 & 3)..
  const int p16 = lane & 15, g1 = (lane >> 4) & 1, rot = p16 >> 2;
  int a_fo[FM], b_fo[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_fo[i] = (((wm * WM + i * 32) / 32 + ROT_A(rot)) % NG_A) * 64 + g1 * 32 + (p16 & 3) * 8;
#pragma unroll
  for (int j = 0; j < FN; ++j) b_fo[j] = (((wn * WN + j * 32) / 32 + ROT_B(rot)) % NG_B) * 64 + g1 * 32 + (p16 & 3) * 8;
  const int sw = (l31 >> 1) & 7;
  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk_all = p.K / BK;
  const int kt0 = (int)((long)nk_all * slice / S), nk = (int)((long)nk_all * (slice + 1) / S);
  if (!(PERSIST && primed)) {
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
      glds16_asm(src[j] + kt0 * (j < A_PW ? kstep_a : kstep_b), smem + dst[j] + ((kt0 + par) & 1) * STAGE_BYTES);
    }
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();

  for (int kt = kt0; kt < nk; ++kt) {
    const char* As = smem + ((kt + par) & 1) * STAGE_BYTES;
    const char* Bs = As + A_BYTES;
    char* nxt = smem + ((kt + 1 + par) & 1) * STAGE_BYTES;
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      bf16x8_t a[KKPP][FM], b[KKPP][FN];
#pragma unroll
      for (int q = 0; q < KKPP; ++q) {
        const int kk = ph * KKPP + q;
        if (!TN) {
          const int so = ((kk * 2 + hi) ^ sw) << 4;
#pragma unroll
          for (int i = 0; i < FM; ++i) a[q][i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * WM + i * 32 + l31) * 128 + so);
#pragma unroll
          for (int j = 0; j < FN; ++j) b[q][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * WN + j * 32 + l31) * 128 + so);
        } else {
          const int kr = kk * 16 + hi * 8 + rot;
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(As + kr * RB_A + a_fo[i]));
            s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(As + (kr + 4) * RB_A + a_fo[i]));
            a[q][i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
          }
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(Bs + kr * RB_B + b_fo[j]));
            s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((UDM_LDS s16x4_t*)(Bs + (kr + 4) * RB_B + b_fo[j]));
            b[q][j] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ph == NPH - 1) wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      {
        // Refills of the next tile go out from INSIDE the MFMA section (an MFMA occupies the pipe for 32 cycles while the wave itself is
        // idle), one per GAP MFMAs, at places that are written out and pinned: a sched_group_barrier recipe cannot reach across the
        // `more` branch the refills used to sit behind (all of them went out in one burst before the first MFMA), so the last tile
        // re-stages itself into the free stage instead (harmless, waited for below) and the section has no branch at all.
        // They are inline asm on purpose: the compiler waits vmcnt(0) in front of every ds_read_b64_tr_b16 (TN form) that follows an
        // LDS-DMA it knows about, because it cannot prove the stages disjoint - i.e. one phase after the issue instead of at the end
        // of the tile.  The waits that matter are the explicit ones.
        constexpr int NMF = KKPP * FM * FN, GAP = (NMF / PER) > 0 ? (NMF / PER) : 1;
        const long knext = more ? kt + 1 : kt;
        // element offset of the refill's source from src[]: the next K tile; on the last K tile of a PERSIST block with another output tile
        // to do, the first K tile of THAT tile (kt0 = 0 there); otherwise the lone tile re-stages itself
        const long ref_a = (PERSIST && !more && has_next) ? next_da : knext * kstep_a;
        const long ref_b = (PERSIST && !more && has_next) ? next_db : knext * kstep_b;
        int jd = ph * PER, cnt = 0;
#pragma unroll
        for (int q = 0; q < KKPP; ++q)
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
              ++cnt;
              if (ph < ISSUE_PH && ((cnt - 1) % GAP) == 0 && jd < (ph + 1) * PER && jd < LOADS) {
                __builtin_amdgcn_sched_barrier(0);
                glds16_asm(src[jd] + (jd < A_PW ? ref_a : ref_b), nxt + dst[jd]);
                __builtin_amdgcn_sched_barrier(0);
                ++jd;
              }
            }
#pragma unroll
        for (; jd < (ph + 1) * PER && jd < LOADS && ph < ISSUE_PH; ++jd) glds16_asm(src[jd] + (jd < A_PW ? ref_a : ref_b), nxt + dst[jd]);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();
  __syncthreads();  // all LDS tile reads are done: the wave-private epilogue patches may overwrite stage memory

  // ---- epilogue ----
  // two 32 x 32 fp32 patches per wave (ping-pong), 64 KiB per block.  PERSIST: inside the stage of the LAST K tile - the other stage already
  // holds the next output tile's first K tile (needs 64 KiB <= STAGE_BYTES: BMX >= 256)
  static_assert(!PERSIST || STAGE_BYTES >= NWAVES * 8192, "persistent form: the epilogue patches must fit one stage");
  float* patch0 = reinterpret_cast<float*>(smem + (PERSIST ? ((nk - 1 + par) & 1) * STAGE_BYTES : 0)) + wave * 2048;
  const int er = lane >> 3, ec = (lane & 7) * 4;               // read side: this lane's row (within 8) and 4-column chunk
  const bool interior = (row0 + BMX <= p.M) && (col0 + BNX <= p.N) && (p.ldc % 4 == 0) && (EPI < UDM_EPI_BIAS_GELU || p.ldaux % 4 == 0);
  // EPI_DGELU + non-null `bias`: the pointer is an fp32 [N] OUTPUT that receives the column sums of the (bf16-rounded) result,
  // i.e. the bias gradient of the Linear whose dgrad this is — saves a separate pass over the [M, 4d] gradient.
  float* colsum = (EPI == UDM_EPI_DGELU) ? const_cast<float*>(p.bias) : nullptr;
  // Wide form (interior tiles, rows of C / aux 16-byte aligned): the wave's TWO 32-column fragments of a 32-row strip go through the two patches together and a
  // lane then owns 8 consecutive columns of the 64 - every global access of the epilogue is 16 bytes per lane and a whole 128-byte line per 8 lanes (the narrow
  // form below: 8 bytes per lane, 64-byte half lines, twice the instructions).  Patch j stores column c at c ^ 4j, so the lanes that read patch 0 and the lanes
  // that read patch 1 in the same instruction hit disjoint banks.
  // Compiled into the persistent form only (whole tiles and the alignment are launch conditions there; both forms in one kernel cost the main loop registers).
  constexpr bool wide = UDM_EPI_WIDE && PERSIST && FN == 2;
  if constexpr (wide) {
    const int s8 = lane & 7, jsel = s8 >> 2, ec8 = (s8 & 3) * 8;
    const int gn = col0 + wn * WN + s8 * 8;
    const float* rd = patch0 + jsel * 1024 + er * 32;
    const int ca = ec8 ^ (4 * jsel), cb = (ec8 + 4) ^ (4 * jsel);
    // per-lane row bases once per tile; every access below is base + (compile-time row) x (uniform leading dimension): scalar arithmetic, one 64-bit add per
    // access (with the row index per lane the compiler multiplied in 64 bits for every store: three quarter-rate integer multiplies per 8 elements and stream)
    const long lrow = (long)(row0 + wm * WM + er);
    char* cbase = reinterpret_cast<char*>(p.C) + ((S > 1 ? slice * p.slice_stride : 0) + lrow * p.ldc + gn) * (OUT_F32 ? 4 : 2);
    bf16_t* abase = (EPI >= UDM_EPI_BIAS_GELU) ? p.aux + lrow * p.ldaux + gn : nullptr;
    float bias8[8], csum8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bias8[e] = (EPI == UDM_EPI_BIAS || EPI == UDM_EPI_BIAS_GELU) ? p.bias[gn + e] : 0.f;
      csum8[e] = 0.f;
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < FM; ++i) {
#pragma clang loop unroll(full)
      for (int j = 0; j < 2; ++j)
#pragma clang loop unroll(full)
        for (int r = 0; r < 16; ++r) patch0[j * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + (l31 ^ (4 * j))] = acc[i][j][r];
      uint4 au[4];
      if (EPI == UDM_EPI_DGELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) au[q] = *reinterpret_cast<const uint4*>(abase + (long)(i * 32 + q * 8) * p.ldaux);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 va = *reinterpret_cast<const float4*>(rd + q * 256 + ca), vb = *reinterpret_cast<const float4*>(rd + q * 256 + cb);
        float x[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        if (EPI == UDM_EPI_BIAS || EPI == UDM_EPI_BIAS_GELU) {   // (not `+ 0.f` otherwise: the compiler keeps that add for the sign of zero)
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += bias8[e];
        }
        const long ro = (long)(i * 32 + q * 8);
        if (EPI == UDM_EPI_BIAS_GELU) {
          float dg[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gelu_tanh_both(bf2f(f2bf(x[e])), x[e], dg[e]);
          *reinterpret_cast<uint4*>(abase + ro * p.ldaux) = make_uint4(pack2bf(dg[0], dg[1]), pack2bf(dg[2], dg[3]), pack2bf(dg[4], dg[5]), pack2bf(dg[6], dg[7]));
        }
        if (EPI == UDM_EPI_DGELU) {
          const uint32_t w[4] = {au[q].x, au[q].y, au[q].z, au[q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[2 * e] *= __uint_as_float(w[e] << 16);
            x[2 * e + 1] *= __uint_as_float(w[e] & 0xffff0000u);
          }
          // (summed whether or not `colsum` is wanted: a block-uniform branch here costs more than the operations it would skip)
          if (OUT_F32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csum8[e] += x[e];
          }
        }
        if (OUT_F32) {
          float* cp = reinterpret_cast<float*>(cbase) + ro * p.ldc;
          if (S == 1 && p.beta != 0.f) {
            const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
            x[0] += p.beta * c0.x; x[1] += p.beta * c0.y; x[2] += p.beta * c0.z; x[3] += p.beta * c0.w;
            x[4] += p.beta * c1.x; x[5] += p.beta * c1.y; x[6] += p.beta * c1.z; x[7] += p.beta * c1.w;
          }
          *reinterpret_cast<float4*>(cp) = make_float4(x[0], x[1], x[2], x[3]);
          *reinterpret_cast<float4*>(cp + 4) = make_float4(x[4], x[5], x[6], x[7]);
        } else {
          const uint32_t pk[4] = {pack2bf(x[0], x[1]), pack2bf(x[2], x[3]), pack2bf(x[4], x[5]), pack2bf(x[6], x[7])};
          if (EPI == UDM_EPI_DGELU) {   // column sums of the ROUNDED values (what the bias gradient of the bf16 tensor is), from the words about to be stored
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              csum8[2 * e] += __uint_as_float(pk[e] << 16);
              csum8[2 * e + 1] += __uint_as_float(pk[e] & 0xffff0000u);
            }
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(cbase) + ro * p.ldc) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
      }
    }
    if (EPI == UDM_EPI_DGELU && colsum) {
      // column totals of the wave's 32 x FM rows: after the three exchanges every lane of a column group holds its group's eight totals; lane L then takes
      // column L of the wave's 64 (element L & 7 of group L >> 3) so that ONE atomic instruction per wave and tile adds 256 contiguous bytes (one instruction
      // per element covered 8 x 4 bytes at a 32-byte stride: eight times the instructions and four times the cache-line operations on addresses that every
      // row tile of the column contends for)
      float mine = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = csum8[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        mine = (er == e) ? v : mine;
      }
      const float tot = __shfl(mine, (lane >> 3) | ((lane & 7) << 3), 64);
      atomicAdd(colsum + col0 + wn * WN + lane, tot);
    }
  } else {
  float bias4[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gn = col0 + wn * WN + j * 32 + ec + e;
      bias4[j][e] = ((EPI == UDM_EPI_BIAS || EPI == UDM_EPI_BIAS_GELU) && gn < p.N) ? p.bias[gn] : 0.f;
    }
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
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[i][j][r];
      const int gn = col0 + wn * WN + j * 32 + ec;
      const int gm0 = row0 + wm * WM + i * 32 + er;
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(patch + (q * 8 + er) * 32 + ec);
      if (interior) {
        uint2 au[4];
        float4 cold[4];
        if (EPI == UDM_EPI_DGELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) au[q] = *reinterpret_cast<const uint2*>(p.aux + (long)(gm0 + q * 8) * p.ldaux + gn);
        }
        if (OUT_F32 && p.beta != 0.f && S == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) cold[q] = *reinterpret_cast<const float4*>(reinterpret_cast<float*>(p.C) + (long)(gm0 + q * 8) * p.ldc + gn);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x[4] = {v[q].x + bias4[j][0], v[q].y + bias4[j][1], v[q].z + bias4[j][2], v[q].w + bias4[j][3]};
          const long gm = gm0 + q * 8;
          if (EPI == UDM_EPI_BIAS_GELU) {
            bf16_t pre[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float dg; gelu_tanh_both(bf2f(f2bf(x[e])), x[e], dg); pre[e] = f2bf(dg); }
            *reinterpret_cast<uint2*>(p.aux + gm * p.ldaux + gn) = make_uint2((uint32_t)pre[0] | ((uint32_t)pre[1] << 16), (uint32_t)pre[2] | ((uint32_t)pre[3] << 16));
          }
          if (EPI == UDM_EPI_DGELU) {
            x[0] *= __uint_as_float(au[q].x << 16); x[1] *= __uint_as_float(au[q].x & 0xffff0000u);
            x[2] *= __uint_as_float(au[q].y << 16); x[3] *= __uint_as_float(au[q].y & 0xffff0000u);
          }
          if (EPI == UDM_EPI_DGELU && colsum) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[j][e] += OUT_F32 ? x[e] : bf2f(f2bf(x[e]));
          }
          if (OUT_F32 && S > 1) {
            float* cp = reinterpret_cast<float*>(p.C) + slice * p.slice_stride + gm * p.ldc + gn;
            if (p.slice_stride) {
              *reinterpret_cast<float4*>(cp) = make_float4(x[0], x[1], x[2], x[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) atomicAdd(cp + e, x[e]);
            }
          } else if (OUT_F32) {
            if (p.beta != 0.f) { x[0] += p.beta * cold[q].x; x[1] += p.beta * cold[q].y; x[2] += p.beta * cold[q].z; x[3] += p.beta * cold[q].w; }
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + gm * p.ldc + gn) = make_float4(x[0], x[1], x[2], x[3]);
          } else {
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + gm * p.ldc + gn) = make_uint2(pack2bf(x[0], x[1]), pack2bf(x[2], x[3]));
          }
        }
      } else {  // edge tiles: element-wise predication
        for (int q = 0; q < 4; ++q) {
          const long gm = gm0 + q * 8;
          if (gm >= p.M || gn >= p.N) continue;
          float x[4] = {v[q].x + bias4[j][0], v[q].y + bias4[j][1], v[q].z + bias4[j][2], v[q].w + bias4[j][3]};
          const int nvalid = min(4, p.N - gn);
          for (int e = 0; e < nvalid; ++e) {
            if (EPI == UDM_EPI_BIAS_GELU) {
              float dg;
              gelu_tanh_both(bf2f(f2bf(x[e])), x[e], dg);
              p.aux[gm * p.ldaux + gn + e] = f2bf(dg);
            }
            if (EPI == UDM_EPI_DGELU) {
              x[e] *= bf2f(p.aux[gm * p.ldaux + gn + e]);
              if (colsum) atomicAdd(colsum + gn + e, OUT_F32 ? x[e] : bf2f(f2bf(x[e])));
            }
            if (OUT_F32 && S > 1) {
              float* cp = reinterpret_cast<float*>(p.C) + slice * p.slice_stride + gm * p.ldc + gn + e;
              if (p.slice_stride) *cp = x[e];
              else atomicAdd(cp, x[e]);
            } else if (OUT_F32) {
              float* cp = reinterpret_cast<float*>(p.C) + gm * p.ldc + gn + e;
              *cp = x[e] + (p.beta != 0.f ? p.beta * *cp : 0.f);
            } else {
              reinterpret_cast<bf16_t*>(p.C)[gm * p.ldc + gn + e] = f2bf(x[e]);
            }
          }
        }
      }
    }
  if (EPI == UDM_EPI_DGELU && colsum && interior) {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = csum[j][e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (er == 0) atomicAdd(colsum + col0 + wn * WN + j * 32 + ec + e, v);
      }
  }
  }  // narrow form
  if (!PERSIST || !has_next) break;
  par = (nk + par) & 1;   // where the last K iteration put the next tile's first K tile
  primed = true;
  }  // output tiles of this block
}

#undef ROT_A
#undef ROT_B
int g_gemm_persist = 1;   // diagnostics (UDM_GEMM_PERSIST=0): 0 = one block per output tile everywhere
// Data-parallel runs: the persistent blocks of a multi-round NT GEMM occupy every CU for the whole launch, and RCCL's channel kernels (the
// gradient all-reduce overlapped with backward) then only get CUs between launches.  UDM_GEMM_CUS = n (or udm_gemm_set_cus) caps the
// persistent grid at n blocks (a multiple of 8: one block per CU, XCD round-robin), leaving 256 - n CUs to the collective.  0 = all 256.
int g_gemm_cus = 0;
int gemm_cus_available() {   // (also reached from gemm_quad.hip through udm_gemm_cus_available below)
  static const bool env_once = [] {
    if (const char* e = getenv("UDM_GEMM_CUS")) { const int n = atoi(e); if (n >= 8 && n <= 256) g_gemm_cus = n / 8 * 8; }
    return true;
  }();
  (void)env_once;
  return g_gemm_cus ? g_gemm_cus : 256;
}

int g_gemm_persist_fwd() {   // the persistent-form switch with its environment default applied (UDM_GEMM_PERSIST, read once)
  static const bool env_once = [] {
    if (const char* e = getenv("UDM_GEMM_PERSIST")) g_gemm_persist = atoi(e);
    return true;
  }();
  (void)env_once;
  return g_gemm_persist;
}
// Does this launch take the persistent wide form (256 blocks walking whole tiles, 16-byte epilogue accesses)?
template <int BMX, int EPI, bool OUT_F32>
bool persistent_wide_ok(const GemmArgs& a) {
  if (BMX < 256) return false;
  const bool aux_ok = EPI < UDM_EPI_BIAS_GELU || a.ldaux % 8 == 0;
  return g_gemm_persist_fwd() && a.splitk <= 1 && (long)((a.M + BMX - 1) / BMX) * ((a.N + 255) / 256) > 256 && a.M % BMX == 0 && a.N % 256 == 0 && a.K / BK >= 2 &&
         (!UDM_EPI_WIDE || ((OUT_F32 ? a.ldc % 4 == 0 : a.ldc % 8 == 0) && aux_ok));
}

template <int BMX, int EPI, bool OUT_F32, bool TN = false>
int launch_big_t(const GemmArgs& a0, hipStream_t stream) {
  GemmArgs a = a0;
  a.tiles_m = (a.M + BMX - 1) / BMX;
  a.tiles_n = (a.N + 255) / 256;
  static const int env_gm = [] { const char* e = getenv("UDM_GEMM_GROUP_M"); return e ? atoi(e) : 0; }();
  a.group_m = env_gm;
  a.exp = udm_exp_flags();
  const size_t lds = (size_t)2 * (BMX + 256) * BK * 2;
  auto kern = gemm_nt_stagger_kernel<BMX, EPI, OUT_F32, TN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if constexpr (!TN && BMX >= 256) {
    // more than one round of whole tiles: 256 persistent blocks walk them (see PERSIST above)
    (void)gemm_cus_available();   // (applies UDM_GEMM_CUS once)
    if (persistent_wide_ok<BMX, EPI, OUT_F32>(a)) {
      auto kp = gemm_nt_stagger_kernel<BMX, EPI, OUT_F32, false, 1, true>;
      static bool attr_p = false;
      if (!attr_p) {
        (void)hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_p = true;
      }
      hipLaunchKernelGGL(kp, dim3(g_gemm_cus ? g_gemm_cus : 256), dim3(512), lds, stream, a);
      UDM_CHECK_LAUNCH("udm_gemm_nt_bf16(big, persistent)");
      return 0;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n * (a.splitk > 1 ? a.splitk : 1)), dim3(512), lds, stream, a);
  UDM_CHECK_LAUNCH("udm_gemm_nt_bf16(big)");
  return 0;
}
template <int BMX>
int launch_big(const GemmArgs& a, int epi, int out_f32, hipStream_t stream) {
  switch (epi) {
    case UDM_EPI_NONE: return out_f32 ? launch_big_t<BMX, UDM_EPI_NONE, true>(a, stream) : launch_big_t<BMX, UDM_EPI_NONE, false>(a, stream);
    case UDM_EPI_BIAS: return out_f32 ? launch_big_t<BMX, UDM_EPI_BIAS, true>(a, stream) : launch_big_t<BMX, UDM_EPI_BIAS, false>(a, stream);
    case UDM_EPI_BIAS_GELU: return launch_big_t<BMX, UDM_EPI_BIAS_GELU, false>(a, stream);
    default: return out_f32 ? launch_big_t<BMX, UDM_EPI_DGELU, true>(a, stream) : launch_big_t<BMX, UDM_EPI_DGELU, false>(a, stream);
  }
}

// Pick the tile whose tile count wastes the least of the 256-CU rounds.  Returns 0 for the 128x128 kernel.
int g_force_tile = -1;  // diagnostics (udm_gemm_set_tile): -1 auto, 0 small kernel, 192/256/320
int choose_tile(long M, long N, long K, long lda, long ldb) {
  if (K % 64 != 0 || K < 128) return 0;  // the LDS-DMA path has no K-tail handling: such shapes take the register-staged kernel
  if (g_force_tile >= 0) return g_force_tile;
  if (M < 192 || N < 192) return 0;
  const double eff[4] = {0.74, 0.93, 1.0, 1.0};  // measured relative throughput of {128x128, 192, 256, 320} x 256 tiles at full occupancy
  const int bms[4] = {128, 192, 256, 320};
  double best = 1e30;
  int pick = 0;
  for (int c = 0; c < 4; ++c) {
    const long tm = (M + bms[c] - 1) / bms[c];
    const long tn = c == 0 ? (N + 127) / 128 : (N + 255) / 256;
    const long slots = c == 0 ? 512 : 256;  // co-resident blocks on the chip
    const long rounds = (tm * tn + slots - 1) / slots;
    double t = rounds * (double)bms[c] * (c == 0 ? 128 : 256) * (c == 0 ? 2.0 : 1.0) / eff[c];  // time ~ rounds x tile area (2 blocks/CU share a CU)
    // short contractions (UniDisc-S: K = 768, 12 K tiles per output tile): prologue and epilogue are a large part of a tile, and only the persistent
    // form (256- / 320-row tiles, whole tiles, more than one round) overlaps the next tile's first loads with them.  Measured at M = 24576
    // (scripts/bench_gemm_s.py): N = 3072 + GELU 162 us persistent 256 vs 178 (192) / 194 (320, ragged); + GELU' 166 vs 187 / 228; N = 2304 plain 111 vs 120 / 114.
    const bool persistent = c >= 2 && M % bms[c] == 0 && N % 256 == 0 && tm * tn > 256;
    if (K <= 1024 && !persistent) t *= 1.15;
    // ... and at any K the persistent form (next tile's first loads under the epilogue, wide epilogue) is worth ~10 % of a launch: M = 9216 (config E), K = 2048,
    // N = 8192 + GELU: 293 us as 5 rounds of persistent 256-row tiles against 313 us as 6 rounds of 192-row tiles (GELU' + column sums 290 vs 316); N = 6144 plain:
    // 207 us (4 rounds of 256) against 217 (3 rounds of ragged 320) - scripts/bench_gemm_epi.py with M / N / TILE from the environment
    static const int env_pb = [] { const char* e = getenv("UDM_GEMM_PERSIST_BONUS"); return e ? atoi(e) : 1; }();   // diagnostics: 0 = the round-2 cost model
    if (persistent && env_pb) t *= 0.90;
    if (t < best || (t == best && c > 0)) { best = t; pick = c == 0 ? 0 : bms[c]; }  // ties go to the larger tile (fewer operand re-reads)
  }
  return pick;
}

template <int EPI>
int launch_gemm(const GemmArgs& a, int out_f32, hipStream_t stream) {
  dim3 grid(a.tiles_m * a.tiles_n), block(NTHREADS);
  const size_t lds = 4 * TILE_BYTES;
  if (out_f32)
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, true>), grid, block, lds, stream, a);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, false>), grid, block, lds, stream, a);
  UDM_CHECK_LAUNCH("udm_gemm_nt_bf16");
  return 0;
}
}  // namespace

extern "C" int udm_gemm_nt_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                int out_f32, int epilogue, const float* bias, void* aux, int64_t ldaux, float beta, hipStream_t stream) {
  UDM_CHECK_ARG(A && B && C, "udm_gemm_nt_bf16: null operand");
  UDM_CHECK_ARG(M > 0 && N > 0 && K > 0, "udm_gemm_nt_bf16: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  UDM_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "udm_gemm_nt_bf16: K, lda, ldb must be multiples of 8 (16-byte rows); K=%ld lda=%ld ldb=%ld",
                (long)K, (long)lda, (long)ldb);
  UDM_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), "udm_gemm_nt_bf16: operands must be 16-byte aligned");
  UDM_CHECK_ARG(beta == 0.f || out_f32, "udm_gemm_nt_bf16: beta accumulate needs fp32 output");
  UDM_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "udm_gemm_nt_bf16: dimension too large");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.bias = bias; a.aux = (bf16_t*)aux;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + BM - 1) / BM); a.tiles_n = (int)((N + BN - 1) / BN);
  a.beta = beta;
  a.splitk = 1;
  a.slice_stride = 0;
  if (epilogue == UDM_EPI_BIAS || epilogue == UDM_EPI_BIAS_GELU) UDM_CHECK_ARG(bias, "udm_gemm_nt_bf16: bias epilogue without bias");
  if (epilogue == UDM_EPI_BIAS_GELU) UDM_CHECK_ARG(aux && !out_f32, "udm_gemm_nt_bf16: EPI_BIAS_GELU needs aux and bf16 output");
  if (epilogue == UDM_EPI_DGELU) UDM_CHECK_ARG(aux, "udm_gemm_nt_bf16: EPI_DGELU needs aux (the saved GELU derivative)");
  UDM_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "udm_gemm_nt_bf16: unknown epilogue %d", epilogue);
  {  // whole tiles: the one-wave-per-SIMD kernel (gemm_quad.hip) where its tile count fills rounds of the chip at least as well
    int fm = 0;
    if (g_force_tile < 0 && (beta == 0.f) && udm_quad_nt_ok(M, N, K, &fm)) {
      const long qt = (M / (64 * fm)) * (N / 256);
      const int t8 = choose_tile(M, N, K, lda, ldb);
      const long t8n = t8 ? ((M + t8 - 1) / t8) * ((N + 255) / 256) : 0;
      const double q_cost = (double)((qt + 255) / 256) * 64 * fm, o_cost = t8 ? (double)((t8n + 255) / 256) * t8 : 1e30;
      // measured (scripts/bench_gemm_quad.py, 1.4 B shapes, random operands): the quad kernel wins 1-3 % on single-round shapes with a plain or
      // bias epilogue and loses 3 % where the GELU / GELU' epilogue runs (one wave per SIMD has nothing to overlap its VALU with);
      // multi-round shapes stay with the persistent 8-wave blocks
      if (udm_quad_mode() == 2 || (qt >= 128 && qt <= 256 && q_cost <= o_cost && epilogue <= UDM_EPI_BIAS)) {
        QuadArgs q{};
        q.A = a.A; q.B = a.B; q.C = C; q.bias = bias; q.aux = (bf16_t*)aux; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.ldaux = ldaux;
        q.M = a.M; q.N = a.N; q.K = a.K; q.beta = 0.f; q.splitk = 1;
        if (ldc % 4 == 0 && (epilogue < UDM_EPI_BIAS_GELU || ldaux % 4 == 0) && !(out_f32 && epilogue != UDM_EPI_NONE))
          return udm_quad_launch_nt(q, fm, epilogue, out_f32, stream);
      }
    }
  }
  {  // one round of 320-row tiles with a ragged last tile row (config E: M = 9216, N = 2048): the one-wave-per-SIMD kernel's ragged form instead of the 8-wave one
    int fm = 0;
    const long rt = ((M + 319) / 320) * ((N + 255) / 256);
    if (g_force_tile < 0 && beta == 0.f && !out_f32 && epilogue <= UDM_EPI_BIAS && M % 320 != 0 && N % 256 == 0 && rt >= 128 && rt <= 256 && ldc % 4 == 0 &&
        choose_tile(M, N, K, lda, ldb) == 320 && udm_quad_nn_ok(M, N, K, &fm) && fm == -5) {
      QuadArgs q{};
      q.A = a.A; q.B = a.B; q.C = C; q.bias = bias; q.lda = lda; q.ldb = ldb; q.ldc = ldc;
      q.M = a.M; q.N = a.N; q.K = a.K; q.beta = 0.f; q.splitk = 1;
      return udm_quad_launch_nt(q, -5, epilogue, 0, stream);
    }
  }
  switch (choose_tile(M, N, K, lda, ldb)) {
    case 192: return launch_big<192>(a, epilogue, out_f32, stream);
    case 256: return launch_big<256>(a, epilogue, out_f32, stream);
    case 320: return launch_big<320>(a, epilogue, out_f32, stream);
    default: break;
  }
  switch (epilogue) {
    case UDM_EPI_NONE: return launch_gemm<UDM_EPI_NONE>(a, out_f32, stream);
    case UDM_EPI_BIAS:
      UDM_CHECK_ARG(bias, "udm_gemm_nt_bf16: EPI_BIAS without bias");
      return launch_gemm<UDM_EPI_BIAS>(a, out_f32, stream);
    case UDM_EPI_BIAS_GELU:
      UDM_CHECK_ARG(bias && aux && !out_f32, "udm_gemm_nt_bf16: EPI_BIAS_GELU needs bias, aux and bf16 output");
      return launch_gemm<UDM_EPI_BIAS_GELU>(a, out_f32, stream);
    case UDM_EPI_DGELU:
      UDM_CHECK_ARG(aux, "udm_gemm_nt_bf16: EPI_DGELU needs aux (the saved GELU derivative)");
      return launch_gemm<UDM_EPI_DGELU>(a, out_f32, stream);
    default: udm_set_error("udm_gemm_nt_bf16: unknown epilogue %d", epilogue); return 2;
  }
}

extern "C" int udm_gemm_tn_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, float beta,
                                hipStream_t stream) {
  // C[M,N] (fp32) = beta*C + A[K,M]^T B[K,N]
  UDM_CHECK_ARG(A && B && C, "udm_gemm_tn_bf16: null operand");
  UDM_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 64 == 0, "udm_gemm_tn_bf16: K must be a positive multiple of 64 (got M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
  UDM_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N && lda >= 8 && ldb >= 8, "udm_gemm_tn_bf16: lda/ldb must be multiples of 8 and cover the rows");
  UDM_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), "udm_gemm_tn_bf16: operands must be 16-byte aligned");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.bias = nullptr; a.aux = nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = 0;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.beta = beta;
  a.splitk = 1;
  a.slice_stride = 0;
  {  // whole tiles that fill the chip: the one-wave-per-SIMD kernel (gemm_quad.hip)
    int fm = 0;
    if (g_force_tile < 0 && udm_quad_tn_ok(M, N, K, &fm) && ((M / (64 * fm)) * (N / 256) >= 128 || udm_quad_mode() == 2)) {
      QuadArgs q{};
      q.A = a.A; q.B = a.B; q.C = C; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.M = a.M; q.N = a.N; q.K = a.K; q.beta = beta; q.splitk = 1;
      return udm_quad_launch_tn(q, fm, stream);
    }
  }
  int tile = choose_tile(M, N, K, lda, ldb);
  if (tile == 0) tile = M <= 192 ? 192 : 256;  // the K-major path has no small-tile kernel; the large one handles any M, N by clamping
  if (beta == 1.0f) {  // accumulate form: few output tiles over a long K (e.g. the 2048x2048 out-proj wgrad, K = B*L) -> split K, atomically add
    const long tiles = ((M + tile - 1) / tile) * ((N + 255) / 256);
    const long nkt = K / 64;
    int sk = 1;
    // Measured on MI355X (2048x2048 output, K = 10240): 4-way split + fp32 atomics is SLOWER (0.31 ms) than one slice per tile on a
    // quarter of the CUs (0.24 ms) — 16.8 M L2 atomics cost more than the idle CUs.  Only split when the output is tiny.
    while (tiles * sk * 2 <= 32 && nkt / (sk * 2) >= 16 && sk < 8) sk *= 2;
    a.splitk = sk;
  }
  switch (tile) {
    case 192: return launch_big_t<192, UDM_EPI_NONE, true, true>(a, stream);
    case 320: return launch_big_t<320, UDM_EPI_NONE, true, true>(a, stream);
    default: return launch_big_t<256, UDM_EPI_NONE, true, true>(a, stream);
  }
}

namespace {
// out = beta * out + sum_s ws[s]: finishes a workspace split-K GEMM (fp32, 16-byte accesses)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long n4, int S, long stride, float beta) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 acc = reinterpret_cast<const float4*>(ws)[i];
    for (int s = 1; s < S; ++s) {
      const float4 v = reinterpret_cast<const float4*>(ws + s * stride)[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (beta != 0.f) {
      const float4 o = reinterpret_cast<const float4*>(out)[i];
      acc.x += beta * o.x; acc.y += beta * o.y; acc.z += beta * o.z; acc.w += beta * o.w;
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}
// the same sum for a bf16 result with a row stride (NT split-K below)
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ ws, bf16_t* __restrict__ out, long M, int N, long ldc, int S, long stride) {
  const int n4 = N / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M * n4; i += (long)gridDim.x * 256) {
    const long r = i / n4;
    const int c = (int)(i % n4) * 4;
    float4 acc = *reinterpret_cast<const float4*>(ws + r * N + c);
    for (int s = 1; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(ws + s * stride + r * N + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<uint2*>(out + r * ldc + c) = make_uint2(pack2bf(acc.x, acc.y), pack2bf(acc.z, acc.w));
  }
}
}  // namespace

// C[M,N] (bf16) = A[M,K] B[N,K]^T for FEW output tiles over a LONG contraction (the vocabulary-head dgrad on the compacted [MASK] rows:
// ~5 k x 2048 outputs over K = 48 512): 320 x 256 tiles, the K range cut into slices so that tiles x slices fill the 256 CUs once, fp32 partial
// tiles in `ws` (>= slices * M * N), a reduce pass that rounds to bf16.  Falls back to udm_gemm_nt_bf16 when splitting does not apply.
extern "C" int udm_gemm_nt_splitk_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                       float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(A && B && C, "udm_gemm_nt_splitk_bf16: null operand");
  UDM_CHECK_ARG(M > 0 && N > 0 && K > 0, "udm_gemm_nt_splitk_bf16: empty problem");
  const long tiles = ((M + 319) / 320) * ((N + 255) / 256);
  const long nkt = K / 64;
  long skl = gemm_cus_available() / tiles;
  if (skl > nkt / 8) skl = nkt / 8;
  if (skl > 32) skl = 32;
  const int sk = skl < 2 ? 1 : (int)skl;
  if (sk == 1 || K % 64 != 0 || !ws || ws_elems < (int64_t)sk * M * N || N % 4 != 0 || ldc % 4 != 0 || M < 320 || N < 256)
    return udm_gemm_nt_bf16(A, B, C, M, N, K, lda, ldb, ldc, 0, UDM_EPI_NONE, nullptr, nullptr, 0, 0.f, stream);
  UDM_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0, "udm_gemm_nt_splitk_bf16: lda/ldb must be multiples of 8");
  UDM_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 8 == 0) && ((uintptr_t)ws % 16 == 0), "udm_gemm_nt_splitk_bf16: operand alignment");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = ws; a.bias = nullptr; a.aux = nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = N; a.ldaux = 0;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.beta = 0.f;
  a.splitk = sk;
  a.slice_stride = (long)M * N;
  if (int rc = launch_big_t<320, UDM_EPI_NONE, true, false>(a, stream)) return rc;
  const long n4 = (long)M * N / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(grid), dim3(256), 0, stream, (const float*)ws, (bf16_t*)C, (long)M, (int)N, (long)ldc, sk, (long)M * N);
  UDM_CHECK_LAUNCH("udm_gemm_nt_splitk_bf16(reduce)");
  return 0;
}

// C[M,N] (fp32) = beta*C + A[K,M]^T B[K,N] for FEW output tiles over a LONG contraction (the 2048 x 2048 out-proj wgrad, K = B*L): the K
// range is cut into slices so that tiles x slices fill the 256 CUs, every slice writes its partial tile into `ws` (no atomics: a 4-way
// atomic split measured slower than leaving 3/4 of the CUs idle) and a reduce pass sums them.  Falls back to udm_gemm_tn_bf16 when the
// workspace is too small or splitting does not pay.  C must be contiguous (ldc == N).
extern "C" int udm_gemm_tn_splitk_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                       float beta, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(A && B && C, "udm_gemm_tn_splitk_bf16: null operand");
  UDM_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 64 == 0, "udm_gemm_tn_splitk_bf16: K must be a positive multiple of 64 (got M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
  int fm = 0;
  bool quad = g_force_tile < 0 && udm_quad_tn_ok(M, N, K, &fm);   // whole 256- or 192-row tiles of the one-wave-per-SIMD kernel
  if (quad && M % 256 == 0) fm = 4;   // (256-row tiles wherever they are whole: fewer partial-tile bytes per flop; UniDisc-S 23.84 vs 23.97 ms with the round-count pick)
  const long tiles = quad ? (M / (64 * fm)) * (N / 256) : ((M + 255) / 256) * ((N + 255) / 256);
  const long nkt = K / 64;
  // as many slices as fill the CUs once (256, or the cap of udm_gemm_set_cus; any count: slices may be uneven), each at least 8 K tiles deep, at most 32
  long skl = gemm_cus_available() / tiles;
  if (skl > nkt / 8) skl = nkt / 8;
  if (skl > 32) skl = 32;
  const int sk = skl < 2 ? 1 : (int)skl;
  if (sk == 1 || !ws || ws_elems < (int64_t)sk * M * N || ldc != N || (M * N) % 4 != 0)
    return udm_gemm_tn_bf16(A, B, C, M, N, K, lda, ldb, ldc, beta, stream);
  UDM_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N, "udm_gemm_tn_splitk_bf16: lda/ldb must be multiples of 8 and cover the rows");
  UDM_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) && ((uintptr_t)ws % 16 == 0), "udm_gemm_tn_splitk_bf16: operands must be 16-byte aligned");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = ws; a.bias = nullptr; a.aux = nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = N; a.ldaux = 0;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.beta = 0.f;
  a.splitk = sk;
  a.slice_stride = (long)M * N;
  if (quad) {   // same slices, one-wave-per-SIMD tiles
    QuadArgs q{};
    q.A = a.A; q.B = a.B; q.C = ws; q.lda = lda; q.ldb = ldb; q.ldc = N; q.M = a.M; q.N = a.N; q.K = a.K; q.beta = 0.f; q.splitk = sk;
    q.slice_stride = (long)M * N;
    if (int rc = udm_quad_launch_tn(q, fm, stream)) return rc;
  } else if (int rc = launch_big_t<256, UDM_EPI_NONE, true, true>(a, stream)) return rc;
  const long n4 = (long)M * N / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, (const float*)ws, (float*)C, n4, sk, (long)M * N, beta);
  UDM_CHECK_LAUNCH("udm_gemm_tn_splitk_bf16(reduce)");
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int udm_gemm_set_tile(int tile) {
  UDM_CHECK_ARG(tile == -1 || tile == 0 || tile == 192 || tile == 256 || tile == 320, "udm_gemm_set_tile: tile must be -1 (auto), 0, 192, 256 or 320");
  g_force_tile = tile;
  return 0;
}

extern "C" int udm_transpose_bf16(const void* in, void* out, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, float* colsum, hipStream_t stream) {
  UDM_CHECK_ARG(in && (out || colsum), "udm_transpose_bf16: null pointer");
  UDM_CHECK_ARG(R > 0 && C > 0, "udm_transpose_bf16: empty");
  UDM_CHECK_ARG(R % 8 == 0 && C % 8 == 0 && ld_in % 8 == 0 && (!out || ld_out % 8 == 0), "udm_transpose_bf16: R, C and strides must be multiples of 8");
  dim3 grid((unsigned)((C + TT - 1) / TT), (unsigned)((R + TT - 1) / TT));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, (int)R, (int)C, (long)ld_in, (long)ld_out, colsum);
  UDM_CHECK_LAUNCH("udm_transpose_bf16");
  return 0;
}

extern "C" int udm_cast_transpose_f32_bf16(const float* in, void* out, void* out_t, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, int64_t ld_t,
                                           hipStream_t stream) {
  UDM_CHECK_ARG(in && (out || out_t), "udm_cast_transpose_f32_bf16: null pointer");
  UDM_CHECK_ARG(R > 0 && C > 0, "udm_cast_transpose_f32_bf16: empty");
  dim3 grid((unsigned)((C + TT - 1) / TT), (unsigned)((R + TT - 1) / TT));
  hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, stream, in, (bf16_t*)out, (bf16_t*)out_t, (int)R, (int)C, (long)ld_in, (long)ld_out, (long)ld_t);
  UDM_CHECK_LAUNCH("udm_cast_transpose_f32_bf16");
  return 0;
}

extern "C" int udm_cast_transpose_multi_f32_bf16(const void* jobs, int64_t njobs, int64_t total_tiles, hipStream_t stream) {
  UDM_CHECK_ARG(jobs && njobs > 0 && total_tiles > 0 && total_tiles < (1LL << 31), "udm_cast_transpose_multi_f32_bf16: bad job table");
  hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, (const CastJob*)jobs, (int)njobs);
  UDM_CHECK_LAUNCH("udm_cast_transpose_multi_f32_bf16");
  return 0;
}

// C[M, N] bf16 = A[M, K] B[K, N] (A row-major with K contiguous, B row-major with N contiguous): the dgrad dX = dY W read from the forward's own
// bf16 shadow of W [out, in], so that no transposed shadow has to be produced by the per-step weight cast.  Whole tiles, or 320-row tiles with a ragged last tile row (udm_gemm_nn_ok).
extern "C" int udm_gemm_nn_ok(int64_t M, int64_t N, int64_t K) {
  int fm = 0;
  return (M > 0 && udm_quad_nn_ok(M, N, K, &fm)) ? 1 : 0;
}
extern "C" int udm_gemm_nn_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, hipStream_t stream) {
  UDM_CHECK_ARG(A && B && C, "udm_gemm_nn_bf16: null pointer");
  int fm = 0;
  UDM_CHECK_ARG(M > 0 && udm_quad_nn_ok(M, N, K, &fm), "udm_gemm_nn_bf16: shape %ld x %ld x %ld does not fit the NN kernel (N %% 256, K %% 64, M %% 192/256/320 or enough 320-row tiles to fill the chip; see udm_gemm_nn_ok)",
                (long)M, (long)N, (long)K);
  UDM_CHECK_ARG(lda >= K && ldb >= N && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "udm_gemm_nn_bf16: bad leading dimensions");
  QuadArgs q{};
  q.A = (const bf16_t*)A; q.B = (const bf16_t*)B; q.C = C; q.lda = lda; q.ldb = ldb; q.ldc = ldc;
  q.M = (int)M; q.N = (int)N; q.K = (int)K; q.beta = 0.f; q.splitk = 1;
  return udm_quad_launch_nn(q, fm, stream);
}

// udm_gemm_nn_bf16 for FEW output tiles: the K range cut into slices so that tiles x slices fill the available CUs once, fp32 partial tiles in `ws`
// (>= slices * M * N), a reduce pass that rounds to bf16.  Its use: the leftover tile rows of a dgrad whose 256 tiles no longer fit one round because a
// collective holds CUs (udm_gemm_set_cus / UDM_GEMM_CUS; kernels.py splits the rows).  Whole tiles only; falls back to udm_gemm_nn_bf16 when splitting does not apply.
extern "C" int udm_gemm_nn_splitk_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                       float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(A && B && C, "udm_gemm_nn_splitk_bf16: null pointer");
  int fm = 0;
  UDM_CHECK_ARG(M > 0 && udm_quad_nn_ok(M, N, K, &fm), "udm_gemm_nn_splitk_bf16: shape %ld x %ld x %ld does not fit the NN kernel (see udm_gemm_nn_ok)", (long)M, (long)N, (long)K);
  const long tiles = fm > 0 ? (M / (64 * fm)) * (N / 256) : 0;
  const long nkt = K / 64;
  long skl = tiles ? gemm_cus_available() / tiles : 1;
  if (skl > nkt / 4) skl = nkt / 4;
  if (skl > 32) skl = 32;
  const int sk = skl < 2 ? 1 : (int)skl;
  if (fm <= 0 || sk == 1 || !ws || ws_elems < (int64_t)sk * M * N || ldc % 4 != 0) return udm_gemm_nn_bf16(A, B, C, M, N, K, lda, ldb, ldc, stream);
  UDM_CHECK_ARG(lda >= K && ldb >= N && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "udm_gemm_nn_splitk_bf16: bad leading dimensions");
  UDM_CHECK_ARG(((uintptr_t)ws % 16 == 0) && ((uintptr_t)C % 8 == 0), "udm_gemm_nn_splitk_bf16: operand alignment");
  QuadArgs q{};
  q.A = (const bf16_t*)A; q.B = (const bf16_t*)B; q.C = ws; q.lda = lda; q.ldb = ldb; q.ldc = N;
  q.M = (int)M; q.N = (int)N; q.K = (int)K; q.beta = 0.f; q.splitk = sk; q.slice_stride = (long)M * N;
  if (int rc = udm_quad_launch_nn_f32(q, fm, stream)) return rc;
  const long n4 = (long)M * N / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(grid), dim3(256), 0, stream, (const float*)ws, (bf16_t*)C, (long)M, (int)N, (long)ldc, sk, (long)M * N);
  UDM_CHECK_LAUNCH("udm_gemm_nn_splitk_bf16(reduce)");
  return 0;
}

// Two weight gradients of one backward step in ONE launch: C0[M0, N] (+)= A0[K, M0]^T B0[K, N] and C1[M1, N] (+)= A1[K, M1]^T B1[K, N] (fp32, beta as
// udm_gemm_tn_bf16), same N and K, M0 / M1 / N multiples of 256, K a multiple of 64: the 256 x 256 one-wave-per-SIMD tiles of both problems share a grid, so
// that e.g. the qkv (6144 x 2048: 192 tiles) and the out-proj (2048 x 2048: 64 tiles) weight gradients of a DiT block fill the 256 CUs exactly once.
// Few tiles over a long contraction (UniDisc-S: 27 + 9 tiles, K = 24 576): K is split as in udm_gemm_tn_splitk_bf16 - as many slices as fill the CUs once -
// through the workspace `ws` (>= slices * (M0 + M1) * N floats), followed by one reduce pass per output; without a (large enough) workspace the launch is unsplit.
// C0 / C1 must be contiguous (ldc == N) when the split is taken.  Returns 3 (and does nothing) when the shapes do not qualify: the caller then issues the
// two plain calls.
extern "C" int udm_gemm_tn_pair_bf16(const void* A0, const void* B0, void* C0, int64_t M0, int64_t lda0, int64_t ldb0, int64_t ldc0, const void* A1,
                                     const void* B1, void* C1, int64_t M1, int64_t lda1, int64_t ldb1, int64_t ldc1, int64_t N, int64_t K, float beta,
                                     float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(A0 && B0 && C0 && A1 && B1 && C1, "udm_gemm_tn_pair_bf16: null operand");
  if (!udm_quad_mode() || M0 <= 0 || M1 <= 0 || M0 % 256 || M1 % 256 || N <= 0 || N % 256 || K < 128 || K % 64) return 3;
  if (lda0 % 8 || ldb0 % 8 || lda1 % 8 || ldb1 % 8 || ldc0 % 4 || ldc1 % 4 || lda0 < M0 || lda1 < M1 || ldb0 < N || ldb1 < N) return 3;
  if (((uintptr_t)A0 | (uintptr_t)B0 | (uintptr_t)C0 | (uintptr_t)A1 | (uintptr_t)B1 | (uintptr_t)C1) % 16) return 3;
  QuadArgs q{};
  q.A = (const bf16_t*)A0; q.B = (const bf16_t*)B0; q.C = C0; q.lda = lda0; q.ldb = ldb0; q.ldc = ldc0;
  q.A2 = (const bf16_t*)A1; q.B2 = (const bf16_t*)B1; q.C2 = C1; q.lda2 = lda1; q.ldb2 = ldb1; q.ldc2 = ldc1;
  q.M = (int)M0; q.N = (int)N; q.K = (int)K; q.beta = beta; q.splitk = 1;
  const long tiles = ((M0 + M1) / 256) * (N / 256), nkt = K / 64;
  long skl = 256 / tiles;
  if (skl > nkt / 8) skl = nkt / 8;
  if (skl > 32) skl = 32;
  const int sk = skl < 2 ? 1 : (int)skl;
  const long area = (long)(M0 + M1) * N;
  if (sk > 1 && ws && ws_elems >= (int64_t)sk * area && ldc0 == N && ldc1 == N && ((uintptr_t)ws % 16) == 0) {
    // slice s of problem one at ws + s * area, of problem two at ws + M0 * N + s * area (both with leading dimension N)
    q.C = ws; q.C2 = ws + (long)M0 * N; q.ldc = N; q.ldc2 = N; q.beta = 0.f; q.splitk = sk; q.slice_stride = area;
    if (int rc = udm_quad_launch_tn_pair(q, M1, stream)) return rc;
    const long n40 = (long)M0 * N / 4, n41 = (long)M1 * N / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((n40 + 255) / 256, 2048)), dim3(256), 0, stream, (const float*)ws, (float*)C0, n40, sk, area, beta);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((n41 + 255) / 256, 2048)), dim3(256), 0, stream, (const float*)(ws + (long)M0 * N), (float*)C1, n41, sk,
                       area, beta);
    UDM_CHECK_LAUNCH("udm_gemm_tn_pair_bf16(reduce)");
    return 0;
  }
  return udm_quad_launch_tn_pair(q, M1, stream);
}

namespace {
struct ReduceSegs { long start4[5]; float* out[4]; int n; };   // segment i = float4 indices [start4[i], start4[i + 1]) of the concatenated partial matrices
// out_i = beta * out_i + sum_s ws[s] over up to four contiguous outputs laid out back to back in every workspace slice
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(const float* __restrict__ ws, ReduceSegs sg, int S, long stride, float beta) {
  const long n4 = sg.start4[sg.n];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 acc = reinterpret_cast<const float4*>(ws)[i];
    for (int s = 1; s < S; ++s) {
      const float4 v = reinterpret_cast<const float4*>(ws + s * stride)[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    int k = 0;
    while (k + 1 < sg.n && i >= sg.start4[k + 1]) ++k;
    float4* o = reinterpret_cast<float4*>(sg.out[k]) + (i - sg.start4[k]);
    if (beta != 0.f) { const float4 c = *o; acc.x += beta * c.x; acc.y += beta * c.y; acc.z += beta * c.z; acc.w += beta * c.w; }
    *o = acc;
  }
}
}  // namespace

// Up to four weight gradients of one backward step over the SAME contraction (the rows of a small DiT block: UniDisc-S's qkv 2304 x 768, out-proj 768 x 768,
// mlp.0 3072 x 768 and mlp.2 768 x 3072 over K = B*L = 24 576) in ONE split-K launch + ONE reduce pass: C_i[M_i, N_i] (fp32, contiguous) = beta * C_i +
// A_i[K, M_i]^T B_i[K, N_i].  All tiles (256 x 256) of all problems share the grid and the K split (as many slices as fill the CUs once), so the partial-tile
// traffic and the launch count are those of one problem of the combined size (separately: three launches at 7 slices each + four reduce passes).
// M_i, N_i multiples of 256, K of 64, at most 128 tiles in total, ws >= slices * sum(M_i N_i) floats.  Returns 3 (and does nothing) when the shapes do not
// qualify: the caller then issues the plain calls.
extern "C" int udm_gemm_tn_multi_bf16(int nprob, const void* const* A, const void* const* B, void* const* C, const int64_t* M, const int64_t* N, const int64_t* lda,
                                      const int64_t* ldb, int64_t K, float beta, float* ws, int64_t ws_elems, hipStream_t stream) {
  UDM_CHECK_ARG(A && B && C && M && N && lda && ldb && nprob >= 1 && nprob <= 4, "udm_gemm_tn_multi_bf16: one to four problems");
  if (!udm_quad_mode() || K < 128 || K % 64 || !ws || ((uintptr_t)ws % 16)) return 3;
  QuadArgs q{};
  long tiles = 0, area = 0;
  for (int i = 0; i < nprob; ++i) {
    if (!A[i] || !B[i] || !C[i] || M[i] <= 0 || N[i] <= 0 || M[i] % 256 || N[i] % 256 || lda[i] % 8 || ldb[i] % 8 || lda[i] < M[i] || ldb[i] < N[i]) return 3;
    if (((uintptr_t)A[i] | (uintptr_t)B[i] | (uintptr_t)C[i]) % 16) return 3;
    q.m_tile_start[i] = (int)tiles;
    q.m_tiles_n[i] = (int)(N[i] / 256);
    q.mA[i] = (const bf16_t*)A[i]; q.mB[i] = (const bf16_t*)B[i];
    q.mC[i] = ws + area; q.mlda[i] = lda[i]; q.mldb[i] = ldb[i]; q.mldc[i] = N[i];   // every slice writes its partial tile; slice s at + s * total area
    tiles += (M[i] / 256) * (N[i] / 256);
    area += M[i] * N[i];
  }
  q.m_tile_start[nprob] = (int)tiles;
  q.nprob = nprob;
  const long nkt = K / 64;
  long skl = gemm_cus_available() / tiles;
  if (skl > nkt / 8) skl = nkt / 8;
  if (skl > 32) skl = 32;
  if (tiles > 128 || skl < 2 || ws_elems < skl * area) return 3;
  q.A = q.mA[0]; q.B = q.mB[0]; q.C = q.mC[0]; q.lda = q.mlda[0]; q.ldb = q.mldb[0]; q.ldc = q.mldc[0];
  q.K = (int)K; q.beta = 0.f; q.splitk = (int)skl; q.slice_stride = area;
  if (int rc = udm_quad_launch_tn_multi(q, stream)) return rc;
  ReduceSegs sg{};
  sg.n = nprob;
  long at = 0;
  for (int i = 0; i < nprob; ++i) { sg.start4[i] = at / 4; sg.out[i] = (float*)C[i]; at += M[i] * N[i]; }
  sg.start4[nprob] = at / 4;
  hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3((unsigned)std::min<long>((at / 4 + 255) / 256, 2048)), dim3(256), 0, stream, (const float*)ws, sg, (int)skl, area, beta);
  UDM_CHECK_LAUNCH("udm_gemm_tn_multi_bf16(reduce)");
  return 0;
}

int udm_gemm_cus_available() { return gemm_cus_available(); }
extern "C" int udm_gemm_set_cus(int cus) {   // 0 = all CUs; otherwise the persistent NT grid (a multiple of 8 in [8, 256])
  UDM_CHECK_ARG(cus == 0 || (cus >= 8 && cus <= 256 && cus % 8 == 0), "udm_gemm_set_cus: 0 or a multiple of 8 in [8, 256]");
  g_gemm_cus = cus;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int udm_gemm_set_quad(int mode) {   // diagnostics / tests: 0 = 8-wave kernels only, 1 = auto (default), 2 = quad wherever the shape fits
  UDM_CHECK_ARG(mode >= 0 && mode <= 2, "udm_gemm_set_quad: mode must be 0, 1 or 2");
  g_quad_mode = mode;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int udm_gemm_set_persist(int enable) {   // diagnostics / tests: 0 = one block per output tile everywhere
  g_gemm_persist = enable ? 1 : 0;
  return 0;
}
