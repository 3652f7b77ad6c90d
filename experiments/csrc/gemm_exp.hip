// EXPERIMENTAL tile/pipeline variants of the NT bf16 GEMM, selectable at run time for A/B measurement
// (scripts/bench_gemm_variants.py).  The production entry point stays udm_gemm_nt_bf16 (gemm.hip); the
// winning structure is folded in there.
//
// Variant = <BM, BN, WGM, WGN, STAGES>: block tile BM x BN x 64, WGM x WGN waves, each wave
// (BM/WGM) x (BN/WGN) as 32x32x16 MFMA fragments; operands are staged with global_load_lds_dwordx4
// (LDS-DMA, no VGPR round trip) into a STAGES-deep ring of XOR-swizzled tiles; waits are counted
// s_waitcnt vmcnt(N) + raw s_barrier so the next tiles' loads stay in flight across barriers.
#include "common.h"
#include "../../include/unidisc_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {
using namespace udm;
constexpr int BK = 64;

struct XArgs {
  const bf16_t* A;
  const bf16_t* B;
  bf16_t* C;
  long lda, ldb, ldc;
  int M, N, K, tiles_m, tiles_n;
  unsigned long long* tl;   // quad timeline (OPT bit 4): cycle stamps of block 0, [wave][K tile 8..15][tag]
  int sg_mod, sg_mul;       // ring: block pid starts its K loop at tile ((pid % sg_mod) * sg_mul) % nk and wraps (0: no stagger)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one 16-byte-per-lane LDS-DMA: lane i lands at lds_base + 16*i
__device__ __forceinline__ void glds16(const void* gptr, char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (UDM_LDS void*)lds_base, 16, 0, 0);
}

template <int BM, int BN, int WGM, int WGN, int STAGES>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_glds_kernel(XArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 32, FN = WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;  // 1 KiB (8 rows) per wave-instruction
  constexpr int NWAVES = WGM * WGN;
  constexpr int A_PW = A_INSTR / NWAVES, B_PW = B_INSTR / NWAVES;    // instructions per wave per stage
  constexpr int LOADS = A_PW + B_PW;
  static_assert(A_INSTR % NWAVES == 0 && B_INSTR % NWAVES == 0, "tile rows must split evenly over waves");

  const int nwg = p.tiles_m * p.tiles_n;
  int pid = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = pid / per_group, first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz, tn = (pid % per_group) / gsz;
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, hi = lane >> 5;

  // staging: wave w issues instruction j -> tile rows (w*PW + j)*8 .. +7; lane i -> row +i/8, LDS slot i%8,
  // global k-slot (i%8) ^ ((row>>1)&7)   (swizzle applied on the SOURCE address, LDS image stays lane-linear)
  const int lrow = lane >> 3, lslot = lane & 7;
  const bf16_t* a_src[A_PW];
  const bf16_t* b_src[B_PW];
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int r = (wave * A_PW + j) * 8 + lrow;
    const int gr = min(row0 + r, p.M - 1);
    a_src[j] = p.A + (long)gr * p.lda + ((lslot ^ ((r >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    const int r = (wave * B_PW + j) * 8 + lrow;
    const int gr = min(col0 + r, p.N - 1);
    b_src[j] = p.B + (long)gr * p.ldb + ((lslot ^ ((r >> 1) & 7)) << 3);
  }
  auto issue = [&](int kt, int stage) {
    char* As = smem + stage * STAGE_BYTES;
    char* Bs = As + A_BYTES;
    const int k = kt * BK;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) glds16(a_src[j] + k, As + (wave * A_PW + j) * 1024);
#pragma unroll
    for (int j = 0; j < B_PW; ++j) glds16(b_src[j] + k, Bs + (wave * B_PW + j) * 1024);
  };

  const int sw = (l31 >> 1) & 7;
  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const char* As = smem + stage * STAGE_BYTES;
    const char* Bs = As + A_BYTES;
    // software-pipelined fragment reads: kk+1's ds_read_b128 are issued before kk's MFMAs (two fragment sets live)
    bf16x8_t a[2][FM], b[2][FN];
    auto ld = [&](int kk, int s) {
      const int so = ((kk * 2 + hi) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < FM; ++i) a[s][i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * WM + i * 32 + l31) * 128 + so);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[s][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * WN + j * 32 + l31) * 128 + so);
    };
    ld(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      __builtin_amdgcn_sched_barrier(0);  // keep the k-steps apart: the scheduler otherwise chains dependent MFMAs and sinks the reads
      if (kk < 3) ld(kk + 1, (kk + 1) & 1);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
      if (kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);  // next k-step's fragment reads first ...
      __builtin_amdgcn_sched_group_barrier(0x008, FM * FN, 0);              // ... then this k-step's MFMAs cover their latency
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = p.K / BK;  // experimental kernels: K % 64 == 0
  if (STAGES == 2) {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      compute(kt & 1);
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) issue(kt + 2, kt & 1);
    }
  } else {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();  // tile kt visible to all waves; every wave has finished compute(kt-1)
      int nxt = st + 2; if (nxt >= STAGES) nxt -= STAGES;
      if (kt + 2 < nk) issue(kt + 2, nxt);
      compute(st);
      st = (st + 1 == STAGES) ? 0 : st + 1;
    }
  }
  __syncthreads();

  // epilogue: bf16 store straight from accumulators (row-strided 2-byte stores; fine for an A/B of the main loop)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int n = col0 + wn * WN + j * 32 + l31;
        if (m < p.M && n < p.N) p.C[(long)m * p.ldc + n] = f2bf(acc[i][j][r]);
      }
}


// ------------------------------------------------------------------------------------------------
// Staggered two-group schedule (2 waves per SIMD): every K tile is cut into phases
//     { fragment ds_reads (+ a slice of the next tile's LDS-DMA) ; s_barrier ; MFMAs ; s_barrier }
// and the second half of the waves runs ONE barrier behind the first half, so while one wave of a SIMD
// owns the matrix pipe its partner is in its read section.  KKPP = k-steps (of 16) per phase.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int KKPP, int OPT = 2>  // OPT bit0: LDS-DMA issued inside the MFMA section; bit1: s_setprio around MFMAs
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_nt_stagger_kernel(XArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 32, FN = WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int NWAVES = WGM * WGN;
  constexpr int A_PW = A_BYTES / 1024 / NWAVES, B_PW = B_BYTES / 1024 / NWAVES;
  constexpr int NPH = 4 / KKPP;                 // phases per K tile
  constexpr int LOADS = A_PW + B_PW;
  constexpr int ISSUE_PH = NPH > 2 ? NPH - 2 : 1;  // phases of a tile that carry the next tile's loads (the last ones stay free = slack)
  static_assert((BM * BK * 2 / 1024) % NWAVES == 0 && (BN * BK * 2 / 1024) % NWAVES == 0, "tile rows must split evenly over waves");

  const int nwg = p.tiles_m * p.tiles_n;
  int pid = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int grp = pid / per_group, first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz, tn = (pid % per_group) / gsz;
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int late = __builtin_amdgcn_readfirstlane(wave >= NWAVES / 2 ? 1 : 0);  // second half runs one barrier behind
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int lrow = lane >> 3, lslot = lane & 7;
  const bf16_t* src[LOADS];
  int dst[LOADS];
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
  const int sw = (l31 >> 1) & 7;
  // OPT bit 4: refills through buffer_load ... lds (buffer resource + 32-bit lane offset + SGPR k offset): no 64-bit vector address per piece
#if defined(__HIP_DEVICE_COMPILE__)   // (the buffer builtins do not exist in the host pass)
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((long)p.M * p.lda) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(((long)p.N * p.ldb) * 2), 0x00020000);
#endif
  int voff[LOADS];
#pragma unroll
  for (int j = 0; j < LOADS; ++j) voff[j] = (int)((reinterpret_cast<const char*>(src[j]) - reinterpret_cast<const char*>(j < A_PW ? p.A : p.B)));
  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
#pragma unroll
  for (int j = 0; j < LOADS; ++j) glds16(src[j], smem + dst[j]);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();
  uint4 stg[LOADS];
  unsigned long long tl[24];   // OPT bit 2: cycle stamps (kept in SGPRs, stored after the loop): K tiles 8, 9 x 4 phases x {reads done, barrier released, MFMAs done}
#pragma unroll
  for (int q = 0; q < 24; ++q) tl[q] = 0;

  for (int kt = 0; kt < nk; ++kt) {
    const char* As = smem + (kt & 1) * STAGE_BYTES;
    const char* Bs = As + A_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    const bool more = kt + 1 < nk;
    const int knext = (kt + 1) * BK;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      bf16x8_t a[KKPP][FM], b[KKPP][FN];
#pragma unroll
      for (int q = 0; q < KKPP; ++q) {
        const int so = (((ph * KKPP + q) * 2 + hi) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < FM; ++i) a[q][i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * WM + i * 32 + l31) * 128 + so);
#pragma unroll
        for (int j = 0; j < FN; ++j) b[q][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * WN + j * 32 + l31) * 128 + so);
      }
      constexpr int PER = (LOADS + ISSUE_PH - 1) / ISSUE_PH;
      if ((OPT & 8) && ph == NPH - 1 && more) {   // register-staged refill: the tile fetched during phase 0 goes to LDS now
        wait_vmcnt<0>();
#pragma unroll
        for (int j = 0; j < LOADS; ++j) *reinterpret_cast<uint4*>(nxt + dst[j] + lane * 16) = stg[j];
      }
      if (!(OPT & 9) && ph < ISSUE_PH && more) {
#pragma unroll
        for (int j = ph * PER; j < (ph + 1) * PER && j < LOADS; ++j) glds16(src[j] + knext, nxt + dst[j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (ph == NPH - 1) wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
      if ((OPT & 4) && NPH == 4) { if (kt == 8) tl[ph * 3] = __builtin_amdgcn_s_memtime(); if (kt == 9) tl[12 + ph * 3] = __builtin_amdgcn_s_memtime(); }
      __builtin_amdgcn_s_barrier();
      if ((OPT & 4) && NPH == 4) { if (kt == 8) tl[ph * 3 + 1] = __builtin_amdgcn_s_memtime(); if (kt == 9) tl[12 + ph * 3 + 1] = __builtin_amdgcn_s_memtime(); }
      __builtin_amdgcn_sched_barrier(0);
      if (OPT & 2) __builtin_amdgcn_s_setprio(1);
      if ((OPT & 1) && !(OPT & 8) && ph < ISSUE_PH && more) {
#pragma unroll
        for (int j = ph * PER; j < (ph + 1) * PER && j < LOADS; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
          if (OPT & 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(j < A_PW ? rsA : rsB, (UDM_LDS void*)(nxt + dst[j]), 16, voff[j], knext * 2, 0, 0);
          else
#endif
            glds16(src[j] + knext, nxt + dst[j]);
        }
      }
      if ((OPT & 8) && ph == 0 && more) {   // OPT bit 3: plain global loads into registers instead of LDS-DMA (A/B of the issue cost)
#pragma unroll
        for (int j = 0; j < LOADS; ++j) stg[j] = *reinterpret_cast<const uint4*>(src[j] + knext);
      }
#pragma unroll
      for (int q = 0; q < KKPP; ++q)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
      if ((OPT & 1) && ph < ISSUE_PH) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      if (OPT & 2) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if ((OPT & 4) && NPH == 4) { if (kt == 8) tl[ph * 3 + 2] = __builtin_amdgcn_s_memtime(); if (kt == 9) tl[12 + ph * 3 + 2] = __builtin_amdgcn_s_memtime(); }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();
  if ((OPT & 4) && p.tl && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int q = 0; q < 24; ++q) p.tl[wave * 24 + q] = tl[q];
  }

#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int n = col0 + wn * WN + j * 32 + l31;
        if (m < p.M && n < p.N) p.C[(long)m * p.ldc + n] = f2bf(acc[i][j][r]);
      }
}

template <int BM, int BN, int WGM, int WGN, int KKPP, int OPT = 2>
int launch_stagger(const XArgs& a0, hipStream_t stream) {
  XArgs a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * BK * 2;
  if (OPT & 4) {
    static int calls = 0;
    if (++calls == 3) {
      auto kern2 = gemm_nt_stagger_kernel<BM, BN, WGM, WGN, KKPP, OPT>;
      (void)hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      unsigned long long* buf = nullptr;
      (void)hipMalloc(&buf, 8 * 24 * 8);
      (void)hipMemset(buf, 0, 8 * 24 * 8);
      a.tl = buf;
      hipLaunchKernelGGL(kern2, dim3(a.tiles_m * a.tiles_n), dim3(64 * WGM * WGN), lds, stream, a);
      (void)hipStreamSynchronize(stream);
      static unsigned long long h[8 * 24];
      (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w) {
        fprintf(stderr, "STL wave %d:", w);
        for (int q = 0; q < 24; ++q) fprintf(stderr, " %lld", (long long)(h[w * 24 + q] - h[0]));
        fprintf(stderr, "\n");
      }
      (void)hipFree(buf);
      return 0;
    }
  }
  auto kern = gemm_nt_stagger_kernel<BM, BN, WGM, WGN, KKPP, OPT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(64 * WGM * WGN), lds, stream, a);
  UDM_CHECK_LAUNCH("udm_gemm_nt_bf16_variant(stagger)");
  return 0;
}

// ---- "quad": 256 x 256 x 64 tile, FOUR waves (2 x 2), 128 x 128 per wave (16 accumulators = 256 AGPRs), ONE wave per SIMD ----
// A 128 x 128 wave tile needs (128 + 128) fragment rows per 16 MFMAs = 512 B of LDS reads per MFMA, against 717-768 B for the
// 160 x 64 / 128 x 64 wave tiles of the 8-wave kernels.  With one wave per SIMD nothing but the wave's own stream hides latency, so
// the loop is software-pipelined by hand: fragments of k-step s+1 are read (into the other register buffer) under the MFMAs of
// k-step s; the single barrier of a K tile sits BEFORE the last k-step's MFMAs (its fragments are already in registers), so the
// barrier wait, the first reads of the next tile and the LDS-DMA refill of the stage just retired all run under those 16 MFMAs.
// Requires M, N multiples of 256 (the production dispatcher keeps the 8-wave kernels for ragged shapes).
// OPT bit 0: interleave hints (sched_group_barrier) inside a k-step.  Ablations (wrong results, timing only): bit 1 drops the
// fragment reads of the loop, bit 2 drops the LDS-DMA refills, bit 3 drops the barrier.
template <int OPT>
__global__ __launch_bounds__(256, 1) void gemm_nt_quad_kernel(XArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, A_BYTES = BM * BK * 2, STAGE_BYTES = 2 * A_BYTES, PW = 8;  // PW: 1 KiB pieces per wave per operand
  const int nwg = p.tiles_m * p.tiles_n;
  int pid = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int grp = pid / per_group, first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz, tn = (pid % per_group) / gsz;
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5, lrow = lane >> 3, lslot = lane & 7;

  // LDS-DMA sources: wave w stages tile rows [64 w, 64 w + 64) of A and of B, 8 rows per piece.  Piece j = 2 q + e covers rows
  // 64 w + 16 q + 8 e + lrow; the swizzle term ((row >> 1) & 7) only depends on e, so two 32-bit lane offsets per operand suffice and
  // everything else is a wave-uniform base (SALU).
  uint32_t offa[2], offb[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int r = wave * 64 + 8 * e + lrow;
    offa[e] = (uint32_t)((r * p.lda + ((lslot ^ ((r >> 1) & 7)) << 3)) * 2);
    offb[e] = (uint32_t)((r * p.ldb + ((lslot ^ ((r >> 1) & 7)) << 3)) * 2);
  }
  const char* abase = reinterpret_cast<const char*>(p.A + (long)row0 * p.lda);
  const char* bbase = reinterpret_cast<const char*>(p.B + (long)col0 * p.ldb);
  const long qa = 16 * p.lda * 2, qb = 16 * p.ldb * 2;
  auto ubase = [&](const char* ptr) {   // pin a wave-uniform pointer into SGPRs so the DMA takes the (SGPR base + 32-bit VGPR offset) form
    const uint64_t u = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi32 << 32) | lo);
  };
  auto dma_piece = [&](int kt, int stage, int j) {   // j in 0..15: 0..7 A pieces, 8..15 B pieces
    const int jj = j & 7, q = jj >> 1, e = jj & 1;
    char* dst = smem + stage * STAGE_BYTES + (j >> 3) * A_BYTES + (wave * PW + jj) * 1024;
    if (j < 8) glds16(ubase(abase + (long)kt * (BK * 2) + q * qa) + (size_t)offa[e], dst);
    else glds16(ubase(bbase + (long)kt * (BK * 2) + q * qb) + (size_t)offb[e], dst);
  };

  // fragment addresses: row (wm 128 + 32 i + l31), 16-byte granule ((2 kk + hi) ^ sw); i, the B region and the stage are immediates
  const int sw = (l31 >> 1) & 7;
  const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;
  uint32_t fa_addr[4], fb_addr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const uint32_t so = ((kk * 2 + hi) ^ sw) << 4;
    fa_addr[kk] = lds0 + (wm * 128 + l31) * 128 + so;
    fb_addr[kk] = lds0 + (wn * 128 + l31) * 128 + so;
  }
  bf16x8_t fa[2][4], fb[2][4];
  auto read_frags = [&](int stage, int kk, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[buf][i] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fa_addr[kk] + stage * STAGE_BYTES + i * 4096));
      fb[buf][i] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fb_addr[kk] + stage * STAGE_BYTES + A_BYTES + i * 4096));
    }
  };
  // Accumulators are zeroed BY the matrix pipe (MFMA of an opaque zero fragment onto the constant 0): 256 v_mov + copies into the
  // AGPR file would put 256 live VGPRs at this point and make the allocator spill the loop's addresses and fragments.
  f32x16_t acc[4][4];
  {
    bf16x8_t zf = {};
    asm volatile("" : "+v"(zf));
    const f32x16_t zc = {};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, zf, zc, 0, 0, 0);
  }
  auto read_one = [&](int stage, int kk, int buf, int n) {   // n in 0..7: A fragment n (n < 4) or B fragment n - 4
    if (n < 4) fa[buf][n] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fa_addr[kk] + stage * STAGE_BYTES + n * 4096));
    else fb[buf][n - 4] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fb_addr[kk] + stage * STAGE_BYTES + A_BYTES + (n - 4) * 4096));
  };

  const int nk = p.K / BK;
#pragma unroll
  for (int j = 0; j < 16; ++j) dma_piece(0, 0, j);
  if (nk > 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) dma_piece(1, 1, j);
    wait_vmcnt<16>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, 0);

  // One K tile.  The issue order is written out and pinned (sched_barrier after every MFMA pair): per pair one fragment read of the
  // next k-step and, in the last k-step, two LDS-DMA pieces of the tile after next -- an in-order wave can only hide those issue
  // slots in the shadow of an MFMA that is already executing, never in a block of their own.
  auto tile_body = [&](int kt, auto stage_c) {
    // stage_c: 0 / 1 = steady-state tile (stage is a compile-time constant, a next tile and a refill always exist: no branches);
    // -1 = one of the last three tiles (everything derived from kt at run time)
    constexpr int STC = decltype(stage_c)::value;
    const int ST = STC >= 0 ? STC : (kt & 1);
    const bool HAS_NEXT = STC >= 0 || kt + 1 < nk, HAS_DMA = STC >= 0 || kt + 2 < nk;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int buf = kk & 1;
      if (kk == 3) {
        // boundary: k-step 3's fragments are in registers; everything of this stage has been read
        __builtin_amdgcn_sched_barrier(0);
        if ((OPT & 16) && p.tl && blockIdx.x == 0 && kt >= 8 && kt < 16) {
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) p.tl[(wave * 8 + kt - 8) * 4 + 0] = t;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (OPT & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if ((OPT & 16) && p.tl && blockIdx.x == 0 && kt >= 8 && kt < 16) {
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long t = __builtin_amdgcn_s_memtime();
          if (lane == 0) p.tl[(wave * 8 + kt - 8) * 4 + 1] = t;
        }
      }
#pragma unroll
      for (int n = 0; n < 8; ++n) {   // MFMA pair n: accumulators (i, j) = (n >> 1, 2 (n & 1)) and (n >> 1, 2 (n & 1) + 1)
        __builtin_amdgcn_sched_barrier(0);
        const int i = n >> 1, j0 = 2 * (n & 1);
        acc[i][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i], fb[buf][j0], acc[i][j0], 0, 0, 0);
        if (kk == 3 && HAS_DMA && !(OPT & 4)) dma_piece(kt + 2, ST, 2 * n);
        acc[i][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i], fb[buf][j0 + 1], acc[i][j0 + 1], 0, 0, 0);
        if (kk == 3 && HAS_DMA && !(OPT & 4)) dma_piece(kt + 2, ST, 2 * n + 1);
        if (!(OPT & 2)) {   // (after the MFMAs: the waitcnt the compiler places before the first MFMA behind the barrier asm must not cover a new read)
          if (kk < 3) read_one(ST, kk + 1, buf ^ 1, n);
          else if (HAS_NEXT) read_one(ST ^ 1, 0, 0, n);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if ((OPT & 16) && p.tl && blockIdx.x == 0 && kt >= 8 && kt < 16) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (lane == 0) p.tl[(wave * 8 + kt - 8) * 4 + 2] = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int kt = 0;
  for (; kt + 3 < nk; kt += 2) {
    tile_body(kt, std::integral_constant<int, 0>{});
    tile_body(kt + 1, std::integral_constant<int, 1>{});
  }
  for (; kt < nk; ++kt) tile_body(kt, std::integral_constant<int, -1>{});
  __syncthreads();  // all LDS tile reads are done: the wave-private epilogue patches may overwrite stage memory

  // epilogue: each 32 x 32 accumulator block goes through a wave-private 4 KiB LDS patch so a lane owns 4 consecutive columns of a row
  float* patch0 = reinterpret_cast<float*>(smem) + wave * 2048;
  const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma clang loop unroll(full)
  for (int i = 0; i < 4; ++i)
#pragma clang loop unroll(full)
    for (int j = 0; j < 4; ++j) {
      float* patch = patch0 + ((i * 4 + j) & 1) * 1024;
#pragma clang loop unroll(full)
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[i][j][r];
      const int gn = col0 + wn * 128 + j * 32 + ec;
      const int gm0 = row0 + wm * 128 + i * 32 + er;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(patch + (q * 8 + er) * 32 + ec);
        *reinterpret_cast<uint2*>(p.C + (long)(gm0 + q * 8) * p.ldc + gn) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
      }
    }
}

template <int OPT>
int launch_quad(const XArgs& a0, hipStream_t stream) {
  XArgs a = a0;
  UDM_CHECK_ARG(a.M % 256 == 0 && a.N % 256 == 0, "udm_gemm_nt_bf16_variant(quad): M, N must be multiples of 256");
  a.tiles_m = a.M / 256;
  a.tiles_n = a.N / 256;
  const size_t lds = 2 * 2 * 256 * BK * 2;
  auto kern = gemm_nt_quad_kernel<OPT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (OPT & 16) {
    static int calls = 0;
    if (++calls == 3) {
      unsigned long long* buf = nullptr;
      (void)hipMalloc(&buf, 4 * 8 * 4 * 8);
      (void)hipMemset(buf, 0, 4 * 8 * 4 * 8);
      a.tl = buf;
      hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, stream, a);
      (void)hipStreamSynchronize(stream);
      static unsigned long long h[4 * 8 * 4];
      (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
      for (int w = 0; w < 4; ++w)
        for (int t = 0; t < 8; ++t)
          fprintf(stderr, "QTL wave %d kt %d: wait@%lld released@%lld end@%lld\n", w, t + 8, (long long)(h[(w * 8 + t) * 4] - h[0]), (long long)(h[(w * 8 + t) * 4 + 1] - h[0]),
                  (long long)(h[(w * 8 + t) * 4 + 2] - h[0]));
      (void)hipFree(buf);
      return 0;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, stream, a);
  UDM_CHECK_LAUNCH("udm_gemm_nt_bf16_variant(quad)");
  return 0;
}

// ---- "ring": the quad tile (256 x 256, four waves, 128 x 128 per wave) over a FOUR-stage ring of 32-wide K tiles --------------
// What the quad timeline showed (variant 37): the 16 LDS-DMA pieces of a K tile, issued back to back under 16 MFMAs, stretch that
// section from 512 to ~1000 cycles (four waves x 1 KiB per 32 cycles = 128 B/clk against the CU's 64 B/clk vector-memory path), and
// the vmcnt(0) + barrier before it waits ~420 cycles (prefetch distance of one 2048-cycle tile).  Here a stage is 32 KiB (256 rows x
// 64 B per operand), so four stages fit: a stage is released by the ONE barrier of its tile (after the last fragment read of the
// tile), the refill pieces go out one per four MFMAs (32 B/clk per CU) and land two to three tiles (2-3 k cycles) before they are read,
// and the wait is a counted vmcnt(16) - the two younger tiles stay in flight.
// LDS tile: row r at r * 64 B, 16-byte granule g stored at slot g ^ ((r >> 2) & 3) (16 consecutive lanes of a ds_read_b128 cover all
// 64 banks).  A piece is 16 rows; the swizzle is applied on the global-source side of the DMA.
template <int OPT>
__global__ __launch_bounds__(256, 1) void gemm_nt_ring_kernel(XArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKR = 32, OP_BYTES = 256 * BKR * 2, STAGE_BYTES = 2 * OP_BYTES;
  const int nwg = p.tiles_m * p.tiles_n;
  int pid = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int grp = pid / per_group, first_m = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tm = first_m + (pid % per_group) % gsz, tn = (pid % per_group) / gsz;
  const int row0 = tm * 256, col0 = tn * 256;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  // LDS-DMA sources: wave w stages tile rows [64 w, 64 w + 64) of A and of B, 16 rows per piece; lane l lands at row l >> 2, slot l & 3
  const int prow = lane >> 2, pslot = lane & 3;
  const uint32_t offa = (uint32_t)((prow * p.lda + ((pslot ^ ((lane >> 4) & 3)) << 3)) * 2);
  const uint32_t offb = (uint32_t)((prow * p.ldb + ((pslot ^ ((lane >> 4) & 3)) << 3)) * 2);
  const char* abase = reinterpret_cast<const char*>(p.A + (long)(row0 + wave * 64) * p.lda);
  const char* bbase = reinterpret_cast<const char*>(p.B + (long)(col0 + wave * 64) * p.ldb);
  const long qa = 16 * p.lda * 2, qb = 16 * p.ldb * 2;
  auto ubase = [&](const char* ptr) {
    const uint64_t u = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi32 << 32) | lo);
  };
#if defined(__HIP_DEVICE_COMPILE__)
  // OPT bit 5: buffer_load ... lds with a wave-uniform descriptor per operand: the per-piece / per-tile part of the address is an SGPR
  // offset (SALU), the per-lane part one constant VGPR - no vector address arithmetic in the loop at all
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)ubase(abase), 0, 0x7ffff000, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)ubase(bbase), 0, 0x7ffff000, 0x00020000);
#endif
  const int qa32 = (int)qa, qb32 = (int)qb;
  const int nk = p.K / BKR;
  // K stagger: blocks that share an operand panel would otherwise request every line at the same moment (they run in lock step) and all
  // of them wait out the one L2 miss; started at rotated K offsets, the first block pulls a line into L2 and the others hit it there.
  const int koff = p.sg_mod > 0 ? ((pid % p.sg_mod) * p.sg_mul) % nk : 0;
  auto dma_piece = [&](int tl_, int stage, int j) {   // j in 0..7: 0..3 A pieces, 4..7 B pieces
    int t = tl_ + koff;
    if (t >= nk) t -= nk;
    const int jj = j & 3;
    char* dst = smem + stage * STAGE_BYTES + (j >> 2) * OP_BYTES + (wave * 4 + jj) * 1024;
    if ((OPT & 64) && wave != 0) return;   // ablation: only wave 0 refills (timing only)
#if defined(__HIP_DEVICE_COMPILE__)
    if (OPT & 32) {
      if (j < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (UDM_LDS void*)dst, 16, offa, t * (BKR * 2) + jj * qa32, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (UDM_LDS void*)dst, 16, offb, t * (BKR * 2) + jj * qb32, 0, 0);
      return;
    }
#endif
    if (j < 4) glds16(ubase(abase + (long)t * (BKR * 2) + jj * qa) + (size_t)offa, dst);
    else glds16(ubase(bbase + (long)t * (BKR * 2) + jj * qb) + (size_t)offb, dst);
  };

  const int sw = (l31 >> 2) & 3;
  const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;
  uint32_t fa_addr[2], fb_addr[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint32_t so = ((kk * 2 + hi) ^ sw) << 4;
    fa_addr[kk] = lds0 + (wm * 128 + l31) * 64 + so;
    fb_addr[kk] = lds0 + OP_BYTES + (wn * 128 + l31) * 64 + so;
  }
  bf16x8_t fa[2][4], fb[2][4];
  auto read_one = [&](int stage, int kk, int buf, int n) {   // n in 0..7: A fragment n (n < 4) or B fragment n - 4
    if (n < 4) fa[buf][n] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fa_addr[kk] + stage * STAGE_BYTES + n * 2048));
    else fb[buf][n - 4] = *reinterpret_cast<UDM_LDS const bf16x8_t*>((size_t)(fb_addr[kk] + stage * STAGE_BYTES + (n - 4) * 2048));
  };
  f32x16_t acc[4][4];
  {
    bf16x8_t zf = {};
    asm volatile("" : "+v"(zf));
    const f32x16_t zc = {};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, zf, zc, 0, 0, 0);
  }

  // prologue: tiles 0..2 whole, pieces 0..3 of tile 3 (steady state: k-step 0 of tile t issues pieces 4..7 of tile t + 3, k-step 1 pieces 0..3 of t + 4)
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < nk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dma_piece(t, t, j);
    }
  if (3 < nk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_piece(3, 3, j);
  }
  if (nk >= 4) wait_vmcnt<20>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int n = 0; n < 8; ++n) read_one(0, 0, 0, n);

  auto tile_body = [&](int t, auto stage_c) {
    // stage_c 0..3: steady-state tile (t + 4 < nk: every refill exists, stage is a constant); -1: one of the last tiles (run-time checks)
    constexpr int STC = decltype(stage_c)::value;
    const int ST = STC >= 0 ? STC : (t & 3);
    const bool HAS_NEXT = STC >= 0 || t + 1 < nk, DMA3 = STC >= 0 || t + 3 < nk, DMA4 = STC >= 0 || t + 4 < nk;
#pragma unroll
    for (int n = 0; n < 8; ++n) {   // k-step 0: fragments of k-step 1 come in, pieces 4..7 of tile t + 3 go out
      __builtin_amdgcn_sched_barrier(0);
      const int i = n >> 1, j0 = 2 * (n & 1);
      acc[i][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j0], acc[i][j0], 0, 0, 0);
      if (!(OPT & 2) && n < 4) read_one(ST, 1, 1, 2 * n);
      if (n >= 4 && DMA3 && !(OPT & 4)) dma_piece(t + 3, (ST + 3) & 3, n);
      acc[i][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j0 + 1], acc[i][j0 + 1], 0, 0, 0);
      if (!(OPT & 2) && n < 4) read_one(ST, 1, 1, 2 * n + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if ((OPT & 16) && p.tl && blockIdx.x == 0 && t >= 16 && t < 24) {
      const unsigned long long ts = __builtin_amdgcn_s_memtime();
      if (lane == 0) p.tl[(wave * 8 + t - 16) * 4 + 0] = ts;
      __builtin_amdgcn_sched_barrier(0);
    }
    // tile t is read completely (lgkmcnt(0)); tile t + 1 has landed once at most the 16 younger pieces are outstanding
    if (STC >= 0) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if ((OPT & 16) && p.tl && blockIdx.x == 0 && t >= 16 && t < 24) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long ts = __builtin_amdgcn_s_memtime();
      if (lane == 0) p.tl[(wave * 8 + t - 16) * 4 + 1] = ts;
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {   // k-step 1: first fragments of tile t + 1 come in, pieces 0..3 of tile t + 4 go into the stage just released
      __builtin_amdgcn_sched_barrier(0);
      const int i = n >> 1, j0 = 2 * (n & 1);
      acc[i][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j0], acc[i][j0], 0, 0, 0);
      if (!(OPT & 2) && HAS_NEXT && n < 4) read_one((ST + 1) & 3, 0, 0, 2 * n);
      if (n >= 4 && DMA4 && !(OPT & 4)) dma_piece(t + 4, ST, n - 4);
      acc[i][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j0 + 1], acc[i][j0 + 1], 0, 0, 0);
      if (!(OPT & 2) && HAS_NEXT && n < 4) read_one((ST + 1) & 3, 0, 0, 2 * n + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if ((OPT & 16) && p.tl && blockIdx.x == 0 && t >= 16 && t < 24) {
      const unsigned long long ts = __builtin_amdgcn_s_memtime();
      if (lane == 0) p.tl[(wave * 8 + t - 16) * 4 + 2] = ts;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int t = 0;
  for (; t + 7 < nk; t += 4) {
    tile_body(t, std::integral_constant<int, 0>{});
    tile_body(t + 1, std::integral_constant<int, 1>{});
    tile_body(t + 2, std::integral_constant<int, 2>{});
    tile_body(t + 3, std::integral_constant<int, 3>{});
  }
  for (; t < nk; ++t) tile_body(t, std::integral_constant<int, -1>{});
  __syncthreads();

  float* patch0 = reinterpret_cast<float*>(smem) + wave * 2048;
  const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma clang loop unroll(full)
  for (int i = 0; i < 4; ++i)
#pragma clang loop unroll(full)
    for (int j = 0; j < 4; ++j) {
      float* patch = patch0 + ((i * 4 + j) & 1) * 1024;
#pragma clang loop unroll(full)
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[i][j][r];
      const int gn = col0 + wn * 128 + j * 32 + ec;
      const int gm0 = row0 + wm * 128 + i * 32 + er;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(patch + (q * 8 + er) * 32 + ec);
        *reinterpret_cast<uint2*>(p.C + (long)(gm0 + q * 8) * p.ldc + gn) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
      }
    }
}

template <int OPT>
int launch_ring(const XArgs& a0, hipStream_t stream) {
  XArgs a = a0;
  UDM_CHECK_ARG(a.M % 256 == 0 && a.N % 256 == 0 && a.K % 32 == 0, "udm_gemm_nt_bf16_variant(ring): M, N must be multiples of 256");
  a.tiles_m = a.M / 256;
  a.tiles_n = a.N / 256;
  a.sg_mod = a.sg_mul = 0;
  if (const char* e = getenv("UDM_RING_STAGGER")) sscanf(e, "%d,%d", &a.sg_mod, &a.sg_mul);
  const size_t lds = 4 * 2 * 256 * 32 * 2;
  auto kern = gemm_nt_ring_kernel<OPT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (OPT & 16) {
    static int calls = 0;
    if (++calls == 3) {
      unsigned long long* buf = nullptr;
      (void)hipMalloc(&buf, 4 * 8 * 4 * 8);
      (void)hipMemset(buf, 0, 4 * 8 * 4 * 8);
      a.tl = buf;
      hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, stream, a);
      (void)hipStreamSynchronize(stream);
      static unsigned long long h[4 * 8 * 4];
      (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
      for (int w = 0; w < 4; ++w)
        for (int t = 0; t < 8; ++t)
          fprintf(stderr, "RTL wave %d t %d: wait@%lld released@%lld end@%lld\n", w, t + 16, (long long)(h[(w * 8 + t) * 4] - h[0]), (long long)(h[(w * 8 + t) * 4 + 1] - h[0]),
                  (long long)(h[(w * 8 + t) * 4 + 2] - h[0]));
      (void)hipFree(buf);
      return 0;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, stream, a);
  UDM_CHECK_LAUNCH("udm_gemm_nt_bf16_variant(ring)");
  return 0;
}

template <int BM, int BN, int WGM, int WGN, int STAGES>
int launch(const XArgs& a0, hipStream_t stream) {
  XArgs a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  auto kern = gemm_nt_glds_kernel<BM, BN, WGM, WGN, STAGES>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(64 * WGM * WGN), lds, stream, a);
  UDM_CHECK_LAUNCH("udm_gemm_nt_bf16_variant");
  return 0;
}
}  // namespace

// variant ids: see scripts/bench_gemm_variants.py
extern "C" int udm_gemm_nt_bf16_variant(int variant, const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                        int64_t ldc, hipStream_t stream) {
  UDM_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && K % (variant >= 50 && variant < 60 ? 32 : 64) == 0, "udm_gemm_nt_bf16_variant: bad arguments (K %% 64 == 0 required)");
  XArgs a{(const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, (long)lda, (long)ldb, (long)ldc, (int)M, (int)N, (int)K, 0, 0, nullptr, 0, 0};
  switch (variant) {
    case 0: return launch<128, 128, 2, 2, 2>(a, stream);
    case 1: return launch<128, 128, 2, 2, 3>(a, stream);
    case 2: return launch<128, 128, 2, 2, 4>(a, stream);
    case 3: return launch<256, 128, 4, 2, 2>(a, stream);
    case 4: return launch<256, 128, 4, 2, 3>(a, stream);
    case 5: return launch<256, 128, 2, 2, 3>(a, stream);
    case 6: return launch<256, 256, 2, 4, 2>(a, stream);
    case 7: return launch<256, 256, 4, 4, 2>(a, stream);
    case 8: return launch<256, 256, 4, 2, 2>(a, stream);
    case 9: return launch<128, 256, 2, 4, 3>(a, stream);
    case 10: return launch_stagger<256, 256, 2, 4, 1>(a, stream);
    case 11: return launch_stagger<256, 256, 2, 4, 2>(a, stream);
    case 12: return launch_stagger<256, 128, 4, 2, 2>(a, stream);
    case 13: return launch_stagger<128, 256, 2, 4, 2>(a, stream);
    case 14: return launch_stagger<256, 256, 4, 2, 1>(a, stream);
    case 15: return launch_stagger<128, 128, 2, 4, 2>(a, stream);
    case 16: return launch_stagger<320, 256, 2, 4, 1>(a, stream);
    case 17: return launch_stagger<256, 320, 2, 4, 1>(a, stream);
    case 20: return launch_stagger<320, 256, 2, 4, 1, 1>(a, stream);
    case 21: return launch_stagger<320, 256, 2, 4, 1, 3>(a, stream);
    case 22: return launch_stagger<320, 256, 2, 4, 2, 1>(a, stream);
    case 23: return launch_stagger<320, 256, 2, 4, 2, 3>(a, stream);
    case 24: return launch_stagger<320, 256, 2, 4, 1, 0>(a, stream);
    case 30: return launch_quad<0>(a, stream);
    case 31: return launch_quad<1>(a, stream);
    case 32: return launch_quad<3>(a, stream);    // no fragment reads
    case 33: return launch_quad<5>(a, stream);    // no refills
    case 34: return launch_quad<7>(a, stream);    // neither
    case 35: return launch_quad<15>(a, stream);   // neither, no barrier: bare MFMA stream
    case 36: return launch_quad<9>(a, stream);    // no barrier only
    case 37: return launch_quad<17>(a, stream);   // timeline
    case 50: return launch_ring<0>(a, stream);
    case 52: return launch_ring<2>(a, stream);    // no fragment reads
    case 53: return launch_ring<4>(a, stream);    // no refills
    case 57: return launch_ring<16>(a, stream);   // timeline
    case 54: return launch_ring<32>(a, stream);   // buffer_load ... lds refills
    case 58: return launch_ring<48>(a, stream);   // buffer_load ... lds refills, timeline
    case 59: return launch_ring<48 + 64>(a, stream);   // timeline, only wave 0 refills (ablation)
    case 38: return launch_stagger<320, 256, 2, 4, 1, 5>(a, stream);   // production-like schedule with cycle stamps
    case 39: return launch_stagger<256, 256, 2, 4, 1, 1>(a, stream);   // 256-row tile, LDS-DMA inside the MFMA section
    case 40: return launch_stagger<256, 256, 2, 4, 1, 8>(a, stream);   // 256-row tile, register-staged refill (global_load + ds_write_b128)
    case 41: return launch_stagger<320, 256, 2, 4, 1, 8>(a, stream);   // 320-row tile, register-staged refill
    case 42: return launch_stagger<320, 256, 2, 4, 1, 1>(a, stream);   // 320-row tile, LDS-DMA by global_load_lds (production form)
    case 43: return launch_stagger<320, 256, 2, 4, 1, 17>(a, stream);  // 320-row tile, LDS-DMA by buffer_load ... lds
    default: udm_set_error("udm_gemm_nt_bf16_variant: unknown variant %d", variant); return 2;
  }
}
