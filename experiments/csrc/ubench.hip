// EXPERIMENTAL issue-cost microbenchmark (scripts/ubench_issue.py): what does one LDS-DMA / LDS read / global load cost an in-order wave
// that is otherwise streaming independent MFMAs?  One wave per SIMD (4 waves per block), a "k-step" of 16 independent 32x32x16 MFMAs
// (or 32 16x16x32 MFMAs: the same 512 matrix-pipe cycles) with NOPS memory instructions spread evenly through it, issue order pinned.
// Reports shader cycles per k-step (s_memtime) for wave 0 of block 0.  Not part of the product path.
#include "common.h"
#include "../../include/unidisc_hip.h"

namespace {
using namespace udm;

struct UArgs {
  const char* src;                 // >= 64 MiB of readable memory
  unsigned long long* out;         // [4] cycles of waves 0..3 of block 0
  float* sink;
  int iters;
  int stride;                      // bytes the source advances per memory instruction (0: always the same lines)
  int win_mask;                    // the per-wave source offset wraps at win_mask + 1 bytes
  int block_stride;                // bytes between the source windows of consecutive blocks (0: all blocks read the same lines)
  int wave_stride;                 // bytes between the source windows of the four waves
};

// KIND: 0 none, 1 global_load_lds b128, 2 buffer_load lds b128, 3 buffer_load lds b32, 4 global_load_dwordx4 to VGPRs, 5 ds_read_b128,
//       6 s_nop filler (16 cycles), 7 buffer_load lds b128 issued by wave 0 only
template <int MFMA16, int KIND, int NOPS>
__global__ __launch_bounds__(256, 1) void ubench_kernel(UArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8_t fa = {}, fb = {};
  asm volatile("" : "+v"(fa), "+v"(fb));
  f32x16_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x16_t{};
  f32x4_t acc4[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc4[i] = f32x4_t{};
  const char* base = p.src + (size_t)blockIdx.x * p.block_stride + (size_t)wave * p.wave_stride;
  const uint32_t voff = lane * 16;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t ub = reinterpret_cast<uint64_t>(base);
  const uint32_t ulo = __builtin_amdgcn_readfirstlane((uint32_t)ub), uhi = __builtin_amdgcn_readfirstlane((uint32_t)(ub >> 32));   // (unsigned: no sign extension of lo)
  const char* sbase = reinterpret_cast<const char*>(((uint64_t)uhi << 32) | ulo);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, 0x7ffff000, 0x00020000);
#endif
  float vv[4] = {1.f, 2.f, 3.f, 4.f}, vk = 0.999f, vc = 0.001f;
  asm volatile("" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vk), "+v"(vc));
  f32x4_t gsum = {};
  bf16x8_t lsum = {};
  const uint32_t lds0 = (uint32_t)(size_t)(UDM_LDS char*)smem;
  int soff = 0;
  auto memop = [&](int slot) {
    char* dst = smem + wave * 16384 + (slot & 15) * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
    if (KIND == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + soff + voff), (UDM_LDS void*)dst, 16, 0, 0);
    if (KIND == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (UDM_LDS void*)dst, 16, voff, soff, 0, 0);
    if (KIND == 7 && wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (UDM_LDS void*)dst, 16, voff, soff, 0, 0);
    if (KIND == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (UDM_LDS void*)dst, 4, voff >> 2, soff, 0, 0);
    if (KIND == 4) {
      f32x4_t v;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff + (uint32_t)soff), "s"(sbase) : "memory");
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // keep the loads in flight, bound the outstanding count
      gsum = v;   // (stale value semantics do not matter: timing only)
    }
    if (KIND == 5) {
      bf16x8_t v;
      asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds0 + wave * 16384 + lane * 16 + (slot & 7) * 1024) : "memory");
      lsum = v;
    }
    if (KIND == 6) asm volatile("s_nop 15" ::: "memory");
    if (KIND == 8) {   // four independent fp32 FMAs (plain VALU) in the shadow of the MFMA just issued
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]) : "v"(vk), "v"(vc));
    }
    if (KIND == 9) {   // two independent exp2 (quarter-rate transcendental)
      asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(vv[0]), "+v"(vv[1]));
    }
#endif
    soff = (soff + p.stride) & p.win_mask;
  };
  __syncthreads();
  unsigned long long t0 = 0;
  for (int it = 0; it < p.iters + 2; ++it) {
    if (it == 2) {
      __builtin_amdgcn_sched_barrier(0);
      t0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MFMA16) {
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc4[m], 0, 0, 0);
        if (NOPS > 0 && (m % (32 / NOPS)) == 1) memop(m / (32 / NOPS));
      }
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m], 0, 0, 0);
        if (NOPS > 0 && (m % (16 / NOPS)) == 0) memop(m / (16 / NOPS));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (KIND >= 1 && KIND <= 3 || KIND == 7) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // three k-steps of pieces stay in flight
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = gsum[0] + (float)lsum[0] + vv[0] + vv[1] + vv[2] + vv[3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0];
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc4[i][0];
  if (s == 12345.f) p.sink[0] = s;
  if (blockIdx.x == 0 && lane == 0) p.out[wave] = (t1 - t0);
}

template <int MFMA16, int KIND, int NOPS>
int launch_u(const UArgs& a, int blocks, hipStream_t stream) {
  auto kern = ubench_kernel<MFMA16, KIND, NOPS>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, stream, a);
  UDM_CHECK_LAUNCH("udm_ubench_issue");
  return 0;
}
}  // namespace

// mode = MFMA16 * 1000 + KIND * 100 + NOPS
extern "C" int udm_ubench_issue(int mode, int blocks, int iters, int stride, int win_mask, int block_stride, int wave_stride, const void* src, void* out4, void* sink,
                                hipStream_t stream) {
  UArgs a{(const char*)src, (unsigned long long*)out4, (float*)sink, iters, stride, win_mask, block_stride, wave_stride};
#define UB(M, K, N) case (M * 1000 + K * 100 + N): return launch_u<M, K, N>(a, blocks, stream);
  switch (mode) {
    UB(0, 0, 0) UB(1, 0, 0)
    UB(0, 1, 2) UB(0, 1, 4) UB(0, 1, 8) UB(0, 1, 16)
    UB(0, 2, 2) UB(0, 2, 4) UB(0, 2, 8) UB(0, 2, 16)
    UB(0, 3, 4) UB(0, 3, 8) UB(0, 3, 16)
    UB(0, 4, 4) UB(0, 4, 8)
    UB(0, 5, 4) UB(0, 5, 8) UB(0, 5, 16)
    UB(0, 6, 4) UB(0, 6, 8)
    UB(0, 7, 4) UB(0, 7, 8)
    UB(1, 2, 4) UB(1, 2, 8) UB(1, 2, 16)
    UB(1, 5, 8) UB(1, 5, 16)
    UB(0, 8, 4) UB(0, 8, 8) UB(0, 8, 16) UB(0, 9, 4) UB(0, 9, 8) UB(0, 9, 16)
    default: udm_set_error("udm_ubench_issue: unknown mode %d", mode); return 2;
  }
#undef UB
}
