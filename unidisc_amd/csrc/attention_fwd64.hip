// Attention forward at head dim 128 without a mask, L % 256 == 0, q pre-scaled by log2(e) / sqrt(D) (UDM_ATTN_Q_PRESCALED): ONE wave per SIMD, 64 queries
// per wave, persistent workgroups - the whole workgroup program is the
// hand-scheduled instruction stream that asmgen/attn_fwd64.py generates (registers, LDS layout, schedule: see that file; it is linted for
// hazards and executed on a CPU emulator by tests/test_asmgen.py before it ships).  This file only computes the block's scalars and launches.
// Replaces flash_attn_qkvpacked_func (reference models/dit.py:843) on the headline path; attention.hip keeps every other shape.
#include "attention_common.h"
#include "gemm_quad.h"
#include "attention_fwd64_gen.h"

#include <stdlib.h>

namespace {
// ABLV != 0: timing-only ablations of the tile loop (wrong results; built with `make UDM_FWD64_ABL="1 3 7 ..."`, picked by UDM_ATTN_FWD64_ABL)
template <int ABLV>
__global__ __launch_bounds__(256) void attn_fwd64_kernel(AttnArgs a, uint32_t nt, uint32_t mg_nt, uint32_t mg_h, uint32_t nfull, uint32_t hashalf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t qstr = (uint32_t)(a.q_stride * 2), kstr = (uint32_t)(a.k_stride * 2), vstr = (uint32_t)(a.v_stride * 2), ostr = (uint32_t)(a.out_stride * 2);
  const uint32_t L = (uint32_t)a.L, nkv = (uint32_t)(a.L / 64), H = (uint32_t)a.H;
  const uint32_t lds = (uint32_t)(size_t)(UDM_LDS char*)smem;
  const uint32_t bid = blockIdx.x, gstride = gridDim.x, tid = threadIdx.x;
#define UDM_FWD64_RUN(TEXT)                                                                                                                                  \
  asm volatile(TEXT : : "s"(a.q), "s"(a.k), "s"(a.v), "s"(a.out), "s"(a.lse), "s"(qstr), "s"(kstr), "s"(vstr), "s"(ostr), "s"(L), "s"(nkv), "s"(H), "s"(nt), \
               "s"(mg_nt), "s"(mg_h), "s"(nfull), "s"(hashalf), "s"(lds), "s"(bid), "s"(gstride), "v"(tid), "s"(a.timeline)                              \
               : UDM_FWD64_CLOBBERS)
  if constexpr (ABLV == 0) UDM_FWD64_RUN(UDM_FWD64_ASM);
#ifdef UDM_FWD64_ASM_ABL1
  if constexpr (ABLV == 1) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL1);
#endif
#ifdef UDM_FWD64_ASM_ABL2
  if constexpr (ABLV == 2) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL2);
#endif
#ifdef UDM_FWD64_ASM_ABL3
  if constexpr (ABLV == 3) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL3);
#endif
#ifdef UDM_FWD64_ASM_ABL4
  if constexpr (ABLV == 4) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL4);
#endif
#ifdef UDM_FWD64_ASM_ABL5
  if constexpr (ABLV == 5) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL5);
#endif
#ifdef UDM_FWD64_ASM_ABL7
  if constexpr (ABLV == 7) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL7);
#endif
#ifdef UDM_FWD64_ASM_ABL9
  if constexpr (ABLV == 9) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL9);
#endif
#ifdef UDM_FWD64_ASM_ABL32
  if constexpr (ABLV == 32) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL32);   // debug: the first block's scalars into the LSE tensor
#endif
#ifdef UDM_FWD64_ASM_ABL16
  if constexpr (ABLV == 16) UDM_FWD64_RUN(UDM_FWD64_ASM_ABL16);   // cycle stamps (correct results) -> a.timeline [workgroups][4 waves][64] uint32
#endif
#undef UDM_FWD64_RUN
}
int g_fwd64 = -1;
unsigned long long* g_timeline = nullptr;
}  // namespace

extern "C" __attribute__((visibility("hidden"))) int udm_attention_set_fwd64_timeline(int64_t device_ptr) {   // diagnostics (a build with UDM_FWD64_ABL=16): stamps of the next launches, 0 = off
  g_timeline = reinterpret_cast<unsigned long long*>(device_ptr);
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int udm_attention_set_fwd64(int enable) {   // tests / A-B measurements: 0 = the 8-wave kernel of attention.hip everywhere
  g_fwd64 = enable;        // 2: without the balanced walk (whole blocks only)
  return 0;
}

// the forward of attention.hip's dispatch for (D = 128, no sample ids): returns false when this kernel does not take the shape
bool udm_launch_attn_fwd64(const void* args, hipStream_t stream) {
  AttnArgs a = *reinterpret_cast<const AttnArgs*>(args);
  a.timeline = g_timeline;
  if (g_fwd64 < 0) { const char* e = getenv("UDM_ATTN_FWD64"); g_fwd64 = e ? atoi(e) : 1; }
  // whole 256-query blocks, at least two trips of the four-tile loop, the XCD-sequential block order of attention.hip (B H a multiple of 8)
  if (!g_fwd64 || !a.q_prescaled || a.H < 2 /* magic(1) wraps: the head divisor would read 0 */ || a.L % 256 != 0 || a.L < 512 || a.out_stride % 8 != 0 || (a.B * a.H) % 8 != 0) return false;
  if (a.q_stride * 2 * 256 >= (1L << 31) || a.k_stride * 2 * 80 >= (1L << 31) || a.v_stride * 2 * 80 >= (1L << 31) || a.out_stride * 2 * 256 >= (1L << 31)) return false;   // 32-bit lane offsets
  const long nt = a.L / 256, nblk = nt * a.B * a.H;
  if ((long)a.B * a.L >= (1L << 30) || nblk >= (1L << 24) || nt > 4096 || a.H > 4096 || (long)a.B * a.H * a.L >= (1L << 29)) return false;   // 32-bit row / lse indices, exact magic divisions
  static const int abl = [] { const char* e = getenv("UDM_ATTN_FWD64_ABL"); return e ? atoi(e) : 0; }();
  auto kern = attn_fwd64_kernel<0>;
  switch (abl) {
    case 1: kern = attn_fwd64_kernel<1>; break;
    case 2: kern = attn_fwd64_kernel<2>; break;
    case 3: kern = attn_fwd64_kernel<3>; break;
    case 4: kern = attn_fwd64_kernel<4>; break;
    case 5: kern = attn_fwd64_kernel<5>; break;
    case 7: kern = attn_fwd64_kernel<7>; break;
    case 9: kern = attn_fwd64_kernel<9>; break;
    case 32: kern = attn_fwd64_kernel<32>; break;
    default: break;
  }
  if (g_timeline) {
#ifdef UDM_FWD64_ASM_ABL16
    kern = attn_fwd64_kernel<16>;
#else
    udm_set_error("udm_attention_fwd: timeline requested but the library was built without UDM_FWD64_ABL=16");
#endif
  }
  static const void* attr_set = nullptr;
  if (attr_set != (const void*)kern) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, UDM_FWD64_LDS_BYTES); attr_set = (const void*)kern; }
  static const int dev_cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n / 8 * 8; }();
  // a persistent workgroup needs a whole CU: while a collective's channel kernels hold CUs (udm_gemm_set_cus, the data-parallel schedule `overlap_planned`) the grid is
  // what is left - a workgroup that finds no CU would start its whole walk only when another has finished its own
  const int plan_cus = udm_gemm_cus_available() / 8 * 8;
  const int cus = plan_cus >= 8 && plan_cus < dev_cus ? plan_cus : dev_cus;
  const auto magic = [](long d) { return (uint32_t)((1ULL << 32) / (unsigned long long)d + 1); };   // n / d == mulhi(n, magic) for n d < 2^32
  const long grid = nblk < cus ? nblk : cus;    // persistent: one workgroup per CU walks blocks id, id + grid, ...
  // balanced walk: when the blocks left behind the whole rounds are exactly half a grid (the headline's 640 blocks on 256 CUs), every workgroup ends with ONE
  // 128-query half block (2.5 units each) instead of a third whole block for half of them (3 vs 2)
  const long rem = nblk % grid;
  const bool halves = g_fwd64 != 2 && rem * 2 == grid && nblk - rem >= grid && grid % 16 == 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), UDM_FWD64_LDS_BYTES, stream, a, (uint32_t)nt, magic(nt), magic(a.H), (uint32_t)(halves ? nblk - rem : nblk), halves ? 1u : 0u);
  return true;
}
