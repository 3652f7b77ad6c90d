// Attention backward, dQ pass (+ delta = rowsum(dO * O) and the planes delta | -lse | -delta the dK / dV pass reads), at head dim 128 without a mask, L % 256 == 0,
// q pre-scaled by log2(e) / sqrt(D) (UDM_ATTN_Q_PRESCALED): ONE wave per SIMD, 64 queries per wave, persistent workgroups - the whole workgroup program is the
// hand-scheduled instruction stream that asmgen/attn_dq64.py generates (registers, LDS layout, schedule: see that file; linted for hazards and executed on a CPU
// emulator by tests/test_asmgen.py before it ships).  This file only fills the program's parameter block and launches.  Replaces the dQ half of the backward of
// flash_attn_qkvpacked_func (reference models/dit.py:843) on the headline path; attn_bwd_dq_kernel of attention.hip keeps every other shape.
#include "attention_common.h"
#include "gemm_quad.h"
#include "attention_dq64_gen.h"

#include <stdlib.h>

namespace {
// the program's parameter block = the kernel's argument (kernarg segment): asmgen/attn_dkv64.py reads it with s_load at these dword offsets (P_*)
struct Dq64Params {
  const void* k; const void* v; uint32_t kstr, vstr, L, nsteps, H, nt, mg_nt, mg_H, nfull, hashalf, gstride, planeB;   // 0, 2, 4 .. 15 (strides in bytes)
  const void* q; const void* dout; const void* o; const float* lse;        // 16, 18, 20, 22
  uint32_t qstr, dostr, ostr, pad;                                         // 24 .. 27
  float* delta; void* dq; uint32_t dqstr; float scale;                     // 28, 30, 32, 33
  unsigned long long* timeline;                                            // 34
};
static_assert(sizeof(Dq64Params) == 4 * UDM_DQ64_PARAM_DWORDS, "parameter block layout");
static_assert(offsetof(Dq64Params, q) == 4 * 16 && offsetof(Dq64Params, qstr) == 4 * 24 && offsetof(Dq64Params, delta) == 4 * 28 && offsetof(Dq64Params, dqstr) == 4 * 32 &&
              offsetof(Dq64Params, timeline) == 4 * 34, "parameter block layout");

template <int ABLV>
__global__ __launch_bounds__(256) void attn_dq64_kernel(Dq64Params p) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();   // (address space 4: a 64-bit pointer in an SGPR pair)
  const uint32_t lds = (uint32_t)(size_t)(UDM_LDS char*)smem;
  const uint32_t bid = blockIdx.x, tid = threadIdx.x;
#define UDM_DQ64_RUN(TEXT) asm volatile(TEXT : : "s"(kp), "s"(bid), "s"(lds), "v"(tid) : UDM_DQ64_CLOBBERS)
  if constexpr (ABLV == 0) UDM_DQ64_RUN(UDM_DQ64_ASM);
#ifdef UDM_DQ64_ASM_ABL1
  if constexpr (ABLV == 1) UDM_DQ64_RUN(UDM_DQ64_ASM_ABL1);
#endif
#ifdef UDM_DQ64_ASM_ABL2
  if constexpr (ABLV == 2) UDM_DQ64_RUN(UDM_DQ64_ASM_ABL2);
#endif
#ifdef UDM_DQ64_ASM_ABL4
  if constexpr (ABLV == 4) UDM_DQ64_RUN(UDM_DQ64_ASM_ABL4);
#endif
#ifdef UDM_DQ64_ASM_ABL8
  if constexpr (ABLV == 8) UDM_DQ64_RUN(UDM_DQ64_ASM_ABL8);
#endif
#ifdef UDM_DQ64_ASM_ABL16
  if constexpr (ABLV == 16) UDM_DQ64_RUN(UDM_DQ64_ASM_ABL16);   // cycle stamps (correct results) -> p.timeline [workgroups][4 waves][64] uint32
#endif
#undef UDM_DQ64_RUN
  (void)p;
}
int g_dq64 = -1;
unsigned long long* g_dq64_timeline = nullptr;
}  // namespace

void udm_attention_set_dq64(int enable) { g_dq64 = enable; }                                     // tests / A-B measurements (through udm_debug_set)
void udm_attention_set_dq64_timeline(int64_t device_ptr) { g_dq64_timeline = reinterpret_cast<unsigned long long*>(device_ptr); }

// the dQ pass of attention.hip's backward dispatch for (D = 128, no sample ids): returns false when this kernel does not take the shape
bool udm_launch_attn_bwd_dq64(const void* args, hipStream_t stream) {
  const AttnArgs& a = *reinterpret_cast<const AttnArgs*>(args);
  if (g_dq64 < 0) { const char* e = getenv("UDM_ATTN_DQ64"); g_dq64 = e ? atoi(e) : 1; }
  // whole 256-query blocks of at least two per (batch, head) (the magic divisions), the XCD-sequential block order (B H a multiple of 8), 16-byte row segments
  if (!g_dq64 || !a.q_prescaled || a.H < 2 || a.L % 256 != 0 || a.L < 512 || (a.B * a.H) % 8 != 0) return false;
  if (a.out_stride % 8 != 0 || a.o_stride % 8 != 0 || a.q_stride % 8 != 0 || a.do_stride % 8 != 0) return false;
  const long lim = 1L << 31;     // 32-bit lane offsets: 64 rows of any operand, and the plane offset
  if (a.q_stride * 2 * 256 >= lim || a.do_stride * 2 * 256 >= lim || a.o_stride * 2 * 256 >= lim || a.k_stride * 2 * 64 >= lim || a.v_stride * 2 * 64 >= lim || a.out_stride * 2 * 256 >= lim) return false;
  const long nt = a.L / 256, nblk = nt * a.B * a.H, plane = (long)a.B * a.H * a.L;
  if ((long)a.B * a.L >= (1L << 30) || nblk >= (1L << 24) || nt > 4096 || a.H > 4096 || plane >= (1L << 29)) return false;   // 32-bit row / plane indices, exact magic divisions
  static const int abl = [] { const char* e = getenv("UDM_ATTN_DQ64_ABL"); return e ? atoi(e) : 0; }();
  auto kern = attn_dq64_kernel<0>;
  switch (abl) {
    case 1: kern = attn_dq64_kernel<1>; break;
    case 2: kern = attn_dq64_kernel<2>; break;
    case 4: kern = attn_dq64_kernel<4>; break;
    case 8: kern = attn_dq64_kernel<8>; break;
    default: break;
  }
  if (g_dq64_timeline) {
#ifdef UDM_DQ64_ASM_ABL16
    kern = attn_dq64_kernel<16>;
#else
    udm_set_error("udm_attention_bwd: dQ timeline requested but the library was built without UDM_DQ64_ABL=16");
#endif
  }
  static const void* attr_set = nullptr;
  if (attr_set != (const void*)kern) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, UDM_DQ64_LDS_BYTES); attr_set = (const void*)kern; }
  static const int dev_cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n / 8 * 8; }();
  // a persistent workgroup needs a whole CU: while a collective's channel kernels hold CUs (udm_gemm_set_cus, the data-parallel schedule `overlap_planned`) the grid is
  // what is left - a workgroup that finds no CU would start its whole walk only when another has finished its own
  const int plan_cus = udm_gemm_cus_available() / 8 * 8;
  const int cus = plan_cus >= 8 && plan_cus < dev_cus ? plan_cus : dev_cus;
  const auto magic = [](long d) { return (uint32_t)((1ULL << 32) / (unsigned long long)d + 1); };   // n / d == mulhi(n, magic) for n d < 2^32, d >= 2
  const long grid = nblk < cus ? nblk : cus;    // persistent: one workgroup per CU walks blocks id, id + grid, ...
  // balanced walk (as the forward): when the blocks behind the whole rounds are exactly half a grid (the headline's 640 blocks on 256 CUs) every workgroup ends with
  // ONE 128-query half block instead of a third whole block for half of them
  const long rem = nblk % grid;
  const bool halves = g_dq64 != 2 && rem * 2 == grid && nblk - rem >= grid && grid % 16 == 0;
  Dq64Params p{};
  p.k = a.k; p.v = a.v; p.kstr = (uint32_t)(a.k_stride * 2); p.vstr = (uint32_t)(a.v_stride * 2); p.L = (uint32_t)a.L; p.nsteps = (uint32_t)(a.L / 32); p.H = (uint32_t)a.H; p.nt = (uint32_t)nt;
  p.mg_nt = magic(nt); p.mg_H = magic(a.H); p.nfull = (uint32_t)(halves ? nblk - rem : nblk); p.hashalf = halves ? 1u : 0u; p.gstride = (uint32_t)grid; p.planeB = (uint32_t)(plane * 4);
  p.q = a.q; p.dout = a.dout; p.o = a.o; p.lse = a.lse;
  p.qstr = (uint32_t)(a.q_stride * 2); p.dostr = (uint32_t)(a.do_stride * 2); p.ostr = (uint32_t)(a.o_stride * 2);
  p.delta = const_cast<float*>(a.delta); p.dq = a.out; p.dqstr = (uint32_t)(a.out_stride * 2); p.scale = a.scale;      // planes: delta | -lse | -delta
  p.timeline = g_dq64_timeline;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), UDM_DQ64_LDS_BYTES, stream, p);
  return true;
}
