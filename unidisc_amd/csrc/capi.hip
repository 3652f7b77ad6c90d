// Error plumbing and version entry points of the C ABI (include/unidisc_hip.h).
#include "common.h"
#include "gemm_quad.h"
#include "../../include/unidisc_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

namespace {
thread_local char g_err[512] = "";
}

void udm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* udm_last_error(void) { return g_err; }
extern "C" int udm_abi_version(void) { return UDM_ABI_VERSION; }

// ---- diagnostics behind ONE entry point (the setters themselves live next to the state they switch; hidden visibility: the library exports exactly what
// include/unidisc_hip.h declares, tests/test_capi.py checks both directions) ----
extern "C" __attribute__((visibility("hidden"))) int udm_gemm_set_tile(int tile);      // gemm.hip: force the tile family (-1 auto, 0 = 128x128 kernel, 192 / 256 / 320 = BM x 256 kernel)
extern "C" __attribute__((visibility("hidden"))) int udm_gemm_set_quad(int mode);      // gemm.hip: one-wave-per-SIMD kernels 0 = off, 1 = auto (default; env UDM_GEMM_QUAD), 2 = wherever the shape fits
extern "C" __attribute__((visibility("hidden"))) int udm_gemm_set_persist(int enable); // gemm.hip: 0 = one block per output tile (default 1: persistent blocks for multi-round NT shapes)
extern "C" __attribute__((visibility("hidden"))) int udm_attention_set_tr_read(int enable);        // attention.hip: 0 = gather V^T fragments with scalar LDS reads
extern "C" __attribute__((visibility("hidden"))) int udm_attention_set_fwd64_timeline(int64_t device_ptr);   // attention_fwd64.hip diagnostics
extern "C" __attribute__((visibility("hidden"))) int udm_attention_set_fwd64(int enable);          // attention_fwd64.hip: 0 = the 8-wave forward also at D = 128, L % 256 == 0 (default 1; env UDM_ATTN_FWD64)

void udm_attention_set_dkv64(int enable);                                     // attention_dkv64.hip: 0 = the wave-specialised dK / dV kernel everywhere, 2 = without the balanced walk
void udm_attention_set_dkv64_timeline(int64_t device_ptr);
void udm_attention_set_dq64(int enable);                                      // attention_dq64.hip: 0 = attn_bwd_dq_kernel everywhere, 2 = without the balanced walk
void udm_attention_set_dq64_timeline(int64_t device_ptr);

namespace { int g_exp = [] { const char* e = getenv("UDM_EXP"); return e ? atoi(e) : 0; }(); }
int udm_exp_flags() { return g_exp; }

extern "C" int udm_debug_set(const char* key, int64_t value) {
  if (!key) { udm_set_error("udm_debug_set: null key"); return 2; }
  const auto is = [&](const char* k) { return strcmp(key, k) == 0; };
  if (is("exp")) { g_exp = (int)value; return 0; }
  if (is("gemm_tile")) return udm_gemm_set_tile((int)value);
  if (is("gemm_quad")) return udm_gemm_set_quad((int)value);
  if (is("gemm_persist")) return udm_gemm_set_persist((int)value);
  if (is("gemm_quad_timeline")) { g_quad_timeline = reinterpret_cast<unsigned*>(value); return 0; }
  if (is("gemm_quad_asm")) { g_quad_asm = (int)value; return 0; }   // 0 = the C++ K loop of gemm_quad.hip everywhere, 1 = the generated asm loop where it exists, -1 = env
  if (is("attention_tr_read")) return udm_attention_set_tr_read((int)value);
  if (is("attention_fwd64")) return udm_attention_set_fwd64((int)value);
  if (is("attention_fwd64_timeline")) return udm_attention_set_fwd64_timeline(value);
  if (is("attention_dkv64")) { udm_attention_set_dkv64((int)value); return 0; }
  if (is("attention_dq64")) { udm_attention_set_dq64((int)value); return 0; }
  if (is("attention_dq64_timeline")) { udm_attention_set_dq64_timeline(value); return 0; }
  if (is("attention_dkv64_timeline")) { udm_attention_set_dkv64_timeline(value); return 0; }
  udm_set_error("udm_debug_set: unknown key '%s'", key);
  return 2;
}

// Diagnostics: occupy `blocks` CUs (one block each: 160 KiB of LDS, so no LDS-using kernel can share the CU) until *flag != 0 - the stand-in for a collective's
// channel kernels when the CU-reservation behaviour of the GEMMs is measured on one GPU (bench.py --hog-cus).  `flag` must be host-visible pinned memory or
// device memory another stream writes; the kernel polls it with system-scope loads.
__global__ __launch_bounds__(64) void cu_hog_kernel(const int* flag) {
  extern __shared__ char hog_lds[];
  if (threadIdx.x == 0) {
    hog_lds[0] = 1;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) __builtin_amdgcn_s_sleep(32);
  }
}
extern "C" int udm_debug_cu_hog(int64_t blocks, const int* flag, hipStream_t stream) {
  UDM_CHECK_ARG(blocks > 0 && blocks <= 128 && flag, "udm_debug_cu_hog: 1..128 blocks and a flag pointer");
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(cu_hog_kernel, dim3((unsigned)blocks), dim3(64), 160 * 1024, stream, flag);
  UDM_CHECK_LAUNCH("udm_debug_cu_hog");
  return 0;
}
