// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 by operand register file (arch VGPR "v" / accumulator "a") of D/C, A, B and by the distance
// between MFMAs on the same accumulator.  One wave per SIMD (256 threads, 512 registers), 256 blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// FORM: 0: D=v A=a B=a | 1: D=v A=v B=a | 2: D=v A=v B=v | 3: D=a A=a B=a | 4: D=a A=v B=v | 5: D=v A=a B=v
#define M4(D0, D1, D2, D3, A, B)                                  \
  "v_mfma_f32_32x32x16_bf16 " D0 ", " A ", " B ", " D0 "\n\t"   \
  "v_mfma_f32_32x32x16_bf16 " D1 ", " A ", " B ", " D1 "\n\t"   \
  "v_mfma_f32_32x32x16_bf16 " D2 ", " A ", " B ", " D2 "\n\t"   \
  "v_mfma_f32_32x32x16_bf16 " D3 ", " A ", " B ", " D3 "\n\t"
#define M2(D0, D1, A, B) \
  "v_mfma_f32_32x32x16_bf16 " D0 ", " A ", " B ", " D0 "\n\t" \
  "v_mfma_f32_32x32x16_bf16 " D1 ", " A ", " B ", " D1 "\n\t" \
  "v_mfma_f32_32x32x16_bf16 " D0 ", " A ", " B ", " D0 "\n\t" \
  "v_mfma_f32_32x32x16_bf16 " D1 ", " A ", " B ", " D1 "\n\t"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15", \
  "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47", \
  "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a255","v200","s40","s41","s42","s43","s44","scc"
#define LOOP(BODY) \
  "s_memtime s[40:41]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s44, 200\n\tL_%=:\n\t" BODY BODY BODY BODY \
  "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 L_%=\n\ts_nop 15\n\ts_memtime s[42:43]\n\ts_waitcnt lgkmcnt(0)\n\t" \
  "s_sub_u32 %0, s42, s40\n\t"

template <int FORM, int DIST>
__global__ __launch_bounds__(256) void k(unsigned* out) {
  unsigned cyc = 0;
  if constexpr (FORM == 0 && DIST == 4) asm volatile(LOOP(M4("v[0:15]", "v[16:31]", "v[32:47]", "v[48:63]", "a[64:67]", "a[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 0 && DIST == 2) asm volatile(LOOP(M2("v[0:15]", "v[16:31]", "a[64:67]", "a[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 1 && DIST == 2) asm volatile(LOOP(M2("v[0:15]", "v[16:31]", "v[64:67]", "a[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 2 && DIST == 2) asm volatile(LOOP(M2("v[0:15]", "v[16:31]", "v[64:67]", "v[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 2 && DIST == 4) asm volatile(LOOP(M4("v[0:15]", "v[16:31]", "v[32:47]", "v[48:63]", "v[64:67]", "v[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 3 && DIST == 2) asm volatile(LOOP(M2("a[0:15]", "a[16:31]", "a[64:67]", "a[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 4 && DIST == 2) asm volatile(LOOP(M2("a[0:15]", "a[16:31]", "v[64:67]", "v[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 4 && DIST == 4) asm volatile(LOOP(M4("a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]", "v[64:67]", "v[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 5 && DIST == 2) asm volatile(LOOP(M2("v[0:15]", "v[16:31]", "a[64:67]", "v[68:71]")) : "=s"(cyc) : : CLOB);
  if constexpr (FORM == 6 && DIST == 2) asm volatile(LOOP(M2("v[0:15]", "v[16:31]", "a[64:67]", "a[64:67]")) : "=s"(cyc) : : CLOB);   // same register as A and B
  if constexpr (FORM == 7 && DIST == 2) asm volatile(LOOP(M2("v[0:15]", "v[16:31]", "a[64:67]", "a[70:73]")) : "=s"(cyc) : : CLOB);   // B at a different bank phase
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = cyc;
}
template <int FORM, int DIST>
void run(const char* name, unsigned* d) {
  unsigned h[1024];
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<FORM, DIST>), dim3(256), dim3(256), 0, 0, d);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<FORM, DIST>), dim3(256), dim3(256), 0, 0, d);
  CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
  const double n = 200.0 * 16;
  printf("%-28s cycles/MFMA %.1f   wall %.1f us  -> %.0f TF, clock %.2f GHz\n", name, s / 1024 / n, ms * 1e3, 1024 * n * 32768.0 * 2 / (ms * 1e-3) / 1e12, s / 1024 / (ms * 1e-3) / 1e9);
}
int main() {
  unsigned* d; CHK(hipMalloc(&d, 4096));
  run<4, 4>("D=a A=v B=v dist4", d);
  run<4, 2>("D=a A=v B=v dist2", d);
  run<2, 4>("D=v A=v B=v dist4", d);
  run<2, 2>("D=v A=v B=v dist2", d);
  run<0, 4>("D=v A=a B=a dist4", d);
  run<0, 2>("D=v A=a B=a dist2", d);
  run<1, 2>("D=v A=v B=a dist2", d);
  run<5, 2>("D=v A=a B=v dist2", d);
  run<3, 2>("D=a A=a B=a dist2", d);
  run<6, 2>("D=v A=a B=a(same) dist2", d);
  run<7, 2>("D=v A=a[64] B=a[70] dist2", d);
  return 0;
}
