// Micro-benchmark: what the matrix pipe SUSTAINS (power-limited clock) on v_mfma_f32_32x32x16_bf16 back to back, one wave per SIMD on all 256 CUs, for ~2 ms per
// launch and 30 launches in a row: operands all zero / random normal bf16 (the step's GEMMs see random-like data), and with one ds_read_b128 per two MFMAs
// (the GEMM main loop's LDS traffic) on top.  Reports TFLOP/s and the effective shader clock (s_memtime cycles / wall time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "s40", "s41", "s42", "s43", "s44", "scc", "vcc", "memory"
#define MF(D, A, B) "v_mfma_f32_32x32x16_bf16 a[" D "], v[" A "], v[" B "], a[" D "]\n\t"
#define MFL(D, A, B, L) MF(D, A, B) "ds_read_b128 v[" L "], v200\n\t"
#include "mfma_power_gen.h"
// variant 101: v_mfma_f32_16x16x32_bf16, 64 per body (the same 2^20 flops as 32 of the 32x32x16 form), GEMM order
#define BODY_16x16x32 "v_mfma_f32_16x16x32_bf16 a[0:3], v[64:67], v[96:99], a[0:3]\n\t" "v_mfma_f32_16x16x32_bf16 a[4:7], v[64:67], v[100:103], a[4:7]\n\t" "v_mfma_f32_16x16x32_bf16 a[8:11], v[64:67], v[104:107], a[8:11]\n\t" "v_mfma_f32_16x16x32_bf16 a[12:15], v[64:67], v[108:111], a[12:15]\n\t" "v_mfma_f32_16x16x32_bf16 a[16:19], v[68:71], v[112:115], a[16:19]\n\t" "v_mfma_f32_16x16x32_bf16 a[20:23], v[68:71], v[116:119], a[20:23]\n\t" "v_mfma_f32_16x16x32_bf16 a[24:27], v[68:71], v[120:123], a[24:27]\n\t" "v_mfma_f32_16x16x32_bf16 a[28:31], v[68:71], v[124:127], a[28:31]\n\t" "v_mfma_f32_16x16x32_bf16 a[32:35], v[72:75], v[96:99], a[32:35]\n\t" "v_mfma_f32_16x16x32_bf16 a[36:39], v[72:75], v[100:103], a[36:39]\n\t" "v_mfma_f32_16x16x32_bf16 a[40:43], v[72:75], v[104:107], a[40:43]\n\t" "v_mfma_f32_16x16x32_bf16 a[44:47], v[72:75], v[108:111], a[44:47]\n\t" "v_mfma_f32_16x16x32_bf16 a[48:51], v[76:79], v[112:115], a[48:51]\n\t" "v_mfma_f32_16x16x32_bf16 a[52:55], v[76:79], v[116:119], a[52:55]\n\t" "v_mfma_f32_16x16x32_bf16 a[56:59], v[76:79], v[120:123], a[56:59]\n\t" "v_mfma_f32_16x16x32_bf16 a[60:63], v[76:79], v[124:127], a[60:63]\n\t" "v_mfma_f32_16x16x32_bf16 a[64:67], v[80:83], v[96:99], a[64:67]\n\t" "v_mfma_f32_16x16x32_bf16 a[68:71], v[80:83], v[100:103], a[68:71]\n\t" "v_mfma_f32_16x16x32_bf16 a[72:75], v[80:83], v[104:107], a[72:75]\n\t" "v_mfma_f32_16x16x32_bf16 a[76:79], v[80:83], v[108:111], a[76:79]\n\t" "v_mfma_f32_16x16x32_bf16 a[80:83], v[84:87], v[112:115], a[80:83]\n\t" "v_mfma_f32_16x16x32_bf16 a[84:87], v[84:87], v[116:119], a[84:87]\n\t" "v_mfma_f32_16x16x32_bf16 a[88:91], v[84:87], v[120:123], a[88:91]\n\t" "v_mfma_f32_16x16x32_bf16 a[92:95], v[84:87], v[124:127], a[92:95]\n\t" "v_mfma_f32_16x16x32_bf16 a[96:99], v[88:91], v[96:99], a[96:99]\n\t" "v_mfma_f32_16x16x32_bf16 a[100:103], v[88:91], v[100:103], a[100:103]\n\t" "v_mfma_f32_16x16x32_bf16 a[104:107], v[88:91], v[104:107], a[104:107]\n\t" "v_mfma_f32_16x16x32_bf16 a[108:111], v[88:91], v[108:111], a[108:111]\n\t" "v_mfma_f32_16x16x32_bf16 a[112:115], v[92:95], v[112:115], a[112:115]\n\t" "v_mfma_f32_16x16x32_bf16 a[116:119], v[92:95], v[116:119], a[116:119]\n\t" "v_mfma_f32_16x16x32_bf16 a[120:123], v[92:95], v[120:123], a[120:123]\n\t" "v_mfma_f32_16x16x32_bf16 a[124:127], v[92:95], v[124:127], a[124:127]\n\t" "v_mfma_f32_16x16x32_bf16 a[128:131], v[64:67], v[96:99], a[128:131]\n\t" "v_mfma_f32_16x16x32_bf16 a[132:135], v[64:67], v[100:103], a[132:135]\n\t" "v_mfma_f32_16x16x32_bf16 a[136:139], v[64:67], v[104:107], a[136:139]\n\t" "v_mfma_f32_16x16x32_bf16 a[140:143], v[64:67], v[108:111], a[140:143]\n\t" "v_mfma_f32_16x16x32_bf16 a[144:147], v[68:71], v[112:115], a[144:147]\n\t" "v_mfma_f32_16x16x32_bf16 a[148:151], v[68:71], v[116:119], a[148:151]\n\t" "v_mfma_f32_16x16x32_bf16 a[152:155], v[68:71], v[120:123], a[152:155]\n\t" "v_mfma_f32_16x16x32_bf16 a[156:159], v[68:71], v[124:127], a[156:159]\n\t" "v_mfma_f32_16x16x32_bf16 a[160:163], v[72:75], v[96:99], a[160:163]\n\t" "v_mfma_f32_16x16x32_bf16 a[164:167], v[72:75], v[100:103], a[164:167]\n\t" "v_mfma_f32_16x16x32_bf16 a[168:171], v[72:75], v[104:107], a[168:171]\n\t" "v_mfma_f32_16x16x32_bf16 a[172:175], v[72:75], v[108:111], a[172:175]\n\t" "v_mfma_f32_16x16x32_bf16 a[176:179], v[76:79], v[112:115], a[176:179]\n\t" "v_mfma_f32_16x16x32_bf16 a[180:183], v[76:79], v[116:119], a[180:183]\n\t" "v_mfma_f32_16x16x32_bf16 a[184:187], v[76:79], v[120:123], a[184:187]\n\t" "v_mfma_f32_16x16x32_bf16 a[188:191], v[76:79], v[124:127], a[188:191]\n\t" "v_mfma_f32_16x16x32_bf16 a[192:195], v[80:83], v[96:99], a[192:195]\n\t" "v_mfma_f32_16x16x32_bf16 a[196:199], v[80:83], v[100:103], a[196:199]\n\t" "v_mfma_f32_16x16x32_bf16 a[200:203], v[80:83], v[104:107], a[200:203]\n\t" "v_mfma_f32_16x16x32_bf16 a[204:207], v[80:83], v[108:111], a[204:207]\n\t" "v_mfma_f32_16x16x32_bf16 a[208:211], v[84:87], v[112:115], a[208:211]\n\t" "v_mfma_f32_16x16x32_bf16 a[212:215], v[84:87], v[116:119], a[212:215]\n\t" "v_mfma_f32_16x16x32_bf16 a[216:219], v[84:87], v[120:123], a[216:219]\n\t" "v_mfma_f32_16x16x32_bf16 a[220:223], v[84:87], v[124:127], a[220:223]\n\t" "v_mfma_f32_16x16x32_bf16 a[224:227], v[88:91], v[96:99], a[224:227]\n\t" "v_mfma_f32_16x16x32_bf16 a[228:231], v[88:91], v[100:103], a[228:231]\n\t" "v_mfma_f32_16x16x32_bf16 a[232:235], v[88:91], v[104:107], a[232:235]\n\t" "v_mfma_f32_16x16x32_bf16 a[236:239], v[88:91], v[108:111], a[236:239]\n\t" "v_mfma_f32_16x16x32_bf16 a[240:243], v[92:95], v[112:115], a[240:243]\n\t" "v_mfma_f32_16x16x32_bf16 a[244:247], v[92:95], v[116:119], a[244:247]\n\t" "v_mfma_f32_16x16x32_bf16 a[248:251], v[92:95], v[120:123], a[248:251]\n\t" "v_mfma_f32_16x16x32_bf16 a[252:255], v[92:95], v[124:127], a[252:255]\n\t"
   // BODY_<name>: 32 MFMAs each, 16 accumulators a[0:255], 8 A fragments v[64:95], 8 B fragments v[96:127] (gen: see the python block in RESULTS.md / this directory)
// variant 100: round robin with one ds_read_b128 per two MFMAs
#define BODY16L \
  MFL("0:15", "64:67", "96:99", "128:131") MF("16:31", "68:71", "100:103") MFL("32:47", "72:75", "104:107", "132:135") MF("48:63", "76:79", "108:111") \
  MFL("64:79", "80:83", "112:115", "136:139") MF("80:95", "84:87", "116:119") MFL("96:111", "88:91", "120:123", "140:143") MF("112:127", "92:95", "124:127") \
  MFL("128:143", "64:67", "100:103", "144:147") MF("144:159", "68:71", "104:107") MFL("160:175", "72:75", "108:111", "148:151") MF("176:191", "76:79", "112:115") \
  MFL("192:207", "80:83", "116:119", "152:155") MF("208:223", "84:87", "120:123") MFL("224:239", "88:91", "124:127", "156:159") MF("240:255", "92:95", "96:99") \
  "s_waitcnt lgkmcnt(4)\n\t"

template <int LDSR>
__global__ __launch_bounds__(256) void k(const uint4* frag, unsigned* out, int iters) {
  extern __shared__ char smem[];
  unsigned cyc = 0;
  const unsigned tid = threadIdx.x;
  const uint4* src = frag + (tid & 63);
  const unsigned ldsaddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + tid * 16;
  // fill 64 KB of LDS with the random fragments too
  for (int i = 0; i < 16; ++i) reinterpret_cast<uint4*>(smem)[i * 256 + tid] = frag[(i * 256 + tid) & 1023];
  __syncthreads();
#define RUN(BODY)                                                                                                                      \
  asm volatile(                                                                                                                        \
      "v_mov_b32 v200, %3\n\t"    \
      "global_load_dwordx4 v[64:67], %4, off offset:0\n\t"   \
      "global_load_dwordx4 v[68:71], %4, off offset:1024\n\t"   \
      "global_load_dwordx4 v[72:75], %4, off offset:2048\n\t"   \
      "global_load_dwordx4 v[76:79], %4, off offset:3072\n\t"   \
      "global_load_dwordx4 v[80:83], %5, off offset:0\n\t"   \
      "global_load_dwordx4 v[84:87], %5, off offset:1024\n\t"   \
      "global_load_dwordx4 v[88:91], %5, off offset:2048\n\t"   \
      "global_load_dwordx4 v[92:95], %5, off offset:3072\n\t"   \
      "global_load_dwordx4 v[96:99], %6, off offset:0\n\t"   \
      "global_load_dwordx4 v[100:103], %6, off offset:1024\n\t"   \
      "global_load_dwordx4 v[104:107], %6, off offset:2048\n\t"   \
      "global_load_dwordx4 v[108:111], %6, off offset:3072\n\t"   \
      "global_load_dwordx4 v[112:115], %7, off offset:0\n\t"   \
      "global_load_dwordx4 v[116:119], %7, off offset:1024\n\t"   \
      "global_load_dwordx4 v[120:123], %7, off offset:2048\n\t"   \
      "global_load_dwordx4 v[124:127], %7, off offset:3072\n\t"   \
      "s_waitcnt vmcnt(0)\n\t"                                                                                                         \
      "s_mov_b32 s44, %2\n\t"                                                                                                          \
      "s_memtime s[40:41]\n\ts_waitcnt lgkmcnt(0)\n\t"                                                                                 \
      "L_%=:\n\t" BODY                                                                                                                 \
      "s_sub_u32 s44, s44, 1\n\ts_cmp_lg_u32 s44, 0\n\ts_cbranch_scc1 L_%=\n\ts_nop 15\n\ts_memtime s[42:43]\n\ts_waitcnt lgkmcnt(0)\n\t" \
      "s_sub_u32 %0, s42, s40\n\t"                                                                                                     \
      : "=s"(cyc), "+v"(src) : "s"(iters), "v"(ldsaddr), "v"(src), "v"(src + 256), "v"(src + 512), "v"(src + 768) : CLOB)
#define CASE(I, N) if constexpr (LDSR == I) RUN(BODY_##N);
  ALL_BODIES(CASE)
  if constexpr (LDSR == 100) RUN(BODY16L BODY16L);
  if constexpr (LDSR == 101) RUN(BODY_16x16x32);
  if (tid % 64 == 0) out[blockIdx.x * 4 + tid / 64] = cyc;
}

template <int LDSR>
void run(const char* name, const uint4* frag, unsigned* d, int iters, int reps) {
  static unsigned h[1024];
  CHK(hipFuncSetAttribute((const void*)k<LDSR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<LDSR>), dim3(256), dim3(256), 65536, 0, frag, d, iters);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<LDSR>), dim3(256), dim3(256), 65536, 0, frag, d, iters);
  CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
  const double n = (double)iters * 32;
  printf("%-44s cycles/MFMA %.1f   %.0f us per launch -> %.0f TF, clock %.2f GHz (loop only)\n", name, s / 1024 / n, ms * 1e3, 1024 * n * 32768.0 / (ms * 1e-3) / 1e12,
         s / 1024 / (ms * 1e-3) / 1e9);
}
int main() {
  unsigned* d; CHK(hipMalloc(&d, 4096));
  std::vector<unsigned short> hr(1024 * 64 * 8), hz(1024 * 64 * 8, 0);
  std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& x : hr) { float f = nd(g); unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
  uint4 *fr, *fz; CHK(hipMalloc(&fr, hr.size() * 2)); CHK(hipMalloc(&fz, hz.size() * 2));
  CHK(hipMemcpy(fr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(fz, hz.data(), hz.size() * 2, hipMemcpyHostToDevice));
  const int iters = 3000;   // 96 k MFMAs per wave ~ 1.7 ms
  for (int round = 0; round < 2; ++round) {
    run<0>("zeros, RR", fz, d, iters, 30);
#define RUNCASE(I, N) run<I>("random bf16, " #N, fr, d, iters, 30);
    ALL_BODIES(RUNCASE)
    run<100>("random, RR + ds_read_b128 per 2 MFMAs", fr, d, iters, 30);
    run<101>("random bf16, 16x16x32 form (2 per 32x32x16 MFMA counted)", fr, d, iters, 30);
  }
  return 0;
}
