"""Emit mfma_acc_operand.hip: cycles per v_mfma_f32_32x32x16_bf16 of a lone wave per SIMD (256 CUs) when the A / B operands come from ArchVGPRs or AccVGPRs and the
accumulators (C = D) live in ArchVGPRs (the S / dP accumulators of the attention kernels) or AccVGPRs.  Round 6: is an AGPR-fed MFMA slower?"""
import sys

def body(acc, afile, bfile):
    out = []
    for i in range(32):
        d = f"{acc}[{16 * (i % 8)}:{16 * (i % 8) + 15}]"
        a = f"{afile}[{128 + 4 * (i % 8)}:{128 + 4 * (i % 8) + 3}]"
        b = f"{bfile}[{160 + 4 * ((i // 2) % 8)}:{160 + 4 * ((i // 2) % 8) + 3}]"
        out.append(f'"v_mfma_f32_32x32x16_bf16 {d}, {a}, {b}, {d}\\n\\t"')
    return " ".join(out)

variants = [("acc VGPR, A VGPR, B VGPR", "v", "v", "v"), ("acc VGPR, A VGPR, B AGPR", "v", "v", "a"), ("acc VGPR, A AGPR, B VGPR", "v", "a", "v"), ("acc VGPR, A AGPR, B AGPR", "v", "a", "a"),
            ("acc AGPR, A VGPR, B VGPR", "a", "v", "v"), ("acc AGPR, A VGPR, B AGPR", "a", "v", "a")]
clob = ", ".join([f'"v{i}"' for i in range(200)] + [f'"a{i}"' for i in range(200)] + ['"s40"', '"s41"', '"s42"', '"s43"', '"s44"', '"scc"', '"memory"'])
src = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', 'template <int V> __global__ __launch_bounds__(256) void k(unsigned* out, int iters) {', '  unsigned cyc = 0;']
for n, (_, acc, af, bf) in enumerate(variants):
    src.append(f'  if constexpr (V == {n}) asm volatile("s_mov_b32 s44, %1\\n\\ts_memtime s[40:41]\\n\\ts_waitcnt lgkmcnt(0)\\n\\tL_%=:\\n\\t" {body(acc, af, bf)} '
               f'"s_sub_u32 s44, s44, 1\\n\\ts_cmp_lg_u32 s44, 0\\n\\ts_cbranch_scc1 L_%=\\n\\ts_nop 15\\n\\ts_memtime s[42:43]\\n\\ts_waitcnt lgkmcnt(0)\\n\\ts_sub_u32 %0, s42, s40\\n\\t" : "=s"(cyc) : "s"(iters) : {clob});')
src += ['  if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = cyc;', '}',
        'template <int V> void run(const char* name, unsigned* d) {', '  static unsigned h[1024]; const int iters = 2000;',
        '  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), 0, 0, d, iters);', '  hipDeviceSynchronize();',
        '  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];',
        '  printf("%-30s cycles per MFMA %.2f\\n", name, s / 1024 / (iters * 32.0));', '}',
        'int main() { unsigned* d; hipMalloc(&d, 4096);']
for n, (name, *_r) in enumerate(variants):
    src.append(f'  run<{n}>("{name}", d);')
src += ['  return 0; }']
open(sys.argv[1] if len(sys.argv) > 1 else "mfma_acc_operand.hip", "w").write("\n".join(src) + "\n")
