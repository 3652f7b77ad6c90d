"""A/B of the NT one-wave-per-SIMD GEMM with the generated asm K loop (csrc/asmgen/gemm_loop.py) against its C++ K loop: bit-equality and time on the step's
shapes, operands rotating over 3 buffer sets, interleaved rounds in one process."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M = 10240
shapes = [(2048, 8192, K.EPI_BIAS, "fc2 fwd"), (2048, 2048, K.EPI_NONE, "out-proj"), (8192, 2048, K.EPI_NONE, "N=8192 plain (4 rounds)"), (6144, 2048, K.EPI_NONE, "qkv")]
g = torch.Generator(device="cuda").manual_seed(0)
K.gemm_set_quad(2)
res = {}
for (N, Kd, epi, name) in shapes:
    sets = [(torch.randn(M, Kd, device="cuda", generator=g).bfloat16(), torch.randn(N, Kd, device="cuda", generator=g).bfloat16()) for _ in range(3)]
    bias = torch.randn(N, device="cuda", generator=g)
    outs = {}
    for flag in (1, 0, 2):
        K.debug_set("gemm_quad_asm", flag)
        outs[flag] = K.gemm_nt(sets[0][0], sets[0][1], epilogue=epi, bias=bias if epi == K.EPI_BIAS else None).clone()
    ref = (sets[0][0][:512].float() @ sets[0][1].float().t()) + (bias if epi == K.EPI_BIAS else 0)
    err = float((outs[1][:512].float() - ref).abs().max() / ref.abs().max())
    equal = bool(torch.equal(outs[0], outs[1]))
    err16 = float((outs[2][:512].float() - ref).abs().max() / ref.abs().max())      # the 16x16x32 loop: another summation order, same tolerance
    diff16 = float((outs[2].float() - outs[0].float()).abs().max() / outs[0].float().abs().max())
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    times = {1: [], 0: [], 2: []}
    for rnd in range(4):
        for flag in (1, 0, 2):
            K.debug_set("gemm_quad_asm", flag)
            ts = []
            for it in range(12):
                a, b = sets[it % 3]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                K.gemm_nt(a, b, out, epilogue=epi, bias=bias if epi == K.EPI_BIAS else None)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            times[flag].append(round(ts[len(ts) // 2], 1))
    fl = 2 * M * N * Kd
    res[name] = dict(N=N, K=Kd, bit_equal=equal, rel_err_vs_fp32=err, rel_err_16x16x32_vs_fp32=err16, max_diff_16x16x32_vs_cpp=diff16, asm_us=times[1],
                     cpp_us=times[0], asm16_us=times[2], asm_tf=round(fl / min(times[1]) / 1e6, 1), cpp_tf=round(fl / min(times[0]) / 1e6, 1), asm16_tf=round(fl / min(times[2]) / 1e6, 1))
    print(name, res[name], flush=True)
K.debug_set("gemm_quad_asm", -1)
K.gemm_set_quad(1)
print(json.dumps(res))
