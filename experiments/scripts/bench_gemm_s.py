"""UniDisc-S GEMM shapes (M = 64 x 384 rows, d = 768): the NT kernel's tile choice (auto / forced 192 / 256 / 320 rows) per shape and epilogue.
Operands rotate over 4 buffer sets so that nothing is served from the Infinity Cache.   python scripts/bench_gemm_s.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M = 64 * 384
SHAPES = [("qkv fwd", 2304, 768, K.EPI_NONE), ("out fwd", 768, 768, K.EPI_NONE), ("fc1 fwd gelu", 3072, 768, K.EPI_BIAS_GELU), ("fc2 fwd bias", 768, 3072, K.EPI_BIAS),
          ("fc2 dgrad gelu'", 3072, 768, K.EPI_DGELU)]
g = torch.Generator(device="cuda").manual_seed(0)
R = 4
for name, N, Kd, epi in SHAPES:
    A = [(torch.rand(M, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16) for _ in range(R)]
    B = (torch.rand(N, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    out = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
    aux = [(torch.rand(M, N, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(R)] if epi in (K.EPI_BIAS_GELU, K.EPI_DGELU) else [None] * R
    bias = torch.zeros(N, dtype=torch.float32, device="cuda")
    line = f"{name:16s} N={N:5d} K={Kd:5d}:"
    for tile in (-1, 192, 256, 320, "quad"):
        if tile == "quad":
            K.gemm_set_tile(-1), K.gemm_set_quad(2)
        else:
            K.gemm_set_tile(tile), K.gemm_set_quad(1)
        kw = dict(N=N, epilogue=epi)
        if epi != K.EPI_NONE:
            kw["bias"] = bias
        ts = []
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for i in range(8):
                K.gemm_nt(A[i % R], B, out=out[i % R], aux=aux[i % R], **kw)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 8 * 1e3)
        t = min(ts)
        line += f"  {'auto' if tile == -1 else tile}: {t:6.1f} us ({2.0 * M * N * Kd / t / 1e6:5.0f} TF)"
    K.gemm_set_tile(-1), K.gemm_set_quad(1)
    print(line)
