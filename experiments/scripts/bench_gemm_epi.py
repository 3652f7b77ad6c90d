"""What each epilogue of the NT GEMM costs on the 1.4 B mlp shapes (M = 10240): plain / bias / bias+GELU (fc1 forward, N = 8192, K = 2048) and
plain / GELU' / GELU' + column sums (fc2 dgrad, N = 8192, K = 2048).  Operands rotate over 3 buffer sets.   python scripts/bench_gemm_epi.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M, N, Kd, R = int(os.environ.get("M", 10240)), int(os.environ.get("N", 8192)), int(os.environ.get("KD", 2048)), 3
g = torch.Generator(device="cuda").manual_seed(0)
A = [(torch.rand(M, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16) for _ in range(R)]
B = (torch.rand(N, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
out = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
aux = [torch.rand(M, N, device="cuda", generator=g).to(torch.bfloat16) for _ in range(R)]
bias = torch.zeros(N, dtype=torch.float32, device="cuda")
cases = [("plain", dict(epilogue=K.EPI_NONE)), ("bias", dict(epilogue=K.EPI_BIAS, bias=bias)), ("bias+gelu", dict(epilogue=K.EPI_BIAS_GELU, bias=bias, aux=True)),
         ("gelu'", dict(epilogue=K.EPI_DGELU, aux=True)), ("gelu'+colsum", dict(epilogue=K.EPI_DGELU, aux=True, bias=bias))]
K.gemm_set_tile(int(os.environ.get("TILE", "-1")))
K.gemm_set_quad(0)   # the 8-wave persistent kernel for every case (what the GELU epilogues run on)
for rnd in range(2):
    for name, kw in cases:
        kw = dict(kw)
        use_aux = kw.pop("aux", False)
        ts = []
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for i in range(12):
                K.gemm_nt(A[i % R], B, out=out[i % R], N=N, aux=aux[i % R] if use_aux else None, **kw)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 12 * 1e3)
        print(f"{name:14s} {min(ts):7.1f} us  {2.0 * M * N * Kd / min(ts) / 1e6:6.0f} TF")
