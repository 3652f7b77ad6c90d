"""GEMM fixed-cost probe: time(K) = a + b*K at fixed M, N separates per-tile prologue/epilogue cost from the steady-state rate."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (M, N) in [(10240, 6144), (10240, 2048), (10240, 8192)]:
    row = {}
    for k in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        a = (torch.rand(M, k, device="cuda") - 0.5).to(torch.bfloat16)
        b = (torch.rand(N, k, device="cuda") - 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        us = timeit(lambda: K.gemm_nt(a, b, out=out))
        ust = timeit(lambda: torch.matmul(a, b.t(), out=out))
        row[k] = (round(us, 1), round(ust, 1))
    print(json.dumps({"M": M, "N": N, "us(ours, hipblaslt) by K": row}))
