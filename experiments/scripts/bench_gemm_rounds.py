"""GEMM round-count probe: same N, K, growing M (1, 2, 4 rounds of the 256 CUs) -- ours vs hipBLASLt."""
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

for (N, k) in [(8192, 8192), (8192, 2048), (6144, 2048)]:
    for M in (2560, 5120, 10240, 20480):
        a = (torch.rand(M, k, device="cuda") - 0.5).to(torch.bfloat16)
        b = (torch.rand(N, k, device="cuda") - 0.5).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        res = {}
        for tile in (0, 256, 320):
            K.gemm_set_tile(tile)
            us = timeit(lambda: K.gemm_nt(a, b, out=out))
            res[f"tile{tile}"] = round(2 * M * N * k / us / 1e6, 1)
        K.gemm_set_tile(0)
        ust = timeit(lambda: torch.matmul(a, b.t(), out=out))
        res["hipblaslt"] = round(2 * M * N * k / ust / 1e6, 1)
        print(json.dumps({"M": M, "N": N, "K": k, "TF": res}))
