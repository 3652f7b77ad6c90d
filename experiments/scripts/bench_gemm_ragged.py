"""Ragged single-round NT shapes of config E (M = 9216, N = 2048): the one-wave-per-SIMD kernel's ragged form against the 8-wave kernel (UDM_QUAD_RAGGED=0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K
M, N, R = int(os.environ.get("M", 9216)), 2048, 3
g = torch.Generator(device="cuda").manual_seed(0)
bias = torch.zeros(N, dtype=torch.float32, device="cuda")
for Kd, kw in ((8192, dict(epilogue=K.EPI_BIAS, bias=bias)), (2048, dict())):
    A = [(torch.rand(M, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16) for _ in range(R)]
    B = (torch.rand(N, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    W = (torch.rand(Kd, N, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    out = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
    for name, fn in (("nt plain", lambda i: K.gemm_nt(A[i % R], B, out=out[i % R], N=N)), ("nt", lambda i: K.gemm_nt(A[i % R], B, out=out[i % R], N=N, **kw)), ("nn", lambda i: K.gemm_nn(A[i % R], W, out=out[i % R])), ("nt", lambda i: K.gemm_nt(A[i % R], B, out=out[i % R], N=N, **kw)), ("nt plain", lambda i: K.gemm_nt(A[i % R], B, out=out[i % R], N=N)), ("nn", lambda i: K.gemm_nn(A[i % R], W, out=out[i % R]))):
        ts = []
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for i in range(12): fn(i)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 12 * 1e3)
        print(f"K={Kd} {name:9s} {min(ts):7.1f} us  {2.0 * M * N * Kd / min(ts) / 1e6:6.0f} TF", flush=True)
