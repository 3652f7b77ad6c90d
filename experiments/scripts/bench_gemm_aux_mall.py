"""Does a saved-derivative tile that sits in the Infinity Cache (MALL) read faster than one in HBM?  GELU' GEMM of ONE round of tiles (M = 2560, N = 8192, K = 2048:
42 MB of `aux`), (i) the same aux buffer every call (stays in the 256 MB MALL, too big for the 32 MB of L2), (ii) 8 rotating aux buffers (336 MB: from HBM),
(iii) aux aliased to one row (L2), (iv) plain epilogue."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M, N, Kd, R = 2560, 8192, 2048, 8
g = torch.Generator(device="cuda").manual_seed(0)
A = [(torch.rand(M, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16) for _ in range(R)]
B = (torch.rand(N, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
out = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
aux = [torch.rand(M, N, device="cuda", generator=g).to(torch.bfloat16) for _ in range(R)]
K.gemm_set_quad(0)
def run(name, **kw):
    ts = []
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for i in range(24):
            k2 = {k: (v[i % R] if isinstance(v, list) else v) for k, v in kw.items()}
            K.gemm_nt(A[i % R], B, out=out[i % R], N=N, **k2)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 24 * 1e3)
    print(f"{name:34s} {min(ts):7.1f} us", flush=True)
for rnd in range(2):
    run("plain", epilogue=K.EPI_NONE)
    run("gelu' aux rotating (HBM)", epilogue=K.EPI_DGELU, aux=aux)
    run("gelu' aux fixed (MALL)", epilogue=K.EPI_DGELU, aux=aux[0])
    run("gelu' aux = one row (L2)", epilogue=K.EPI_DGELU, aux=aux[0], ldaux=0)
