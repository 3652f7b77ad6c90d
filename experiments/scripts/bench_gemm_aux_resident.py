"""Is the GELU' epilogue's extra time the HBM read of the saved derivative?  Same GEMM (M = 10240, N = 8192, K = 2048) with the epilogue's `aux` operand (i) in HBM
(three rotating [M, N] buffers), (ii) aliased to ONE row (ldaux = 0: every read hits the L2), against the plain epilogue."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M, N, Kd, R = 10240, 8192, 2048, 3
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
        for i in range(12):
            k2 = {k: (v[i % R] if isinstance(v, list) else v) for k, v in kw.items()}
            K.gemm_nt(A[i % R], B, out=out[i % R], N=N, **k2)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 12 * 1e3)
    print(f"{name:28s} {min(ts):7.1f} us", flush=True)
for rnd in range(2):
    run("plain", epilogue=K.EPI_NONE)
    run("gelu' aux in HBM", epilogue=K.EPI_DGELU, aux=aux)
    run("gelu' aux = one row (L2)", epilogue=K.EPI_DGELU, aux=aux[0], ldaux=0)
