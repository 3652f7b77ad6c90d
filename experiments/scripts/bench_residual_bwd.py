import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from unidisc_amd import kernels as K
from bench_kernels import timeit
M, d, L = 10240, 2048, 1280
x = torch.randn(M, d, device="cuda"); y = torch.randn(M, d, device="cuda").bfloat16(); w = torch.ones(d, device="cuda"); dw = torch.zeros(d, device="cuda")
xo, rb, _ = K.residual_fwd(x, y, L, w_b=w)
print("sandwich", timeit(lambda: K.residual_bwd(x, y, L, w_b=w, rstd=rb, dw_b=dw)))
print("plain", timeit(lambda: K.residual_bwd(x, y, L)))
print("drop", timeit(lambda: K.residual_bwd(x, y, L, w_b=w, rstd=rb, dw_b=dw, p_drop=0.1, seed=5)))
