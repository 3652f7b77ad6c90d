import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K
kc, m, n = (int(x) for x in sys.argv[1:4])
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.rand(kc, m, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
b = (torch.rand(kc, n, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
out = torch.empty(m, n, dtype=torch.float32, device="cuda")
for _ in range(5): K.gemm_tn(a, b, out)
torch.cuda.synchronize()
