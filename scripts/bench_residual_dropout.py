"""What the Philox regeneration costs in the residual kernels at the headline shape (M = 10 240, d = 2048): fused residual + next norm forward and fused
norm + residual backward with p = 0 and p = 0.1, interleaved rounds, operands rotating over 3 buffer sets (not L2 / MALL resident)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M, d, L = 10240, 2048, 1280
g = torch.Generator(device="cuda").manual_seed(0)
sets = []
for _ in range(3):
    sets.append(dict(x=torch.randn(M, d, device="cuda", generator=g), br=torch.randn(M, d, device="cuda", generator=g).bfloat16(),
                     dy=torch.randn(M, d, device="cuda", generator=g).bfloat16(), dx=torch.randn(M, d, device="cuda", generator=g)))
w, wb = torch.ones(d, device="cuda"), torch.ones(d, device="cuda")
dw, dwb, dbias = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
res = {}
for rnd in range(3):
    for p in (0.0, 0.1):
        for which in ("fwd", "bwd"):
            ts = []
            for it in range(15):
                s = sets[it % 3]
                if which == "bwd":
                    _, rstd_b, _, (h, rstd_n, _) = K.residual_fwd(s["x"], s["br"], L, w_b=wb, p_drop=p, seed=7, next_w=w)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if which == "fwd":
                    K.residual_fwd(s["x"], s["br"], L, w_b=wb, p_drop=p, seed=7, next_w=w)
                else:
                    K.norm_residual_bwd(s["dy"], s["x"], rstd_n, None, w, K.NORM_RMS, L, s["dx"], dw, s["br"], w_b=wb, rstd_b=rstd_b, dw_b=dwb, p_drop=p, seed=7, dbias=dbias)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res.setdefault(f"{which}_p{p}", []).append(round(ts[len(ts) // 2], 1))
print(json.dumps(res))
