"""qknorm_rope forward / backward at the headline shape (M = 10 240, d = 2048, D = 128), operands rotating over 3 buffer sets (not cache resident)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M, d, L, D = 10240, 2048, 1280, 128
g = torch.Generator(device="cuda").manual_seed(0)
sets = [dict(qkv=torch.randn(M, 3 * d, device="cuda", generator=g).bfloat16(), dqkr=torch.randn(M, 2 * d, device="cuda", generator=g).bfloat16(),
             dqkv=torch.empty(M, 3 * d, device="cuda", dtype=torch.bfloat16)) for _ in range(3)]
cos, sin = torch.randn(L, D // 2, device="cuda", generator=g), torch.randn(L, D // 2, device="cuda", generator=g)
gq, bq, gk, bk = (torch.randn(d, device="cuda", generator=g) for _ in range(4))
dg = torch.zeros(4 * d, device="cuda")
qs = K.attention_q_scale(D)
res = {}
for rnd in range(3):
    for which in ("fwd", "bwd"):
        ts = []
        for it in range(15):
            s = sets[it % 3]
            if which == "bwd":
                _, stats = K.qknorm_rope_fwd(s["qkv"], cos, sin, L, D, gq=gq, bq=bq, gk=gk, bk=bk, q_scale=qs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if which == "fwd":
                K.qknorm_rope_fwd(s["qkv"], cos, sin, L, D, gq=gq, bq=bq, gk=gk, bk=bk, q_scale=qs)
            else:
                K.qknorm_rope_bwd(s["dqkr"], s["qkv"], s["dqkv"], cos, sin, L, D, gq=gq, gk=gk, stats=stats, dgq=dg[:d], dbq=dg[d:2 * d], dgk=dg[2 * d:3 * d], dbk=dg[3 * d:], q_scale=qs)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res.setdefault(which, []).append(round(ts[len(ts) // 2], 1))
print(json.dumps(res))
