"""Attention backward (dQ pass + dK/dV pass) at the headline shape, engine layout, pre-scaled q; cold variant: a 512 MB fill between calls.
A-B by environment (one process each): UDM_DQ_PRE=0 - the dQ kernel with the scale / lse / delta arithmetic behind the MFMAs."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

B, H, L, D = 8, 16, 1280, 128
d, M = H * D, B * L
g = torch.Generator(device="cuda").manual_seed(0)
qkr = torch.randn(M, 2 * d, device="cuda", generator=g)
qkr[:, :d] *= K.attention_q_scale(D)
qkr = qkr.to(torch.bfloat16)
qkv = torch.randn(M, 3 * d, device="cuda", generator=g).to(torch.bfloat16)
do = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
o, lse = K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True)
dqkr, dqkv = torch.empty_like(qkr), torch.empty_like(qkv)
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
res = {}
for cold in (False, True, False, True):
    ts = []
    for _ in range(20):
        if cold:
            junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    res.setdefault("cold" if cold else "warm", []).append(round(ts[len(ts) // 2], 1))
res["dq_checksum"] = float(dqkr[:, :d].float().abs().sum())
print(json.dumps(res))
