"""Timing-only ablations of the dK / dV program (WRONG results; library built with `make -C unidisc_amd/csrc regen all UDM_DKV64_ABL="1 2 4 8"`), one process per
variant: UDM_ATTN_DKV64_ABL = 0 (the product), 1 = no softmax VALU, 2 = no fragment reads, 4 = no refills / waits / barriers, 8 = no MFMAs.  Prints the whole backward
(dQ pass + dK/dV pass) in microseconds; the dQ pass is the same in every variant."""
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
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(json.dumps({"abl": int(os.environ.get("UDM_ATTN_DKV64_ABL", "0")), "bwd_us_median": round(ts[len(ts) // 2], 1), "bwd_us_min": round(ts[0], 1)}))
