"""A/B of the attention forward at the headline shape (B8 H16 L1280 D128, engine layout): 64-queries-per-wave kernel vs the 8-wave kernel,
interleaved rounds in one process; cold variant: a 256 MB fill between calls (the in-step situation: inputs not L2-resident)."""
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
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
fl = 4 * B * H * L * L * D


def timed(flag, cold, n=20):
    K.set_attention_fwd64(flag)
    ts = []
    for _ in range(n):
        if cold:
            junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


res = {}
for rnd in range(3):
    for flag in (1, 2, 0):
        for cold in (False, True):
            med, mn = timed(flag, cold)
            res.setdefault({1: "fwd64", 2: "fwd64_whole_blocks", 0: "8wave"}[flag] + ("_cold" if cold else "_warm"), []).append((round(med, 1), round(mn, 1)))
K.set_attention_fwd64(True)
out = {k: dict(median_us=v, tf=round(fl / (min(x[0] for x in v) * 1e-6) / 1e12, 1)) for k, v in res.items()}
print(json.dumps(out))
