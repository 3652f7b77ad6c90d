"""A/B of the attention backward's two passes (dq64_dkv64 = both generated programs; dkv64 = the generated dK / dV pass behind the 8-wave dQ kernel; ws8 = both 8-wave kernels).
A/B of the dK / dV pass at the headline shape (B 8, H 16, L 1280, D 128, engine layout, pre-scaled q): the generated one-wave-per-SIMD kernel
(csrc/attention_dkv64.hip) against the wave-specialised 8-wave kernel, alternating in ONE process, warm and cold (a 512 MB fill between calls).  The time is
the whole udm_attention_bwd call (dQ pass + dK/dV pass): the dQ pass is the same kernel on both sides, so the difference is the dK/dV pass.  UDM_SHAPE=B,H,L."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

B, H, L = (int(x) for x in os.environ.get("UDM_SHAPE", "8,16,1280").split(","))
D = 128
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
res = {"shape": [B, H, L, D]}


def timed(flag, cold, n=20):
    K.set_attention_dkv64(flag if flag < 3 else 1)
    K.set_attention_dq64(1 if flag == 3 else 0)
    ts = []
    for _ in range(n):
        if cold:
            junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


for rep in range(3):
    for cold in (False, True):
        for name, flag in (("dq64_dkv64", 3), ("dkv64", 1), ("ws8", 0), ("dkv64_unbalanced", 2)):
            res.setdefault(("cold_" if cold else "warm_") + name + "_us", []).append(timed(flag, cold))
K.set_attention_dkv64(1)
K.set_attention_dq64(0)
K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
a = (dqkr[:, d:].float().clone(), dqkv[:, 2 * d:].float().clone())
K.set_attention_dkv64(0)
K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
res["dk_rel_vs_ws8"] = float((a[0] - dqkr[:, d:].float()).norm() / dqkr[:, d:].float().norm())
res["dv_rel_vs_ws8"] = float((a[1] - dqkv[:, 2 * d:].float()).norm() / dqkv[:, 2 * d:].float().norm())
K.set_attention_dkv64(1)
K.set_attention_dq64(1)
print(json.dumps(res))
