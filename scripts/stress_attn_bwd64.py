"""Stress test of the generated attention backward under memory contention: N iterations at the headline shape with fresh random operands, a second stream hammering HBM
(copies of 1 GB) while the kernels run; dV must equal the 8-wave kernel's bit for bit (it does not depend on delta), dK and dQ to the fp32 rounding of delta (the generated dQ pass sums dO * O with
v_dot2c, the 8-wave kernel element by element), nothing may be non-finite.  Catches ordering
assumptions (LDS-DMA vs fragment reads, counted vmcnt across stores) that only fail when latencies move."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

N = int(os.environ.get("N", "150"))
B, H, L, D = 8, 16, 1280, 128
d, M = H * D, B * L
dev = "cuda"
side = torch.cuda.Stream()
junk_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
junk_b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
worst_dq = 0.0
for it in range(N):
    qkr = torch.randn(M, 2 * d, device=dev, generator=g) * (0.5 + (it % 5))
    qkr[:, :d] *= K.attention_q_scale(D)
    qkr = qkr.to(torch.bfloat16)
    qkv = torch.randn(M, 3 * d, device=dev, generator=g).to(torch.bfloat16)
    do = torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16)
    o, lse = K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True)
    outs = []
    for flag in (1, 0):
        K.set_attention_dq64(flag)
        K.set_attention_dkv64(flag)
        dqkr, dqkv = torch.empty_like(qkr), torch.empty_like(qkv)
        torch.cuda.synchronize()
        if flag:      # contention only while the generated programs run
            with torch.cuda.stream(side):
                for _ in range(3):
                    junk_b.copy_(junk_a)
        K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
        torch.cuda.synchronize()
        outs.append((dqkr, dqkv))
    (a, av), (b_, bv) = outs
    ok = torch.equal(av[:, 2 * d:], bv[:, 2 * d:]) and bool(torch.isfinite(a.float()).all()) and bool(torch.isfinite(av[:, 2 * d:].float()).all())
    rel = max(float((a[:, :d].float() - b_[:, :d].float()).norm() / b_[:, :d].float().norm()), float((a[:, d:].float() - b_[:, d:].float()).norm() / b_[:, d:].float().norm()))
    worst_dq = max(worst_dq, rel)
    if not ok or rel > 1e-3:
        bad += 1
        print("MISMATCH at iteration", it, ok, rel)
K.set_attention_dq64(1)
K.set_attention_dkv64(1)
print(json.dumps(dict(iterations=N, mismatches=bad, worst_dq_dk_rel_vs_8wave=worst_dq)))
sys.exit(1 if bad else 0)
