"""A/B of the backward attention kernels at head dim 128 (no document mask): dQ of attention_dq_w64.hip vs the 8-wave dQ kernel - bit-identity and wall time."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K


def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(B, H, L, D=128, bench=True):
    g = torch.Generator(device="cuda").manual_seed(L)
    q, k, v, do = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(4))
    K.set_attention_w64(0)
    o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D)
    r0 = K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D)
    K.set_attention_w64(1)
    r1 = K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D)
    torch.cuda.synchronize()
    rec = dict(B=B, H=H, L=L, dq_equal=bool(torch.equal(r0[0], r1[0])), dk_equal=bool(torch.equal(r0[1], r1[1])), dv_equal=bool(torch.equal(r0[2], r1[2])),
               dq_maxdiff=float((r0[0].float() - r1[0].float()).abs().max()), dq_absmax=float(r0[0].float().abs().max()), finite=bool(torch.isfinite(r1[0].float()).all()))
    if bench:
        fl = 10 * B * H * L * L * D
        K.set_attention_w64(0)
        t0 = timeit(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D))
        K.set_attention_w64(1)
        t1 = timeit(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D))
        rec.update(old_us=round(t0 * 1e3, 1), new_us=round(t1 * 1e3, 1), old_tf_alg=round(fl / t0 / 1e9), new_tf_alg=round(fl / t1 / 1e9))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    for (B, H, L) in [(1, 1, 128), (2, 3, 256), (1, 2, 384), (3, 1, 1152), (3, 5, 640)]:
        run(B, H, L, bench=False)
    for (B, H, L) in [(8, 16, 1280), (8, 16, 1024), (8, 16, 2048), (2, 16, 4608)]:
        run(B, H, L)
