"""A/B of the attention kernels in the ENGINE's layout (q, k from the roped [M, 2d] buffer, v from qkv [M, 3d]): new (one wave per SIMD) vs 8-wave kernels,
bit-identity of O / LSE / dq|dk / dv and wall time per call."""
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
    d, M = H * D, B * L
    g = torch.Generator(device="cuda").manual_seed(L)
    qkr = torch.randn(M, 2 * d, device="cuda", generator=g).to(torch.bfloat16)
    qkv = torch.randn(M, 3 * d, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
    outs = []
    for mode in (0, 1):
        K.set_attention_w64(mode)
        o, lse = K.attention_fwd(qkr, qkv, B, L, H, D)
        dqkr = torch.zeros(M, 2 * d, dtype=torch.bfloat16, device="cuda")
        dqkv = torch.zeros(M, 3 * d, dtype=torch.bfloat16, device="cuda")
        K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D)
        outs.append((o, lse, dqkr, dqkv))
    torch.cuda.synchronize()
    rec = dict(B=B, H=H, L=L, equal=[bool(torch.equal(a, b)) for a, b in zip(*outs)])
    if bench:
        o, lse = outs[0][0], outs[0][1]
        dqkr, dqkv = torch.empty_like(outs[0][2]), torch.empty_like(outs[0][3])
        for mode, tag in ((0, "old"), (1, "new")):
            K.set_attention_w64(mode)
            rec[tag + "_fwd_us"] = round(timeit(lambda: K.attention_fwd(qkr, qkv, B, L, H, D)) * 1e3, 1)
            rec[tag + "_bwd_us"] = round(timeit(lambda: K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D)) * 1e3, 1)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    for (B, H, L) in [(1, 1, 128), (2, 3, 256), (3, 5, 640)]:
        run(B, H, L, bench=False)
    for (B, H, L) in [(8, 16, 1280), (2, 16, 4608)]:
        run(B, H, L)
