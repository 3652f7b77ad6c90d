"""Attention forward / backward time against the number of 128-query blocks (B swept at H = 16, L = 1280, D = 128): how much of a launch is the
partial last round of the 512 block slots (two 4-wave workgroups per CU), and what a lone block per CU costs against a co-resident pair."""
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
    return e0.elapsed_time(e1) / n * 1e3

H, L, D = 16, 1280, 128
g = torch.Generator(device="cuda").manual_seed(0)
for B in [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 16]:
    q, k, v, do = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(4))
    o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D)
    f = timeit(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D))
    b = timeit(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D))
    blocks = B * H * L // 128
    print(json.dumps(dict(B=B, blocks=blocks, rounds=round(blocks / 512, 2), fwd_us=round(f, 1), bwd_us=round(b, 1), fwd_us_per_round=round(f / (blocks / 512), 1))), flush=True)
