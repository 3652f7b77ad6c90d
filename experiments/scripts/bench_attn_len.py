"""Attention forward / backward rate against sequence length (B 8, H 16, D 128): shows the block-count quantisation of the 128-query grid
(L / 128 * B * H blocks on 512 slots: 1024 -> 2 rounds, 1280 -> 2.5, 1536 -> 3)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K


def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, H, D = 8, 16, 128
for L in (1024, 1152, 1280, 1536, 2048):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, do = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(4))
    o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D)
    fl = 4 * B * H * L * L * D
    f = timeit(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D))
    b = timeit(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D))
    print(json.dumps(dict(L=L, blocks=L // 128 * B * H, fwd_us=round(f * 1e3, 1), fwd_tf=round(fl / f / 1e9, 1), bwd_us=round(b * 1e3, 1), bwd_tf_executed=round(3.5 * fl / b / 1e9, 1))), flush=True)
