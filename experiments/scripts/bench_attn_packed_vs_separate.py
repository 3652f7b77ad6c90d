"""The packed-sample attention (B = 2, L = 4608 = 4 documents of 1152, document mask + tile skipping) against the SAME work as separate samples
(B = 8, L = 1152, no mask): what the sample-id path costs per key tile.   python scripts/bench_attn_packed_vs_separate.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


H, D = 16, 128
d = H * D
g = torch.Generator().manual_seed(0)
for name, B, L, docs in (("separate", 8, 1152, 1), ("separate+ids", 8, 1152, -1), ("packed", 2, 4608, 4), ("packed B=8", 8, 4608, 4), ("headline", 8, 1280, 1)):
    q, k, v, do = (torch.randn(B * L, d, generator=g).bfloat16().cuda() for _ in range(4))
    force_ids, docs = docs < 0, abs(docs)   # "+ids": one document per row, but through the sample-id kernels
    sid = (torch.arange(L) // (L // docs))[None].repeat(B, 1).cuda() if (docs > 1 or force_ids) else None
    r = K.attention_doc_ranges(sid) if sid is not None else None
    o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D, sid, r)
    f = t(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D, sid, r))
    b = t(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D, sid, r))
    fl = 4.0 * B * H * docs * (L // docs) ** 2 * D
    print(f"{name:13s} B={B} L={L}: fwd {f:7.1f} us ({fl / f / 1e6:5.0f} TF)  bwd {b:7.1f} us ({3.5 * fl / b / 1e6:5.0f} TF executed)")
