#!/usr/bin/env python3
"""Attention kernels on packed samples (config E shape: B=1..2, H=16, D=128, L=4608 = 4 documents of 1152): time with and without tile skipping."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for B in (1, 2):
    H, D, L, docs = 16, 128, 4608, 4
    d = H * D
    g = torch.Generator().manual_seed(0)
    q, k, v, do = (torch.randn(B * L, d, generator=g).bfloat16().cuda() for _ in range(4))
    sid = (torch.arange(L) // (L // docs))[None].repeat(B, 1).cuda()
    r = K.attention_doc_ranges(sid)
    for name, rr, ss in (("no mask", None, None), ("mask, all tiles", None, sid), ("mask, tile skipping", r, sid)):
        o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D, ss, rr)
        f = t(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D, ss, rr))
        b = t(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D, ss, rr))
        print(f"B={B} {name:22s} fwd {f:8.1f} us  bwd {b:8.1f} us", flush=True)
    print(f"B={B} doc_ranges kernel {t(lambda: K.attention_doc_ranges(sid)):.1f} us")
