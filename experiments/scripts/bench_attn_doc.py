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

# fp8 forward (quantise pass and kernel timed separately) against the bf16 forward, config C shape and the packed config E shape
from unidisc_amd import _lib
for (B, L, docs) in ((8, 1280, 1), (2, 4608, 4)):
    H, D = 16, 128
    d = H * D
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B * L, d, generator=g).bfloat16().cuda() for _ in range(3))
    sid = (torch.arange(L) // (L // docs))[None].repeat(B, 1).cuda() if docs > 1 else None
    r = K.attention_doc_ranges(sid) if sid is not None else None
    f16 = t(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D, sid, r))
    o, lse, (q8, k8, v8t, scales) = K.attention_fwd_fp8_generic(q, k, v, B, L, H, D, sid, r, return_quantized=True)
    amax = torch.empty(3, dtype=torch.int32, device="cuda")
    p_ = lambda x: x.data_ptr() if x is not None else 0
    tq = t(lambda: _lib.call("udm_attention_quantize_fp8", p_(q), p_(k), p_(v), p_(q8), p_(k8), p_(v8t), p_(scales), p_(amax), B, H, L, D, d, d, d, K._s()))
    tf = t(lambda: _lib.call("udm_attention_fwd_fp8", p_(q8), p_(k8), p_(v8t), p_(scales), p_(o), p_(lse), p_(sid), p_(r), B, H, L, D, d, K._s()))
    print(f"B={B} L={L} docs={docs}: bf16 fwd {f16:7.1f} us | fp8 quantise {tq:6.1f} us + fp8 fwd {tf:7.1f} us", flush=True)
