"""Attention micro-bench (1.4B shape): fwd / bwd wall time per call; UDM_DKV_WS=0 selects the single-role dK/dV kernel."""
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

B, H, L, D = 8, 16, 1280, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(4))
o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D)
fl = 4 * B * H * L * L * D
f = timeit(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D))
b = timeit(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D))
print(json.dumps(dict(fwd_ms=round(f, 4), fwd_tf=round(fl / f / 1e9, 1), bwd_ms=round(b, 4), bwd_tf_alg=round(2.5 * fl / b / 1e9, 1))))
