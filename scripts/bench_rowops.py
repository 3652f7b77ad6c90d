"""Micro-benchmark of the HBM-bound row kernels exactly as the 1.4 B step calls them (M = 10240, d = 2048, rms sandwich norms, LayerNorm qk-norm,
per-sample rotary tables, dropout on the MLP branch), with a plain device copy as the achievable-bandwidth yard-stick.  Diagnostic tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidisc_amd import kernels as K

DEV, BF16 = "cuda", torch.bfloat16
M, d, L, D, B = 10240, 2048, 1280, 128, 8


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def show(name, us, nbytes):
    print(f"{name:52s} {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s  ({nbytes / 1e6:.0f} MB)", flush=True)


g = torch.Generator(device=DEV).manual_seed(0)
x = torch.randn(M, d, device=DEV, generator=g)
x2 = torch.empty_like(x)
show("copy fp32 [M,d] (torch)", timeit(lambda: x2.copy_(x)), M * d * 8)
big = torch.randn(M, 4 * d, device=DEV, generator=g)
big2 = torch.empty_like(big)
show("copy fp32 [M,4d] (torch)", timeit(lambda: big2.copy_(big)), M * 4 * d * 8)
w = torch.ones(d, device=DEV) + 0.1 * torch.randn(d, device=DEV, generator=g)
br = (torch.randn(M, d, device=DEV, generator=g) * 0.5).to(BF16)
for p in (0.0, 0.1):
    show(f"residual_norm_fwd sandwich (attn branch) p={p}", timeit(lambda: K.residual_fwd(x, br, L, w_b=w, norm_type=0, next_w=w)), M * d * 12)
    show(f"residual_norm_fwd sandwich + dropout p={p}", timeit(lambda: K.residual_fwd(x, br, L, w_b=w, norm_type=0, p_drop=p, seed=5, next_w=w)), M * d * 12)
    xo, rb, _, (h, rn_, _) = K.residual_fwd(x, br, L, w_b=w, norm_type=0, p_drop=p, seed=5, next_w=w)
    dw = torch.zeros(d, device=DEV)
    show(f"residual_bwd sandwich p={p}", timeit(lambda: K.residual_bwd(x, br, L, w_b=w, rstd=rb, norm_type=0, dw_b=dw, p_drop=p, seed=5)), M * d * 8)
y, rstd, _ = K.norm_fwd(x, w, 0, L)
show("norm_fwd", timeit(lambda: K.norm_fwd(x, w, 0, L)), M * d * 6)
dx = torch.zeros_like(x)
dw = torch.zeros(d, device=DEV)
show("norm_bwd accumulate", timeit(lambda: K.norm_bwd(y, x, rstd, None, w, 0, L, dx, dw, accumulate=True)), M * d * 14)
qkv = (torch.randn(M, 3 * d, device=DEV, generator=g) * 0.5).to(BF16)
ang = torch.randn(B, L, D // 2, device=DEV, generator=g)
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
bq = 0.1 * torch.randn(d, device=DEV, generator=g)
show("qknorm_rope_fwd (per-sample tables)", timeit(lambda: K.qknorm_rope_fwd(qkv, cos, sin, L, D, gq=w, bq=bq, gk=w, bk=bq)), M * d * 8)
qkr, st = K.qknorm_rope_fwd(qkv, cos, sin, L, D, gq=w, bq=bq, gk=w, bk=bq)
dqkr = (torch.randn(M, 2 * d, device=DEV, generator=g) * 0.5).to(BF16)
dqkv = torch.empty(M, 3 * d, dtype=BF16, device=DEV)
G = torch.zeros(4 * d, device=DEV)
show("qknorm_rope_bwd", timeit(lambda: K.qknorm_rope_bwd(dqkr, qkv, dqkv, cos, sin, L, D, gq=w, gk=w, stats=st, dgq=G[:d], dbq=G[d:2 * d], dgk=G[2 * d:3 * d], dbk=G[3 * d:])),
     M * d * 12)
wt = torch.randn(8192, 2048, device=DEV, generator=g)
o, ot = torch.empty(8192, 2048, dtype=BF16, device=DEV), torch.empty(2048, 8192, dtype=BF16, device=DEV)
show("cast_transpose 8192x2048", timeit(lambda: K.cast_transpose(wt, o, ot)), wt.numel() * 8)
wt2 = torch.randn(2048, 2048, device=DEV, generator=g)
o2, ot2 = torch.empty(2048, 2048, dtype=BF16, device=DEV), torch.empty(2048, 2048, dtype=BF16, device=DEV)
show("cast_transpose 2048x2048", timeit(lambda: K.cast_transpose(wt2, o2, ot2)), wt2.numel() * 8)
du = (torch.randn(M, d, device=DEV, generator=g)).to(BF16)
cs = torch.zeros(d, device=DEV)
show("colsum [M,d] bf16 (mlp.2 bias grad)", timeit(lambda: K.colsum(du, cs)), M * d * 2)
