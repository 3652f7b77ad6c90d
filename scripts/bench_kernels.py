"""Kernel micro-benchmarks on one MI355X: achieved TFLOP/s (GEMM, attention) and GB/s (row kernels) per launch.
Diagnostic tool for kernel work; the judged benchmark is bench.py."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K  # noqa: E402

DEV = "cuda"
BF16 = torch.bfloat16


def timeit(fn, warmup=3, iters=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    res = {}
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: (torch.randn(*s, device=DEV, generator=g) * 0.5).to(BF16)
    M = 10240
    for name, (m, n, k, f32) in {
        "qkv_fwd": (M, 6144, 2048, False), "out_fwd": (M, 2048, 2048, False), "fc1_fwd": (M, 8192, 2048, False), "fc2_fwd": (M, 2048, 8192, False),
        "head_fwd": (M, 48385, 2048, False), "fc1_wgrad": (8192, 2048, M, True), "qkv_wgrad": (6144, 2048, M, True), "sq4096": (4096, 4096, 4096, False),
    }.items():
        a, b = rn(m, k), rn(n, k)
        ldc = (n + 127) // 128 * 128
        out = torch.empty((m, ldc), dtype=torch.float32 if f32 else BF16, device=DEV)
        ms = timeit(lambda: K.gemm_nt(a, b, out=out, N=n))
        ms_t = timeit(lambda: torch.matmul(a, b.t()))
        res[f"gemm/{name}"] = dict(ms=round(ms, 4), tflops=round(2 * m * n * k / ms / 1e9, 1), torch_matmul_tflops=round(2 * m * n * k / ms_t / 1e9, 1))
        del a, b, out
    for name, (kc, m, n) in {"fc1_wgrad_tn": (M, 8192, 2048), "fc2_wgrad_tn": (M, 2048, 8192), "qkv_wgrad_tn": (M, 6144, 2048), "out_wgrad_tn": (M, 2048, 2048)}.items():
        a, b = rn(kc, m), rn(kc, n)
        out = torch.empty((m, n), dtype=torch.float32, device=DEV)
        ms = timeit(lambda: K.gemm_tn(a, b, out, beta=1.0))
        res[f"gemm/{name}"] = dict(ms=round(ms, 4), tflops=round(2 * m * n * kc / ms / 1e9, 1))
        if name == "out_wgrad_tn":
            ms = timeit(lambda: K.gemm_tn_splitk(a, b, out, beta=0.0))
            res["gemm/out_wgrad_tn_splitk_ws"] = dict(ms=round(ms, 4), tflops=round(2 * m * n * kc / ms / 1e9, 1))
        del a, b, out
    # epilogue variants
    a, b, bias = rn(M, 2048), rn(8192, 2048), torch.randn(8192, device=DEV)
    aux = torch.empty((M, 8192), dtype=BF16, device=DEV)
    ms = timeit(lambda: K.gemm_nt(a, b, epilogue=K.EPI_BIAS_GELU, bias=bias, aux=aux))
    res["gemm/fc1_bias_gelu"] = dict(ms=round(ms, 4), tflops=round(2 * M * 8192 * 2048 / ms / 1e9, 1))
    del a, b, aux
    # transposes
    x = rn(M, 8192)
    ms = timeit(lambda: K.transpose(x))
    res["transpose/10240x8192"] = dict(ms=round(ms, 4), gbps=round(2 * x.numel() * 2 / ms / 1e6, 1))
    w = torch.randn(8192, 2048, device=DEV)
    o, ot = torch.empty((8192, 2048), dtype=BF16, device=DEV), torch.empty((2048, 8192), dtype=BF16, device=DEV)
    ms = timeit(lambda: K.cast_transpose(w, o, ot))
    res["cast_transpose/8192x2048"] = dict(ms=round(ms, 4), gbps=round(w.numel() * 8 / ms / 1e6, 1))
    del x, w, o, ot
    # attention
    for (B, H, L, D) in [(8, 16, 1280, 128), (16, 12, 384, 64)]:
        d = H * D
        q, k, v, do = (rn(B * L, d) for _ in range(4))
        for tr in (True, False):
            K.set_tr_read(tr)
            ms = timeit(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D))
            o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D)
            msb = timeit(lambda: K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D))
            fl = 4 * B * H * L * L * D
            res[f"attn/B{B}H{H}L{L}D{D}/tr{int(tr)}"] = dict(fwd_ms=round(ms, 4), fwd_tflops=round(fl / ms / 1e9, 1), bwd_ms=round(msb, 4),
                                                            bwd_tflops_algorithmic=round(2.5 * fl / msb / 1e9, 1))
        K.set_tr_read(True)
        del q, k, v, do
    # row kernels at M=10240, d=2048
    d, Lr = 2048, 1280
    x = torch.randn(M, d, device=DEV)
    w = torch.ones(d, device=DEV)
    ms = timeit(lambda: K.norm_fwd(x, w, 0, Lr))
    res["norm_fwd"] = dict(ms=round(ms, 4), gbps=round(M * d * 6 / ms / 1e6, 1))
    y, rstd, _ = K.norm_fwd(x, w, 0, Lr)
    dx, dw = torch.zeros_like(x), torch.zeros(d, device=DEV)
    ms = timeit(lambda: K.norm_bwd(y, x, rstd, None, w, 0, Lr, dx, dw))
    res["norm_bwd"] = dict(ms=round(ms, 4), gbps=round(M * d * 14 / ms / 1e6, 1))
    ms = timeit(lambda: K.residual_fwd(x, y, Lr, w_b=w))
    res["residual_fwd_sandwich"] = dict(ms=round(ms, 4), gbps=round(M * d * 10 / ms / 1e6, 1))
    xo, rb, _ = K.residual_fwd(x, y, Lr, w_b=w)
    ms = timeit(lambda: K.residual_bwd(dx, y, Lr, w_b=w, rstd=rb, dw_b=dw))
    res["residual_bwd_sandwich"] = dict(ms=round(ms, 4), gbps=round(M * d * 8 / ms / 1e6, 1))
    qkv = rn(M, 3 * d)
    ang = torch.randn(Lr, 64, device=DEV)
    cos, sin = ang.cos(), ang.sin()
    ms = timeit(lambda: K.qknorm_rope_fwd(qkv, cos, sin, Lr, 128, gq=w, bq=w, gk=w, bk=w))
    res["qknorm_rope_fwd"] = dict(ms=round(ms, 4), gbps=round(M * d * 8 / ms / 1e6, 1))
    # CE
    V, Vp = 48385, 48512
    logits = rn(M, Vp)
    x0 = torch.randint(0, 32000, (M,), device=DEV)
    xt = x0.clone()
    xt[::2] = 32000
    mod = torch.zeros(M, dtype=torch.int64, device=DEV)
    ms = timeit(lambda: K.subs_ce_fwd(logits, x0, xt, mod, V, 32001, 32000, True))
    res["subs_ce_fwd(50% masked)"] = dict(ms=round(ms, 4), gbps=round(M / 2 * 32001 * 2 / ms / 1e6, 1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
