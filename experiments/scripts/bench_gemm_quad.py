"""A/B of the one-wave-per-SIMD GEMM kernels (gemm_quad.hip) against the 8-wave kernels and torch.matmul (hipBLASLt yard-stick) on the step's shapes,
random operands, interleaved rounds in one process.  usage: bench_gemm_quad.py [tn|nt|all] [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidisc_amd import kernels as K

dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "all"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Mtok = 10240
TN = [("fc1 wgrad", 8192, 2048), ("fc2 wgrad", 2048, 8192), ("qkv wgrad", 6144, 2048)]
NT = [("qkv fwd", 6144, 2048), ("out / dgrads N=2048 K=2048", 2048, 2048), ("fc1 fwd", 8192, 2048), ("fc2 fwd", 2048, 8192), ("qkv dgrad", 2048, 6144)]


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def ab(name, flops, cands):
    res = {k: [] for k in cands}
    for r in range(rounds):
        for k, (pre, fn) in cands.items():
            pre()
            res[k].append(timeit(fn))
    K.gemm_set_quad(1)
    line = f"{name:34s}"
    for k, v in res.items():
        v.sort()
        med = v[len(v) // 2]
        line += f" | {k}: {med:7.1f} us {flops / med / 1e6:6.0f} TF (min {v[0]:.1f})"
    print(line, flush=True)


g = torch.Generator(device=dev).manual_seed(0)
if which in ("tn", "all"):
    for name, M, N in TN:
        a = (torch.randn(Mtok, M, device=dev, generator=g) * 0.5).bfloat16()
        b = (torch.randn(Mtok, N, device=dev, generator=g) * 0.5).bfloat16()
        out = torch.empty(M, N, device=dev)
        ab(f"TN {name} {M}x{N}x{Mtok}", 2.0 * M * N * Mtok, {
            "8-wave": (lambda: K.gemm_set_quad(0), lambda: K.gemm_tn(a, b, out, M=M, N=N)),
            "quad": (lambda: K.gemm_set_quad(2), lambda: K.gemm_tn(a, b, out, M=M, N=N)),
            "torch": (lambda: None, lambda: torch.matmul(a.t(), b)),
        })
if which in ("nt", "all"):
    for name, N, Kd in NT:
        a = (torch.randn(Mtok, Kd, device=dev, generator=g) * 0.5).bfloat16()
        b = (torch.randn(N, Kd, device=dev, generator=g) * 0.5).bfloat16()
        out = torch.empty(Mtok, N, device=dev, dtype=torch.bfloat16)
        ab(f"NT {name} {Mtok}x{N}x{Kd}", 2.0 * Mtok * N * Kd, {
            "8-wave": (lambda: K.gemm_set_quad(0), lambda: K.gemm_nt(a, b, out)),
            "quad": (lambda: K.gemm_set_quad(2), lambda: K.gemm_nt(a, b, out)),
            "torch": (lambda: None, lambda: torch.matmul(a, b.t())),
        })

if which in ("epi", "all"):
    for name, N, Kd, epi in [("fc1 fwd +bias+GELU", 8192, 2048, "gelu"), ("fc2 dgrad *GELU'", 8192, 2048, "dgelu"), ("fc2 fwd +bias", 2048, 8192, "bias")]:
        a = (torch.randn(Mtok, Kd, device=dev, generator=g) * 0.5).bfloat16()
        b = (torch.randn(N, Kd, device=dev, generator=g) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev, generator=g)
        out = torch.empty(Mtok, N, device=dev, dtype=torch.bfloat16)
        aux = (torch.rand(Mtok, N, device=dev, generator=g)).bfloat16()
        dbias = torch.zeros(N, device=dev)
        if epi == "gelu":
            fn = lambda: K.gemm_nt(a, b, out, epilogue=K.EPI_BIAS_GELU, bias=bias, aux=aux)
        elif epi == "dgelu":
            fn = lambda: K.gemm_nt(a, b, out, epilogue=K.EPI_DGELU, aux=aux, bias=dbias)
        else:
            fn = lambda: K.gemm_nt(a, b, out, epilogue=K.EPI_BIAS, bias=bias)
        ab(f"NT {name} {Mtok}x{N}x{Kd}", 2.0 * Mtok * N * Kd, {"8-wave": (lambda: K.gemm_set_quad(0), fn), "quad": (lambda: K.gemm_set_quad(2), fn)})
