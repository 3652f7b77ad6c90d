"""Phase experiment for the persistent NT GEMM (epilogues are chip-wide memory bursts because all 256 blocks are in lock-step).
One GEMM [M = 10240, N = 8192, K = 2048] as ONE launch (256 persistent blocks, 4 tiles of 320 rows each) against the same GEMM as TWO concurrent launches on two
streams: rows 0..5119 with 320-row tiles (128 blocks x 4 tiles) and rows 5120..10239 with 256-row tiles (128 blocks x 5 tiles) - equal work per block, epilogues
at different times.   python scripts/bench_gemm_phase.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M, N, Kd, R = 10240, 8192, 2048, 3
g = torch.Generator(device="cuda").manual_seed(0)
A = [(torch.rand(M, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16) for _ in range(R)]
B = (torch.rand(N, Kd, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
out = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
aux = [torch.rand(M, N, device="cuda", generator=g).to(torch.bfloat16) for _ in range(R)]
bias = torch.zeros(N, dtype=torch.float32, device="cuda")
K.gemm_set_quad(0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
H = M // 2


def single(i, kw, use_aux):
    K.gemm_set_cus(0), K.gemm_set_tile(320)
    K.gemm_nt(A[i % R], B, out=out[i % R], N=N, aux=aux[i % R] if use_aux else None, **kw)


def pair(i, kw, use_aux, second=256):
    main = torch.cuda.current_stream()
    s1.wait_stream(main), s2.wait_stream(main)
    a, o, x = A[i % R], out[i % R], aux[i % R]
    K.gemm_set_cus(128)
    with torch.cuda.stream(s1):
        K.gemm_set_tile(320)
        K.gemm_nt(a[:H], B, out=o[:H], N=N, aux=x[:H] if use_aux else None, **kw)
    with torch.cuda.stream(s2):
        K.gemm_set_tile(second)
        K.gemm_nt(a[H:], B, out=o[H:], N=N, aux=x[H:] if use_aux else None, **kw)
    main.wait_stream(s1), main.wait_stream(s2)


cases = [("plain", dict(epilogue=K.EPI_NONE), False), ("gelu'", dict(epilogue=K.EPI_DGELU), True), ("bias+gelu", dict(epilogue=K.EPI_BIAS_GELU, bias=bias), True)]
for rnd in range(2):
    for name, kw, use_aux in cases:
        line = f"{name:10s}"
        for label, fn in (("one launch", single), ("two launches, same phase", lambda i, kw, ua: pair(i, kw, ua, 320)), ("two phases", pair)):
            ts = []
            for rep in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                s.record()
                for i in range(12):
                    fn(i, kw, use_aux)
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / 12 * 1e3)
            line += f"   {label}: {min(ts):6.1f} us"
        print(line)
K.gemm_set_cus(0), K.gemm_set_tile(-1), K.gemm_set_quad(1)
