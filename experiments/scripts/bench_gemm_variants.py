"""A/B of GEMM tile/pipeline variants (gemm_exp.hip) against the production kernel and torch.matmul (hipBLASLt yard-stick).
Interleaved rounds in ONE process, random [-0.5,0.5]-ish operands, correctness checked against torch fp32."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import _lib, kernels as K  # noqa: E402

NAMES = {0: "128x128 w2x2 s2", 1: "128x128 w2x2 s3", 2: "128x128 w2x2 s4", 3: "256x128 w4x2 s2", 4: "256x128 w4x2 s3", 5: "256x128 w2x2 s3",
         6: "256x256 w2x4 s2", 7: "256x256 w4x4 s2", 8: "256x256 w4x2 s2", 9: "128x256 w2x4 s3", 10: "stg 256x256 w2x4 p1", 11: "stg 256x256 w2x4 p2",
         12: "stg 256x128 w4x2 p2", 13: "stg 128x256 w2x4 p2", 14: "stg 256x256 w4x2 p1", 15: "stg 128x128 w2x4 p2", 16: "stg 320x256 w2x4 p1", 17: "stg 256x320 w2x4 p1", 20: "320 p1 mf", 21: "320 p1 mf+prio", 22: "320 p2 mf", 23: "320 p2 mf+prio", 24: "320 p1 plain", 30: "quad 256x256 w2x2", 31: "quad + hints", 39: "stg256 dma", 40: "stg256 regstage", 41: "stg320 regstage", 42: "stg320 global_lds", 43: "stg320 buffer_lds"}
lib = _lib.load()
fn = _lib.load_experiments().udm_gemm_nt_bf16_variant
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 6 + [ctypes.c_void_p]
fn.restype = ctypes.c_int


def variant(v, a, b, out):
    rc = fn(v, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], b.shape[0], a.shape[1], a.stride(0), b.stride(0), out.stride(0),
            torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError(_lib.load_experiments().udm_last_error().decode())


def timeit(fn_, iters=8):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn_()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else sorted(NAMES)
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = {"qkv_fwd": (10240, 6144, 2048), "fc2_fwd": (10240, 2048, 8192), "out_fwd": (10240, 2048, 2048), "fc1_wgrad": (8192, 2048, 10240), "sq4096": (4096, 4096, 4096),
              "head_fwd": (10240, 48384, 2048), "fc1_dgrad": (10240, 2048, 8192), "qkv_dgrad": (10240, 2048, 6144), "out_wgrad": (2048, 2048, 10240), "qkv_wgrad": (6144, 2048, 10240)}
    res = {}
    for sname, (m, n, k) in shapes.items():
        a = (torch.rand(m, k, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
        b = (torch.rand(n, k, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
        out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        ref = None
        if m * n <= 10240 * 6144:
            ref = (a[:512].float() @ b.float().t())
        cands = {"prod": lambda: K.gemm_nt(a, b, out=out), "torch": lambda: torch.matmul(a, b.t(), out=out)}
        for v in variants:
            cands[NAMES[v]] = (lambda v=v: variant(v, a, b, out))
        ok = {}
        for name, f in cands.items():
            out.zero_()
            f()
            torch.cuda.synchronize()
            if ref is not None:
                err = ((out[:512].float() - ref).norm() / ref.norm()).item()
                ok[name] = err < 5e-3
        best = {name: 1e9 for name in cands}
        for rnd in range(4):
            for name, f in cands.items():
                best[name] = min(best[name], timeit(f))
        res[sname] = {name: dict(tflops=round(2 * m * n * k / t / 1e9), ok=ok.get(name)) for name, t in best.items()}
        print(sname, json.dumps(res[sname]), flush=True)
        del a, b, out


if __name__ == "__main__":
    main()
