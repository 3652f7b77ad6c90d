"""Does operand data (bit toggling -> power -> clocks) change GEMM throughput?  zeros vs small-range vs full-range random operands."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import _lib, kernels as K
lib = _lib.load()
fn = _lib.load_experiments().udm_gemm_nt_bf16_variant
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 6 + [ctypes.c_void_p]

def timeit(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

m, n, k = 8192, 2048, 10240
for kind in ("zeros", "randn", "rand01"):
    if kind == "zeros": a, b = torch.zeros(m, k, device="cuda"), torch.zeros(n, k, device="cuda")
    elif kind == "ones": a, b = torch.ones(m, k, device="cuda"), torch.ones(n, k, device="cuda")
    elif kind == "rand01": a, b = torch.rand(m, k, device="cuda") - 0.5, torch.rand(n, k, device="cuda") - 0.5
    elif kind == "randn": a, b = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda")
    else: a, b = torch.randn(m, k, device="cuda") * 100, torch.randn(n, k, device="cuda") * 100
    a, b = a.to(torch.bfloat16), b.to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    r = {}
    r["prod"] = timeit(lambda: K.gemm_nt(a, b, out=out))
    for v in (31, 32, 33, 34, 35, 36):
        r[f"quad{v}"] = timeit(lambda: fn(v, a.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, k, k, n, torch.cuda.current_stream().cuda_stream))
    r["torch"] = timeit(lambda: torch.matmul(a, b.t(), out=out))
    print(kind, json.dumps({x: round(2 * m * n * k / t / 1e9) for x, t in r.items()}))
