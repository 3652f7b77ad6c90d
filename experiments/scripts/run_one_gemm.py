"""Run ONE gemm variant on one shape a few times (for rocprofv3 --pmc).  usage: run_one_gemm.py <variant|-1 prod|-2 torch> M N K"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import _lib, kernels as K
v, m, n, k = (int(x) for x in sys.argv[1:5])
lib = _lib.load()
fn = _lib.load_experiments().udm_gemm_nt_bf16_variant
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 6 + [ctypes.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.rand(m, k, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
b = (torch.rand(n, k, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    if v == -1: K.gemm_nt(a, b, out=out)
    elif v == -2: torch.matmul(a, b.t(), out=out)
    else: fn(v, a.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, k, k, n, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
