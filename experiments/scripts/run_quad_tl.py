"""Quad GEMM timeline (variant 37): prints per-K-tile cycle stamps of block 0."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import _lib  # (exp library: experiments/exp_lib.py)
lib = _lib.load()
fn = _lib.load_experiments().udm_gemm_nt_bf16_variant
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 6 + [ctypes.c_void_p]
m, n, k = 10240, 2048, 8192
a = torch.randn(m, k, device="cuda").to(torch.bfloat16); b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
import sys as _s
V = int(_s.argv[1]) if len(_s.argv) > 1 else 37
for _ in range(4): fn(V, a.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, k, k, n, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
