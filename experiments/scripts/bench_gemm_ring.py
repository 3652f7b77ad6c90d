"""Ring / quad / production / hipBLASLt (torch.matmul) on the shapes the 256 x 256 tile quantises well on, zeros and random operands;
correctness of the experimental variants against fp32 on a row block."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import _lib, kernels as K
lib = _lib.load()
fn = _lib.load_experiments().udm_gemm_nt_bf16_variant
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 6 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
VARS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [30, 50]

def timeit(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def run(v, a, b, out):
    m, k = a.shape; n = b.shape[0]
    rc = fn(v, a.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, k, k, n, torch.cuda.current_stream().cuda_stream)
    if rc: raise RuntimeError(_lib.load_experiments().udm_last_error().decode())

shapes = {"fc1_wgrad": (8192, 2048, 10240), "fc1_fwd": (10240, 8192, 2048), "out_fwd": (10240, 2048, 2048), "qkv_fwd": (10240, 6144, 2048), "k96": (512, 512, 96), "k224": (768, 512, 224)}
for sname, (m, n, k) in shapes.items():
    for kind in ("randn", "zeros"):
        if kind == "zeros": a, b = torch.zeros(m, k, device="cuda"), torch.zeros(n, k, device="cuda")
        else: a, b = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda")
        a, b = a.to(torch.bfloat16), b.to(torch.bfloat16)
        out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        r = {}
        if kind == "randn":
            ref = a[:512].float() @ b.float().t()
            for v in VARS:
                out.zero_(); run(v, a, b, out); torch.cuda.synchronize()
                err = (out[:512].float() - ref).abs().max().item() / ref.abs().max().item()
                r[f"err{v}"] = round(err, 5)
                full = (out.float() - (a.float() @ b.float().t())).abs().max().item() if m * n <= 768 * 512 else None
                if full is not None: r[f"fullerr{v}"] = round(full, 4)
        if sname.startswith("k"):
            print(sname, kind, json.dumps(r)); continue
        r["prod"] = timeit(lambda: K.gemm_nt(a, b, out=out))
        for v in VARS: r[f"v{v}"] = timeit(lambda: run(v, a, b, out))
        r["torch"] = timeit(lambda: torch.matmul(a, b.t(), out=out))
        print(sname, kind, json.dumps({x: (round(2 * m * n * k / t / 1e9) if not x.startswith("err") else t) for x, t in r.items()}), flush=True)
