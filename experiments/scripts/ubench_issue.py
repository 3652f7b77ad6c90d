"""Issue-cost microbenchmark (csrc/ubench.hip): shader cycles per k-step of 16 independent 32x32x16 MFMAs (512 matrix-pipe cycles) with N
memory instructions of one kind spread through it.  One wave per SIMD, 4 waves per block.
  part 1: idle memory system (1 block): what the instruction itself costs the issuing wave
  part 2: 256 blocks streaming through source windows of different footprint / sharing: what the memory system sustains per CU"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import _lib
lib = _lib.load()
fn = _lib.load_experiments().udm_ubench_issue
fn.argtypes = [ctypes.c_int] * 7 + [ctypes.c_void_p] * 4
fn.restype = ctypes.c_int
src = torch.zeros(256 * 4 * 65536 + (1 << 20), dtype=torch.uint8, device="cuda")
out = torch.zeros(4, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, dtype=torch.float32, device="cuda")
KINDS = {0: "none", 1: "global_load_lds b128", 2: "buffer_load lds b128", 3: "buffer_load lds b32", 4: "global_load_dwordx4 -> VGPR", 5: "ds_read_b128", 6: "s_nop 15", 7: "buffer lds b128, wave 0 only", 8: "4 x v_fma_f32", 9: "2 x v_exp_f32"}
iters = 200

def run(mode, blocks, stride, win, bstride, wstride):
    rc = fn(mode, blocks, iters, stride, win - 1, bstride, wstride, src.data_ptr(), out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc: raise RuntimeError(_lib.load_experiments().udm_last_error().decode())
    torch.cuda.synchronize()
    return [round(x / iters) for x in out.tolist()]

part = sys.argv[1] if len(sys.argv) > 1 else "12"
if "1" in part:
    for mode in [0, 1000, 102, 104, 108, 116, 202, 204, 208, 216, 304, 308, 316, 404, 408, 504, 508, 516, 604, 608, 804, 808, 816, 904, 908, 916, 1204, 1208, 1216, 1508, 1516]:
        kind, nops, m16 = (mode % 1000) // 100, mode % 100, mode // 1000
        c = run(mode, 1, 1024, 65536, 4 * 65536, 65536)
        print(f"idle  mfma {'16x16x32' if m16 else '32x32x16'} {KINDS[kind]:30s} x{nops:2d}: cycles/k-step {c[0]}", flush=True)
if "2" in part:
    # footprints: (name, window per wave, block stride, wave stride)
    cases = [("all blocks share 256 KiB (L2 hits, hot lines)", 65536, 0, 65536),
             ("64 KiB per block, 16 MiB total (L2 resident)", 16384, 65536, 16384),
             ("256 KiB per block, 64 MiB total (beyond L2)", 65536, 4 * 65536, 65536),
             ("8 blocks share a window (GEMM-like panel reuse), 8 MiB total", 65536, 0, 65536)]
    for name, win, bs, ws in cases:
        for mode in [202, 204, 208, 216]:
            nops = mode % 100
            c = run(mode, 256, 1024, win, bs, ws)
            bpc = 4 * nops * 1024 / c[0]
            print(f"busy  {name:62s} buffer lds b128 x{nops:2d}: cycles/k-step {c[0]:5d}  -> {bpc:5.1f} B/clk/CU", flush=True)
