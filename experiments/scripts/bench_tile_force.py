#!/usr/bin/env python3
"""A/B of the GEMM tile choice on a whole step: python scripts/bench_tile_force.py <workload> <tile|-1> [steps]."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from unidisc_amd import kernels as K
wl, tile = sys.argv[1], int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.manual_seed(42)
cfg, diff = bench.build(wl, torch.device("cuda"), 0.1)
B = bench.WORKLOADS[wl]["batch"]
batch = {k: v.cuda() for k, v in bench.synthetic_batch(wl, B, 42).items()}
K.gemm_set_tile(tile)
def step(i):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i); out.loss.backward()
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps): step(3 + i)
torch.cuda.synchronize(); print(f"{wl} tile={tile}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step")
