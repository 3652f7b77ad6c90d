"""How long the HOST needs to enqueue one training step (no synchronisation inside the loop) beside the GPU's step time, and where the host time goes
(cProfile).  When enqueue time ~ step time the workload is launch-bound and faster kernels do not show.   python scripts/host_profile.py [workload] [steps]"""
import cProfile
import importlib.util
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

workload = sys.argv[1] if len(sys.argv) > 1 else "unidisc-s-l384"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg, diff = bench.build(workload, dev, 0.1)
w = bench.WORKLOADS[workload]
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(workload, w["batch"], 42).items()}


def step(i):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i)
    out.loss.backward()
    return out


for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(3 + i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{workload}: host enqueue {1e3 * t_enq / steps:.2f} ms/step, step (synchronised at the end) {1e3 * t_all / steps:.2f} ms/step")

# host-only cost: same loop, synchronising after every step so the enqueue time is not hidden behind the GPU
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(steps):
    step(3 + steps + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
