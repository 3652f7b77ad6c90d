"""Host-side cost of queueing one step (launch overhead): time the Python loop while the device queue is empty enough not to
back-pressure, then the synchronised step time.  If enqueue time approaches step time the run is launch-bound."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg, diff = bench.build("unidisc-1.4b-l1280", dev, 0.1)
diff.backbone.compact_head = os.environ.get("COMPACT", "1") == "1"
batch = {k: v.to(dev) for k, v in bench.synthetic_batch("unidisc-1.4b-l1280", 8, 42).items()}
def step(i):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i)
    out.loss.backward()
for i in range(2): step(i)
torch.cuda.synchronize()
res = []
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(2 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res.append((round((t1 - t0) * 1e3, 1), round((t2 - t0) * 1e3, 1)))
print("compact", diff.backbone.compact_head, "(enqueue_ms, total_ms) per step:", res)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6): step(10 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("compact", diff.backbone.compact_head, "6 steps back to back: enqueue_ms/step", round((t1 - t0) / 6 * 1e3, 1), "total_ms/step", round((t2 - t0) / 6 * 1e3, 1))
