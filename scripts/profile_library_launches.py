"""Which call sites launch library kernels (fills, copies, sorts, elementwise glue) inside one 1.4 B training step: torch.profiler with Python stacks, CUDA
kernels that are not this repository's grouped by the innermost unidisc_amd / bench frame.  Diagnostic tool (RESULTS.md round 4)."""
import collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
WL = os.environ.get("WORKLOAD", "unidisc-1.4b-l1280")
cfg, diff = bench.build(WL, dev, 0.1)
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(WL, bench.WORKLOADS[WL]["batch"], 42).items()}
def step(i):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i); out.loss.backward(); return out
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
by = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        continue
    kern = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
    if not kern:
        continue
    names = [k.name for k in kern]
    if all("anonymous namespace" in n and "at::" not in n for n in names):
        continue   # our own kernels
    site = "?"
    for fr in (ev.stack or []):
        if "unidisc_amd" in fr or "bench.py" in fr:
            site = fr.split("/")[-1][:70]
            break
    key = (ev.name[:40], names[0][:60], site)
    by[key] += 1
    dur[key] += sum(k.duration for k in kern)
for k, c in sorted(by.items(), key=lambda kv: -dur[kv[0]])[:40]:
    print(f"{c:4d} x {dur[k]:8.1f} us  {k[0]:40s} {k[1]:60s} {k[2]}")
print("total library-launch time per step (us):", sum(dur.values()), " launches:", sum(by.values()))
