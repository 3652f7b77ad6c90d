"""Which Python call sites launch the small fill / copy kernels of one training step (1.4 B workload)?"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
cfg, diff = bench.build("unidisc-1.4b-l1280", dev, 0.1)
batch = {k: v.to(dev) for k, v in bench.synthetic_batch("unidisc-1.4b-l1280", 8, 42).items()}
def step(i):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i); out.loss.backward(); return out
for i in range(2): step(i)
torch.cuda.synchronize()
counts = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "unidisc_amd" in fr.filename or "bench.py" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "other"
def wrap(obj, name, tag):
    orig = getattr(obj, name)
    def f(*a, **k):
        t = a[0] if a and isinstance(a[0], torch.Tensor) else None
        dt = str(k.get("dtype", t.dtype if t is not None else "")).replace("torch.", "")
        counts[(tag, site(), dt)] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for n in ("zeros", "zeros_like", "full", "ones", "ones_like", "where", "cat", "argsort"): wrap(torch, n, n)
for n in ("zero_", "fill_", "copy_", "contiguous", "clone", "to", "index_select", "index_copy_", "sum", "masked_fill_"): wrap(torch.Tensor, n, "T." + n)
step(2)
torch.cuda.synchronize()
for (tag, s, dt), c in sorted(counts.items(), key=lambda x: -x[1])[:45]:
    print(f"{c:5d}  {tag:14s} {dt:10s} {s}")
