"""Where does the host wait for the device inside one training step?  torch.profiler, CPU side: every `aten::item` / `aten::_local_scalar_dense` / `aten::nonzero` /
`aten::is_nonzero` and every hipStreamSynchronize / hipEventSynchronize / hipMemcpy (sync) with the innermost unidisc_amd / bench frame.  WORKLOAD=... (default: the
packed 4608-token workload).  Diagnostic tool (RESULTS.md round 4)."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = os.environ.get("WORKLOAD", "unidisc-1.4b-interleaved-l4608")
dev = torch.device("cuda", 0)
cfg, diff = bench.build(wl, dev, 0.1)
B = bench.WORKLOADS[wl]["batch"]
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(wl, B, 42).items()}
def step(i):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i); out.loss.backward(); return out
for i in range(3): step(i)
torch.cuda.synchronize()
hits = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if ("unidisc_amd" in fr.filename or "bench.py" in fr.filename) and "find_host_syncs" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "other"
def wrap(obj, name, tag):
    orig = getattr(obj, name)
    def f(*a, **k):
        hits[(tag, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for n in ("item", "tolist", "nonzero", "__bool__", "__int__", "__float__", "cpu", "__index__"):
    wrap(torch.Tensor, n, "T." + n)
wrap(torch.cuda.Event, "synchronize", "Event.synchronize")
wrap(torch.cuda.Stream, "synchronize", "Stream.synchronize")
wrap(torch.cuda, "synchronize", "cuda.synchronize")
orig_getitem = torch.Tensor.__getitem__
def gi(self, idx):
    if isinstance(idx, torch.Tensor) and idx.dtype == torch.bool and self.is_cuda:
        hits[("bool-mask index", site())] += 1
    return orig_getitem(self, idx)
torch.Tensor.__getitem__ = gi
step(3)
torch.cuda.synchronize()
for (tag, s), c in sorted(hits.items(), key=lambda x: -x[1]):
    print(f"{c:4d}  {tag:22s} {s}")
