"""Which torch ops (fills, copies, elementwise glue) the 1.4 B training step issues besides the C-ABI kernels, by call site.  Diagnostic tool."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from torch.utils._python_dispatch import TorchDispatchMode

spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda")
torch.manual_seed(42)
cfg, diff = bench.build("unidisc-1.4b-l1280", dev, 0.1)
batch = {k: v.to(dev) for k, v in bench.synthetic_batch("unidisc-1.4b-l1280", 8, 42).items()}
for i in range(2):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i); out.loss.backward()
cnt = collections.Counter()
byt = collections.Counter()


class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        r = func(*args, **(kwargs or {}))
        name = str(func)
        if "empty" in name or "view" in name or "as_strided" in name or "detach" in name or "slice" in name or "_local_scalar" in name:
            return r
        st = [f"{f.filename.split('/')[-1]}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "unidisc_amd" in f.filename]
        n = r.numel() * r.element_size() if isinstance(r, torch.Tensor) else 0
        key = (name, st[-1] if st else "?")
        cnt[key] += 1
        byt[key] += n
        return r


with M():
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, 3); out.loss.backward()
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -byt[kv[0]])[:40]:
    print(f"{v:4d} x {k[0]:44s} {k[1]:24s} {byt[k] / 1e6:10.2f} MB")
print("total ops", sum(cnt.values()))
print("---- by count")
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{v:4d} x {k[0]:44s} {k[1]:24s} {byt[k] / 1e6:10.2f} MB")
