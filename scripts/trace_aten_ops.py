"""Every aten operator that runs on the GPU inside one 1.4 B training step, by Python call site (TorchDispatchMode on the forward thread and, patched in, on the
autograd thread's engine backward).  Diagnostic tool: what is left of library launches in the timed region."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
WL = os.environ.get("WORKLOAD", "unidisc-1.4b-l1280")
dev = torch.device("cuda", 0)
cfg, diff = bench.build(WL, dev, 0.1)
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(WL, bench.WORKLOADS[WL]["batch"], 42).items()}
counts = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-3]):
        if ("unidisc_amd" in fr.filename or "bench.py" in fr.filename) and "trace_aten" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "other"


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        flat = [a for a in list(args) + [out] if isinstance(a, torch.Tensor)]
        if isinstance(out, (tuple, list)):
            flat += [a for a in out if isinstance(a, torch.Tensor)]
        name = str(func).replace("aten.", "")
        skip = ("view", "reshape", "expand", "slice", "select", "as_strided", "detach", "alias", "unsqueeze", "squeeze", "t.default", "transpose", "permute", "empty", "_unsafe_view",
                "is_pinned", "record_stream", "split", "unbind", "item", "_local_scalar")
        if any(t.is_cuda for t in flat) and not any(name.startswith(s) for s in skip):
            n = max((t.numel() for t in flat if t.is_cuda), default=0)
            counts[(name, site(), n)] += 1
        return out


def step(i, mode=None):
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, i)
    out.loss.backward()
    return out


for i in range(2):
    step(i)
torch.cuda.synchronize()
bb = type(diff.backbone)
orig_bwd = bb._engine_backward


def traced_bwd(self, *a, **k):
    with Mode():
        return orig_bwd(self, *a, **k)


bb._engine_backward = traced_bwd
with Mode():
    step(2)
torch.cuda.synchronize()
tot = 0
for (name, s, n), c in sorted(counts.items(), key=lambda x: (x[0][1], x[0][0])):
    print(f"{c:4d}  {name:34s} numel<={n:<10d} {s}")
    tot += c
print("total GPU aten ops in the step:", tot)
