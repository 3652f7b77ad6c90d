"""GPU time of FusedAdamW.step() alone at 1.4 B (events around the call after a real backward), split by kernel family, and the bytes it moves."""
import json, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unidisc_amd import FusedAdamW, _lib
dev = torch.device("cuda", 0)
cfg, diff = bench.build("unidisc-1.4b-l1280", dev, 0.1)
bb = diff.backbone
batch = {k: v.to(dev) for k, v in bench.synthetic_batch("unidisc-1.4b-l1280", 8, 42).items()}
opt = FusedAdamW(bb, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0)
times = []
per = collections.Counter(); cnt = collections.Counter()
orig = _lib.call
def timed(name, *a):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(name, *a); e.record()
    recs.append((name, s, e))
for it in range(5):
    out = diff.training_step(batch, it); out.loss.backward()
    torch.cuda.synchronize()
    recs = []
    if it == 4: _lib.call = timed
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); opt.step(); e.record(); torch.cuda.synchronize()
    _lib.call = orig
    times.append(s.elapsed_time(e))
    opt.zero_grad(set_to_none=True)
for name, s, e in recs:
    per[name] += s.elapsed_time(e); cnt[name] += 1
n = sum(p.numel() for p in bb.parameters())
print(json.dumps(dict(step_ms=[round(t, 2) for t in times], params=n, per_entry_ms={k: round(v, 2) for k, v in per.items()}, launches=dict(cnt),
                      ideal_ms_at_5p5TBs=round(n * 31 / 5.5e12 * 1e3, 2))))
