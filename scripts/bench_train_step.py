"""Full optimisation step (fwd + bwd + clip + AdamW) of the 1.4 B workload: torch.optim.AdamW(fused=True) + clip_grad_norm_ with the
per-forward weight re-cast (what the reference's loop does around the backbone) vs unidisc_amd.FusedAdamW (HIP, shadows maintained)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from unidisc_amd import FusedAdamW

dev = torch.device("cuda", 0)
res = {}
for mode in ("torch_fused_adamw", "udm_fused_adamw"):
    torch.manual_seed(42)
    cfg, diff = bench.build("unidisc-1.4b-l1280", dev, 0.1)
    bb = diff.backbone
    batch = {k: v.to(dev) for k, v in bench.synthetic_batch("unidisc-1.4b-l1280", 8, 42).items()}
    if mode == "torch_fused_adamw":
        opt = torch.optim.AdamW(bb.parameters(), lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, fused=True)
    else:
        opt = FusedAdamW(bb, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0)
    def step(i):
        out = diff.training_step(batch, i)
        out.loss.backward()
        if mode == "torch_fused_adamw":
            torch.nn.utils.clip_grad_norm_(bb.parameters(), 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
        return out
    for i in range(3): out = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for i in range(n): out = step(3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    res[mode] = dict(ms_per_step=round(dt * 1e3, 2), tokens_per_s=round(8 * 1280 / dt), loss=round(float(out.loss), 4))
    del opt, diff, bb, cfg, batch
    torch.cuda.empty_cache()
print(json.dumps(res))
