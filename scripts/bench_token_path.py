"""Token data path (N4) measurement: the assembly kernel against its HBM roofline, and batches/s of TokenBatcher (resident shard vs host-staged rows)
against what the 1.4 B training step consumes (8 x 1280 tokens per ~100 ms)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K, token_data as TD

Vt, Lt, Li = 32001, 128, 1152
res = {}
g = torch.Generator().manual_seed(0)
for B in (8, 4096):
    n = max(B, 65536)
    txt = torch.randint(0, Vt - 1, (n, Lt), generator=g, dtype=torch.int32).cuda()
    img = torch.randint(0, 16384, (n, Li), generator=g, dtype=torch.int32).to(torch.int16).cuda()
    msk = (torch.rand(n, Lt, generator=g) < 0.9).cuda()
    idx = torch.randint(0, n, (B,), generator=g).cuda()
    for _ in range(5): K.assemble_joint_tokens(txt, msk, img, Vt, idx=idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 200 if B == 8 else 50
    e0.record()
    for _ in range(it): K.assemble_joint_tokens(txt, msk, img, Vt, idx=idx)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    bytes_ = B * (Lt * 4 + Lt + Li * 2 + (Lt + Li) * (8 + 1 + 8))   # algorithmic: fields read once, three outputs written once
    res[f"assemble_B{B}"] = dict(us=round(us, 2), algorithmic_bytes=bytes_, GBps=round(bytes_ / us / 1e3, 1), frac_of_8TBps=round(bytes_ / us / 1e3 / 8000, 4))

n = 200000
f = dict(txt_input_ids=np.random.randint(0, Vt - 1, (n, Lt), dtype=np.int32), txt_attention_mask=np.random.rand(n, Lt) < 0.9,
         img_input_ids=np.random.randint(0, 16384, (n, Li)).astype(np.int16))
for resident in (True, False):
    tb = TD.TokenBatcher([TD.TokenShard(f, "s0")], [1.0], 8, Vt, "cuda", seed=1, resident=resident)
    for _ in range(5): tb.next()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 300
    for _ in range(N): b = tb.next()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res["batcher_resident" if resident else "batcher_host_staged"] = dict(batches_per_s=round(N / dt, 1), tokens_per_s=round(N * 8 * (Lt + Li) / dt), ms_per_batch=round(dt / N * 1e3, 3))
res["training_consumes_tokens_per_s"] = 104000
print(json.dumps(res))
