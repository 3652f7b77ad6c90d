"""Cycle timeline of the persistent 64-queries-per-wave attention forward (library built with `make -C unidisc_amd/csrc UDM_FWD64_ABL=16`):
stamps of the LAST unit (whole or half block) every workgroup processed, indexed by workgroup (s_memtime; index map in csrc/asmgen/attn_fwd64.py::stamp call sites)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

B, H, L, D = 8, 16, 1280, 128
d, M = H * D, B * L
g = torch.Generator(device="cuda").manual_seed(0)
qkr = torch.randn(M, 2 * d, device="cuda", generator=g)
qkr[:, :d] *= K.attention_q_scale(D)
qkr = qkr.to(torch.bfloat16)
qkv = torch.randn(M, 3 * d, device="cuda", generator=g).to(torch.bfloat16)
nblk = B * H * (L // 256)
grid = min(nblk, torch.cuda.get_device_properties(0).multi_processor_count // 8 * 8)
tl = torch.zeros(grid, 4, 64, dtype=torch.int32, device="cuda")
for _ in range(3):
    K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True)
torch.cuda.synchronize()
K.debug_set("attention_fwd64_timeline", tl.data_ptr())
K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True)
torch.cuda.synchronize()
K.debug_set("attention_fwd64_timeline", 0)
t = tl.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
names = {0: "entry", 1: "entry:issued", 2: "entry:drained", 3: "blk:start", 4: "blk:waited", 5: "blk:K0frags", 6: "blk:ready", 40: "epi:start", 41: "epi:end", 42: "done"}
for tag in range(8):
    names[8 + 3 * tag], names[9 + 3 * tag], names[10 + 3 * tag] = f"t{tag}:top", f"t{tag}:A", f"t{tag}:B"
for last_bid in (0, 5, grid // 2 + 3, grid - 1):   # workgroup ids
    for wave in (0, 3):
        row = t[last_bid, wave]
        if row[0] == 0:
            continue
        order = sorted((int(row[i]), names.get(i, str(i))) for i in range(62) if row[i])
        t0 = order[0][0]
        print(f"workgroup {last_bid} wave {wave}: total {order[-1][0] - t0} cycles")
        prev = t0
        line = []
        for c, n in order:
            line.append(f"{n}+{c - prev}")
            prev = c
        print("   " + " ".join(line))
