"""Cycle timeline of the persistent 64-keys-per-wave dK / dV pass (library built with `make -C unidisc_amd/csrc regen all UDM_DKV64_ABL=16`): stamps of the LAST block
every workgroup processed (s_memtime; index map: csrc/asmgen/attn_dkv64.py::stamp call sites) - the last trip's four steps, the tail steps, the epilogue."""
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
do = torch.randn(M, d, device="cuda", generator=g).to(torch.bfloat16)
o, lse = K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True)
dqkr, dqkv = torch.empty_like(qkr), torch.empty_like(qkv)
nblk = B * H * (L // 256)
grid = min(nblk, torch.cuda.get_device_properties(0).multi_processor_count // 8 * 8)
tl = torch.zeros(grid, 4, 64, dtype=torch.int32, device="cuda")
for _ in range(3):
    K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
torch.cuda.synchronize()
WHICH = os.environ.get("UDM_TL", "dkv64")      # dkv64 | dq64 (library built with UDM_DKV64_ABL=16 / UDM_DQ64_ABL=16)
K.debug_set(f"attention_{WHICH}_timeline", tl.data_ptr())
K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
torch.cuda.synchronize()
K.debug_set(f"attention_{WHICH}_timeline", 0)
t = tl.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
names = {0: "entry", 1: "entry:issued", 3: "blk:start", 4: "blk:waited", 40: "epi:start", 41: "epi:end", 42: "done"}
for j in range(4):
    names[8 + j], names[12 + j] = f"main{j}", f"tail{j}"
for k in list(names):
    if 3 <= k < 42:
        names[k + 16] = "H:" + names[k]        # the half-block program's stamps
out = {}
for wgid in (0, 5, grid // 2 + 3, grid - 1):
    for wave in (0, 3):
        row = t[wgid, wave]
        if row[0] == 0:
            continue
        order = sorted((int(row[i]), names.get(i, str(i))) for i in range(62) if row[i])
        t0 = order[0][0]
        print(f"workgroup {wgid} wave {wave}: total {order[-1][0] - t0} cycles")
        prev, line = t0, []
        for c, n in order:
            line.append(f"{n}+{c - prev}")
            prev = c
        print("   " + " ".join(line))
        out[f"wg{wgid}_w{wave}"] = {n: c - t0 for c, n in order}
print(json.dumps(out))
