"""Cycle timeline of the wave-specialised attention forward (block 0, waves 0..7): s_memtime stamps per tile and role."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K, _lib

B, H, L, D = 8, 16, int(os.environ.get("L", 1280)), 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(3))
K.set_attention_w64(2)
for _ in range(3): K.attention_fwd_generic(q, k, v, B, L, H, D)
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.udm_attention_w64_timeline(ctypes.c_void_p(buf.data_ptr()))
K.attention_fwd_generic(q, k, v, B, L, H, D)
torch.cuda.synchronize()
lib.udm_attention_w64_timeline(ctypes.c_void_p(0))
t = buf.cpu().reshape(8, 64)
t00 = int(t[:, 0].min())
for w in (0, 1, 4, 5):
    s = [int(x) - t00 for x in t[w]]
    if w < 4:
        print(json.dumps(dict(wave=w, role="score", start=s[0], tile_start=[s[4 + 2 * i] for i in range(20)], phase_cycles=[s[5 + 2 * i] - s[4 + 2 * i] for i in range(20)])))
    else:
        print(json.dumps(dict(wave=w, role="pv", start=s[0], tile_start=[s[4 + 3 * i] for i in range(20)], dma_cycles=[s[5 + 3 * i] - s[4 + 3 * i] for i in range(20)],
                              reads_cycles=[s[6 + 3 * i] - s[5 + 3 * i] for i in range(1, 20)], mfma_cycles=[s[4 + 3 * (i + 1)] - s[6 + 3 * i] for i in range(1, 19)])))
