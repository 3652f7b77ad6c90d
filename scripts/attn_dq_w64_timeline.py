"""Cycle timeline of the one-wave-per-SIMD dQ kernel (block 0, four waves): s_memtime stamps per 64-key tile."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K, _lib

B, H, L, D = 8, 16, int(os.environ.get("L", 1280)), 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(4))
K.set_attention_w64(1)
o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D)
for _ in range(3): K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D)
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.udm_debug_set(b"attention_w64_timeline", ctypes.c_int64(buf.data_ptr()))
K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D)
torch.cuda.synchronize()
lib.udm_debug_set(b"attention_w64_timeline", ctypes.c_int64(0))
t = buf.cpu().reshape(8, 64)
nkv = L // 64
for w in range(2):
    s = [int(x) - int(t[w, 0]) for x in t[w]]
    n = min(nkv, 30)
    print(json.dumps(dict(wave=w, landed=s[1], tile_start=[s[2 + 2 * i] for i in range(n)], even_half=[s[3 + 2 * i] - s[2 + 2 * i] for i in range(n)],
                          tile_cycles=[s[4 + 2 * i] - s[2 + 2 * i] for i in range(n - 1)], loop_end=s[62], end=s[63])))
