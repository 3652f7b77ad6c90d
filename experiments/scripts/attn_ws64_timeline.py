"""Cycle timeline of the wave-specialised attention forward (block 0, waves 0..7): s_memtime stamps per tile and role."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K, _lib

B, H, L, D = 8, 16, int(os.environ.get("L", 1280)), 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(3))
exp = _lib.load_experiments()   # the wave-specialised kernel lives in the experiments library (make -C unidisc_amd/csrc exp)
d = H * D
o = torch.empty((B * L, d), dtype=torch.bfloat16, device="cuda")
lse = torch.empty((B, H, L), dtype=torch.float32, device="cuda")
vp, i64 = (lambda t: ctypes.c_void_p(t.data_ptr())), ctypes.c_int64


def run(tl):
    rc = exp.udm_exp_attention_fwd_ws64(vp(q), vp(k), vp(v), vp(o), vp(lse), i64(B), i64(H), i64(L), i64(d), i64(d), i64(d), i64(d), ctypes.c_void_p(tl),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, exp.udm_last_error().decode()


for _ in range(3): run(0)
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
run(buf.data_ptr())
torch.cuda.synchronize()
t = buf.cpu().reshape(8, 64)
t00 = int(t[:, 0].min())
for w in (0, 1, 4, 5):
    s = [int(x) - t00 for x in t[w]]
    if w < 4:
        print(json.dumps(dict(wave=w, role="score", start=s[0], tile_start=[s[4 + 2 * i] for i in range(20)], phase_cycles=[s[5 + 2 * i] - s[4 + 2 * i] for i in range(20)])))
    else:
        print(json.dumps(dict(wave=w, role="pv", start=s[0], tile_start=[s[4 + 3 * i] for i in range(20)], dma_cycles=[s[5 + 3 * i] - s[4 + 3 * i] for i in range(20)],
                              reads_cycles=[s[6 + 3 * i] - s[5 + 3 * i] for i in range(1, 20)], mfma_cycles=[s[4 + 3 * (i + 1)] - s[6 + 3 * i] for i in range(1, 19)])))
