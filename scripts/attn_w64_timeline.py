"""Cycle timeline of the one-wave-per-SIMD attention forward (blocks 0 and 300, four waves): s_memtime stamps -> per-section cycle counts."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K, _lib

B, H, L, D = 8, 16, int(os.environ.get("L", 1280)), 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(3))
for _ in range(3): K.attention_fwd_generic(q, k, v, B, L, H, D)
buf = torch.zeros(2 * 4 * 64, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.udm_debug_set(b"attention_w64_timeline", ctypes.c_int64(buf.data_ptr()))
K.attention_fwd_generic(q, k, v, B, L, H, D)
torch.cuda.synchronize()
lib.udm_debug_set(b"attention_w64_timeline", ctypes.c_int64(0))
t = buf.cpu().reshape(2, 4, 64)
nkv = (L + 63) // 64
for blk in range(2):
    for w in range(4):
        s = t[blk, w]
        t0 = int(s[0])
        tags = {"q_loaded": 1, "kv_landed": 2, "prologue_done": 3, "loop_end": 60, "epi_barrier": 61, "epi_lds": 62, "end": 63}
        rec = {k_: int(s[i]) - t0 for k_, i in tags.items()}
        tiles = [int(s[4 + 2 * i]) - t0 for i in range(min(nkv, 28))]
        mids = [int(s[5 + 2 * i]) - t0 for i in range(min(nkv, 28))]
        rec["tile_starts"] = tiles
        rec["phaseA_cycles"] = [m - a for a, m in zip(tiles, mids)]
        rec["tile_cycles"] = [b - a for a, b in zip(tiles[:-1], tiles[1:])]
        if os.environ.get("BRIEF"):
            tc = sorted(rec["tile_cycles"]); pa = sorted(rec["phaseA_cycles"])
            print(json.dumps(dict(abl=os.environ.get("UDM_ATTN_W64_ABL", "0"), block=blk, wave=w, first_tile=tiles[0], tile_med=tc[len(tc) // 2], tile_max=tc[-1], phaseA_med=pa[len(pa) // 2],
                                  last_tile_start=tiles[-1], loop_end=rec["loop_end"], end=rec["end"])))
        else:
            print(json.dumps(dict(block=blk, wave=w, **rec)))
