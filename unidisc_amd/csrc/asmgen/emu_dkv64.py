"""Run the generated persistent dK / dV program on the CPU emulator for ONE workgroup (all the blocks it walks) and compare with a float64 attention backward."""
import numpy as np
import isa
import attn_dkv64 as g


def bf16_bits(x):
    return isa._bf16_round(x.astype(np.float32)).astype(np.uint16)


def bits_f32(b):
    return isa._bf16_to_f32(b.astype(np.uint32))


def magic(d):
    return ((1 << 32) // d + 1) & 0xFFFFFFFF


def walk(nblk, grid):
    """the launcher's split (attention_dkv64.hip): whole blocks below nfull, then one half block per workgroup when the remainder is exactly half a grid"""
    rem = nblk % grid
    if rem * 2 == grid and nblk - rem >= grid and grid % 16 == 0:
        return nblk - rem, 1
    return nblk, 0


def run(B=1, H=8, L=512, grid=8, wg_id=0, mode="late", seed=0, prog=None):
    rng = np.random.default_rng(seed)
    d = g.D
    M = B * L
    c = np.float32(1.4426950408889634 / np.sqrt(d))
    # engine layout: q | k in one [M, 2 H d] buffer, v at column 2 H d of [M, 3 H d]; dO [M, H d]; dK | dV written into one [M, 2 H d] buffer
    qk_f = rng.standard_normal((M, 2 * H * d)) * 1.2
    qk_f[:, :H * d] *= c
    qk = bf16_bits(qk_f)
    qkv = bf16_bits(rng.standard_normal((M, 3 * H * d)))
    dout = bf16_bits(rng.standard_normal((M, H * d)))
    dkv = np.zeros((M, 2 * H * d), np.uint16)
    # the planes the dQ pass leaves behind: delta | -lse | -delta
    planes = np.zeros((3, B, H, L), np.float32)
    ref = {}
    for b in range(B):
        for h in range(H):
            q = bits_f32(qk[b * L:(b + 1) * L, h * d:(h + 1) * d]).astype(np.float64)
            k = bits_f32(qk[b * L:(b + 1) * L, H * d + h * d:H * d + (h + 1) * d]).astype(np.float64)
            v = bits_f32(qkv[b * L:(b + 1) * L, 2 * H * d + h * d:2 * H * d + (h + 1) * d]).astype(np.float64)
            do = bits_f32(dout[b * L:(b + 1) * L, h * d:(h + 1) * d]).astype(np.float64)
            s = q @ k.T
            m = s.max(1, keepdims=True)
            lse = m + np.log2(np.exp2(s - m).sum(1, keepdims=True))
            p = np.exp2(s - lse)
            o = p @ v
            delta = (do * o).sum(1, keepdims=True)
            dp = do @ v.T
            ds = p * (dp - delta)
            planes[0, b, h], planes[1, b, h], planes[2, b, h] = delta[:, 0], -lse[:, 0], -delta[:, 0]
            ref[(b, h)] = (np.log(2.0) * (ds.T @ q), p.T @ do)
    wg = isa.Workgroup(lds_bytes=g.LDS_TOTAL, mode=mode)
    a_qk, a_qkv, a_do, a_dkv, a_pl = wg.add_buffer(qk), wg.add_buffer(qkv), wg.add_buffer(dout), wg.add_buffer(dkv), wg.add_buffer(planes)
    nt = L // 256
    nblk = B * H * nt
    nfull, hashalf = walk(nblk, grid)
    par = np.zeros(g.PARAM_DWORDS, np.uint32)

    def put64(i, val):
        par[i], par[i + 1] = val & 0xFFFFFFFF, val >> 32
    plane_b = B * H * L * 4
    put64(g.P_Q, a_qk)
    put64(g.P_DO, a_do)
    put64(g.P_NLSE, a_pl + plane_b)
    par[g.P_QSTR], par[g.P_DOSTR], par[g.P_L], par[g.P_NSTEPS], par[g.P_H], par[g.P_NT] = 2 * H * d * 2, H * d * 2, L, L // 32, H, nt
    par[g.P_MG_NT], par[g.P_MG_H], par[g.P_NFULL], par[g.P_HASHALF], par[g.P_GSTRIDE], par[g.P_PLANEB] = magic(nt), magic(H), nfull, hashalf, grid, plane_b
    put64(g.P_K, a_qk + H * d * 2)
    put64(g.P_V, a_qkv + 2 * H * d * 2)
    par[g.P_KSTR], par[g.P_VSTR] = 2 * H * d * 2, 3 * H * d * 2
    put64(g.P_DK, a_dkv)
    put64(g.P_DV, a_dkv + H * d * 2)
    par[g.P_DKSTR], par[g.P_DVSTR] = 2 * H * d * 2, 2 * H * d * 2
    par[g.P_SCALE] = np.float32(np.log(2.0)).view(np.uint32)
    a_par = wg.add_buffer(par)
    if prog is None:
        prog, _ = g.build()
    waves = []
    for wid in range(4):
        w = isa.Wave(wg, wid)
        w.s[g.s_par.idx], w.s[g.s_par.idx + 1] = np.uint32(a_par & 0xFFFFFFFF), np.uint32(a_par >> 32)
        w.s[g.s_bid.idx], w.s[g.s_lds.idx] = np.uint32(wg_id), np.uint32(0)
        w.v[g.tmp[3].idx] = np.arange(64, dtype=np.uint32) + 64 * wid
        waves.append(w)
    steps = isa.run_workgroup(prog, wg, waves, max_steps=40_000_000)
    worst_k, worst_v, nb = 0.0, 0.0, 0
    touched = np.zeros((M, H), bool)
    units = [(bid, 0, 256) for bid in range(wg_id, nfull, grid)] + ([(nfull + 8 * (wg_id >> 4) + (wg_id & 7), 128 * ((wg_id >> 3) & 1), 128)] if hashalf else [])
    for bid, r0, nr in units:
        j, x = bid >> 3, bid & 7
        bh, tile = (j // nt) * 8 + x, j % nt
        b, h = bh // H, bh % H
        lo = tile * 256 + r0
        rows = slice(b * L + lo, b * L + lo + nr)
        dk_ref, dv_ref = ref[(b, h)]
        dk = bits_f32(dkv[rows, h * d:(h + 1) * d]).astype(np.float64)
        dv = bits_f32(dkv[rows, H * d + h * d:H * d + (h + 1) * d]).astype(np.float64)
        worst_k = max(worst_k, np.linalg.norm(dk - dk_ref[lo:lo + nr]) / np.linalg.norm(dk_ref[lo:lo + nr]))
        worst_v = max(worst_v, np.linalg.norm(dv - dv_ref[lo:lo + nr]) / np.linalg.norm(dv_ref[lo:lo + nr]))
        touched[rows, h] = True
        nb += 1
    stray = 0
    for h in range(H):
        stray += int((dkv[~touched[:, h], h * d:(h + 1) * d] != 0).sum()) + int((dkv[~touched[:, h], H * d + h * d:H * d + (h + 1) * d] != 0).sum())
    return dict(dk_rel=worst_k, dv_rel=worst_v, blocks=nb, steps=steps, stray_writes=stray)


if __name__ == "__main__":
    import sys
    kw = dict(B=1, H=8, L=512, grid=8, wg_id=0)
    for mode in ("late", "early"):
        print(mode, run(mode=mode, **kw))
