"""Run the generated persistent dQ program on the CPU emulator for ONE workgroup (all the blocks it walks) and compare with a float64 attention backward: dQ, and the
planes delta | -lse | -delta it leaves behind for the dK / dV pass."""
import numpy as np
import isa
import attn_dq64 as g
from emu_dkv64 import bf16_bits, bits_f32, magic, walk


def run(B=1, H=8, L=512, grid=8, wg_id=0, mode="late", seed=0, prog=None):
    rng = np.random.default_rng(seed)
    d = g.D
    M = B * L
    c = np.float32(1.4426950408889634 / np.sqrt(d))
    qk_f = rng.standard_normal((M, 2 * H * d)) * 1.2
    qk_f[:, :H * d] *= c
    qk = bf16_bits(qk_f)
    qkv = bf16_bits(rng.standard_normal((M, 3 * H * d)))
    dout = bf16_bits(rng.standard_normal((M, H * d)))
    dqk = np.zeros((M, 2 * H * d), np.uint16)
    o_b = np.zeros((M, H * d), np.uint16)
    lse_a = np.zeros((B, H, L), np.float32)
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
            o16 = bf16_bits(p @ v)                                   # the forward stores O in bf16; delta is made from the stored rows
            o_b[b * L:(b + 1) * L, h * d:(h + 1) * d] = o16
            lse_a[b, h] = lse[:, 0]
            delta = (do * bits_f32(o16).astype(np.float64)).sum(1, keepdims=True)
            ds = p * (do @ v.T - delta)
            ref[(b, h)] = (np.log(2.0) * (ds @ k), delta[:, 0], lse[:, 0])
    wg = isa.Workgroup(lds_bytes=g.LDS_TOTAL, mode=mode)
    a_qk, a_qkv, a_do, a_o, a_lse, a_dq, a_pl = (wg.add_buffer(x) for x in (qk, qkv, dout, o_b, lse_a, dqk, planes))
    nt = L // 256
    nblk = B * H * nt
    nfull, hashalf = walk(nblk, grid)
    par = np.zeros(g.PARAM_DWORDS, np.uint32)

    def put64(i, val):
        par[i], par[i + 1] = val & 0xFFFFFFFF, val >> 32
    plane_b = B * H * L * 4
    put64(g.P_K, a_qk + H * d * 2)
    put64(g.P_V, a_qkv + 2 * H * d * 2)
    par[g.P_KSTR], par[g.P_VSTR], par[g.P_L], par[g.P_NSTEPS], par[g.P_H], par[g.P_NT] = 2 * H * d * 2, 3 * H * d * 2, L, L // 32, H, nt
    par[g.P_MG_NT], par[g.P_MG_H], par[g.P_NFULL], par[g.P_HASHALF], par[g.P_GSTRIDE], par[g.P_PLANEB] = magic(nt), magic(H), nfull, hashalf, grid, plane_b
    put64(g.P_Q, a_qk)
    put64(g.P_DO, a_do)
    put64(g.P_O, a_o)
    put64(g.P_LSE, a_lse)
    par[g.P_QSTR], par[g.P_DOSTR], par[g.P_OSTR] = 2 * H * d * 2, H * d * 2, H * d * 2
    put64(g.P_DELTA, a_pl)
    put64(g.P_DQ, a_dq)
    par[g.P_DQSTR] = 2 * H * d * 2
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
    worst_q, worst_pl, nb = 0.0, 0.0, 0
    touched = np.zeros((M, H), bool)
    units = [(bid, 0, 256) for bid in range(wg_id, nfull, grid)] + ([(nfull + 8 * (wg_id >> 4) + (wg_id & 7), 128 * ((wg_id >> 3) & 1), 128)] if hashalf else [])
    for bid, r0, nr in units:
        j, x = bid >> 3, bid & 7
        bh, tile = (j // nt) * 8 + x, j % nt
        b, h = bh // H, bh % H
        lo = tile * 256 + r0
        rows = slice(b * L + lo, b * L + lo + nr)
        dq_ref, delta, lse = ref[(b, h)]
        dq = bits_f32(dqk[rows, h * d:(h + 1) * d]).astype(np.float64)
        worst_q = max(worst_q, np.linalg.norm(dq - dq_ref[lo:lo + nr]) / np.linalg.norm(dq_ref[lo:lo + nr]))
        for pl, want in ((0, delta), (1, -lse), (2, -delta)):
            worst_pl = max(worst_pl, float(np.abs(planes[pl, b, h, lo:lo + nr] - want[lo:lo + nr]).max() / max(1.0, np.abs(want).max())))
        touched[rows, h] = True
        nb += 1
    stray = 0
    for h in range(H):
        stray += int((dqk[~touched[:, h], h * d:(h + 1) * d] != 0).sum())
    stray += int((dqk[:, H * d:] != 0).sum())
    pt = np.zeros((B, H, L), bool)
    for bid, r0, nr in units:
        j, x = bid >> 3, bid & 7
        bh, tile = (j // nt) * 8 + x, j % nt
        pt[bh // H, bh % H, tile * 256 + r0:tile * 256 + r0 + nr] = True
    stray += int((planes[:, ~pt] != 0).sum())
    return dict(dq_rel=worst_q, planes_err=worst_pl, blocks=nb, steps=steps, stray_writes=stray)


if __name__ == "__main__":
    for mode in ("late", "early"):
        print(mode, run(mode=mode, B=1, H=8, L=512, grid=8, wg_id=0))
