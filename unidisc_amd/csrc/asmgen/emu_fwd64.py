"""Run the generated persistent forward on the CPU emulator for ONE workgroup (all the blocks it walks) and compare with a numpy attention."""
import numpy as np
import isa
import attn_fwd64 as g


def bf16_bits(x):
    return isa._bf16_round(x.astype(np.float32)).astype(np.uint16)


def bits_f32(b):
    return isa._bf16_to_f32(b.astype(np.uint32))


def magic(d):
    return (1 << 32) // d + 1


def walk(nblk, grid):
    """the launcher's split (attention_fwd64.hip): whole blocks below nfull, then one half block per workgroup when the remainder is exactly half a grid"""
    rem = nblk % grid
    if rem * 2 == grid and nblk - rem >= grid and grid % 16 == 0:
        return nblk - rem, 1
    return nblk, 0


def run(B=1, H=8, L=512, grid=8, wg_id=0, mode="late", seed=0, spike=False, prog=None, spike_at=None):
    rng = np.random.default_rng(seed)
    d = g.D
    M = B * L
    # engine layout: q | k in one [M, 2 H d] buffer, v at column 2 H d of [M, 3 H d]
    c = np.float32(1.4426950408889634 / np.sqrt(d))
    qk_f = rng.standard_normal((M, 2 * H * d)) * 1.5
    qk_f[:, :H * d] *= c                                   # the engine stores q * log2(e) / sqrt(D)
    qk = bf16_bits(qk_f)
    qkv = bf16_bits(rng.standard_normal((M, 3 * H * d)))
    if spike:
        # a key far above the rest in a LATE tile for query 70 (the mid-block rescale) and one in the first tile for query 300; orthogonal sign patterns, so that
        # each query only sees its own spike
        alt = np.where(np.arange(d) % 2 == 0, 1.0, -1.0)
        sh, q_late, q_first = spike_at or (wg_id % H, 70, 300)
        for (b, h, key, qrow, pat) in [(0, sh, L - 100, q_late, np.ones(d)), (0, sh, 3, q_first, alt)]:
            qk[b * L + key, H * d + h * d: H * d + (h + 1) * d] = bf16_bits(6.0 * pat)
            qk[b * L + qrow, h * d:(h + 1) * d] = bf16_bits(6.0 * c * pat)
    out = np.zeros((M, H * d), np.uint16)
    lse = np.zeros((B, H, L), np.float32)
    wg = isa.Workgroup(lds_bytes=g.LDS_TOTAL, mode=mode)
    a_qk, a_qkv, a_out, a_lse = wg.add_buffer(qk), wg.add_buffer(qkv), wg.add_buffer(out), wg.add_buffer(lse)
    nt = L // 256
    nblk = B * H * nt
    nfull, hashalf = walk(nblk, grid)
    tl = np.zeros((grid, 4, 64), np.uint32)
    a_tl = wg.add_buffer(tl)
    if prog is None:
        prog, _ = g.build()
    vals = dict(qb=a_qk, kb=a_qk + H * d * 2, vb=a_qkv + 2 * H * d * 2, ob=a_out, lseb=a_lse, qstr=2 * H * d * 2, kstr=2 * H * d * 2, vstr=3 * H * d * 2, ostr=H * d * 2,
                L=L, nkv=L // 64, H=H, nt=nt, mg_nt=magic(nt), mg_H=magic(H), nfull=nfull, hashalf=hashalf, lds=0, bid=wg_id, gstride=grid)
    waves = []
    for wid in range(4):
        w = isa.Wave(wg, wid)
        for name, val in vals.items():
            r = g.S_.names[name]
            w.s[r.idx] = np.uint32(val & 0xFFFFFFFF)
            if r.n == 2:
                w.s[r.idx + 1] = np.uint32(val >> 32)
        w.v[g.tmp[0].idx] = np.arange(64, dtype=np.uint32) + 64 * wid
        w.s[g.s_dec[0].idx], w.s[g.s_dec[0].idx + 1] = np.uint32(a_tl & 0xFFFFFFFF), np.uint32(a_tl >> 32)   # (the timeline build's raw operand copy)
        waves.append(w)
    steps = isa.run_workgroup(prog, wg, waves)
    # reference over the blocks this workgroup owns
    worst_o, worst_l, nb = 0.0, 0.0, 0
    touched = np.zeros((M, H), bool)
    units = [(bid, 0, 256) for bid in range(wg_id, nfull, grid)] + ([(nfull + 8 * (wg_id >> 4) + (wg_id & 7), 128 * ((wg_id >> 3) & 1), 128)] if hashalf else [])
    for bid, r0, nr in units:
        j, x = bid >> 3, bid & 7
        bh, tile = (j // nt) * 8 + x, j % nt
        b, h = bh // H, bh % H
        rows = slice(b * L + tile * 256 + r0, b * L + tile * 256 + r0 + nr)
        q = bits_f32(qk[rows, h * d:(h + 1) * d]).astype(np.float64)
        k = bits_f32(qk[b * L:(b + 1) * L, H * d + h * d:H * d + (h + 1) * d]).astype(np.float64)
        v = bits_f32(qkv[b * L:(b + 1) * L, 2 * H * d + h * d:2 * H * d + (h + 1) * d]).astype(np.float64)
        s = q @ k.T                  # base-2 exponents: q is pre-scaled
        mrow = s.max(1, keepdims=True)
        p = np.exp2(s - mrow)
        l = p.sum(1, keepdims=True)
        o_ref = (p @ v) / l
        o = bits_f32(out[rows, h * d:(h + 1) * d]).astype(np.float64)
        worst_o = max(worst_o, np.abs(o - o_ref).max() / np.abs(o_ref).max())
        worst_l = max(worst_l, np.abs(lse[b, h, tile * 256 + r0:tile * 256 + r0 + nr] - (mrow + np.log2(l))[:, 0]).max())
        touched[rows, h] = True
        nb += 1
    stray = 0
    for h in range(H):
        stray += int((out[~touched[:, h], h * d:(h + 1) * d] != 0).sum())
    run.timeline = tl
    return dict(o_rel=worst_o, lse_err=worst_l, blocks=nb, steps=steps, stray_writes=stray)


if __name__ == "__main__":
    for mode in ("late", "early"):
        print(mode, run(mode=mode))
