"""Generator of the one-wave-per-SIMD attention backward dQ pass for gfx950 (head dim 128, no mask, q pre-scaled, L % 256 == 0).

Replaces the dQ half of the backward of `flash_attn_qkvpacked_func` (reference models/dit.py:843) on the headline path; `attn_bwd_dq_kernel` of attention.hip keeps every
other shape.  One workgroup = 4 waves = one wave per SIMD = 256 queries of one (batch, head); a wave owns TWO 32-query blocks (q = 0, 1) and the whole 512-register
file, so that every K / V fragment read from LDS feeds two MFMAs.  It also computes delta = rowsum(dO * O) and leaves the planes delta | -lse | -delta behind for the
dK / dV pass (attn_dkv64.py), like the kernel it replaces.

    S^T  = K Q^T   (C operand of the chain's first MFMA = -lse of the lane's query: the accumulator IS s - lse)    A = K row fragment (LDS), B = Q fragment (resident)
    dP^T = V dO^T  (C operand = -delta)                                                                            A = V row fragment,      B = dO fragment (resident)
    P = exp2(S), dS = P (dP - delta)    VALU; dS packed to bf16 in place over dP
    dQ^T += K^T dS^T                                                                                               A = transposing reads of the K tile, B = dS from registers

Registers of a wave:
    a[0:127]    dQ^T accumulators [q][32-column group i][16];  a[128:191] Q fragments, a[192:255] dO fragments (MFMA B operands) [q][k-step][4], loaded once per block
    v[0:63]     S^T of TWO 32-key steps [buffer][q][16] (the muls of step t read buffer t & 1 while the MFMAs of S(t+1) write the other)
    v[64:95]    dP^T of one step [q][16];  v[96:159] -lse / -delta of the lane's queries in all 16 registers of a block (the C operands)
    v[160:191]  a ring of eight 4-register fragment slots;  v[192:..] addresses, LDS-DMA offsets, scratch

Step t (32 keys, 48 MFMAs, ONE barrier):
    G0  16 MFMAs dP(t)       under them: exp2 of S(t)
    G1  16 MFMAs S(t+1)      under them: dS(t) = P dP, packed in place
    G2  16 MFMAs dQ^T += ..  under them: the LDS-DMA of step t + 3
The first 16-key chunk of S(t+1) is exponentiated under the dQ group of step t, the second under the dP group of step t+1.
LDS ring, persistence, balanced walk (128-query half blocks, NQ = 1), the per-wave staging area for the next block's Q / dO / O rows, the row-store epilogue, counted waits
(`auto_waits`), lint and emulation: as in attn_dkv64.py."""
import sys
from isa import *   # noqa: F401,F403
from attn_dkv64 import Gaps, auto_waits, epi_addresses as _epi_addresses, epi_block

D, KS = 128, 8
NST, PD = 4, 3
PIECE = 1040
TILE = 8 * PIECE              # 32 rows
STG = 2 * TILE                # K tile | V tile
LDS_RING = NST * STG
# Behind the ring: ONE area of two tiles per wave.  The next block's Q rows, then its dO rows, then its O rows pass through it as LDS-DMA pieces (coalesced 256-byte
# rows; a direct fragment load touches 32 cache lines per instruction and stalled the issue ~200 cycles each: 50 of them cost 12 k cycles per block, first timeline)
# and are picked up as row fragments, each a step or more after its DMA went out: Q rows leave under the third-last step and are read under the second-last (behind
# the block's last S group), dO rows leave under the last step and are read at its end, O rows fly through the epilogue and are read by the next block start.
A_BASE = LDS_RING
AREA = 2 * TILE
E_BASE = A_BASE + 4 * AREA    # epilogue staging: 4 KiB per wave (attn_dkv64.epi_block), not inside the area
E_STG = 4096
LDS_TOTAL = E_BASE + 4 * E_STG

# ---- the kernel's parameter block (kernarg segment; attention_dq64.hip declares the same struct): dword offsets
P_K, P_V, P_KSTR, P_VSTR, P_L, P_NSTEPS, P_H, P_NT, P_MG_NT, P_MG_H, P_NFULL, P_HASHALF, P_GSTRIDE, P_PLANEB = 0, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
P_Q, P_DO, P_O, P_LSE, P_QSTR, P_DOSTR, P_OSTR = 16, 18, 20, 22, 24, 25, 26
P_DELTA, P_DQ, P_DQSTR, P_SCALE, P_TL = 28, 30, 32, 33, 34
PARAM_DWORDS = 36

NQ = 2        # 32-query blocks per wave of the program being generated: 2 (256-query blocks) or 1 (the 128-query half blocks that balance the walk)

V = Alloc("v", 0, 255)
A = Alloc("a", 0, 256)
S_ = Alloc("s", 36, 100)

Sb = V("S", 64, 2)
dPb = V("dP", 32, 2)
negl, negd = V("negl", 32), V("negd", 32)
Fq = V("Fq", 32)
qaddr, taddr = V("qaddr"), V("taddr")
dk, dv = V("dk", 2), V("dv", 2)
aaddr = V("aaddr")            # row-fragment base inside this wave's staging area
lr, lcs = V("lr"), V("lcs")   # LDS-DMA lane constants: row of the piece (lane >> 2) & 3, byte (4 (lane >> 4) + (lane & 3)) * 16 inside the row
da = V("da", 4)               # LDS-DMA source offsets of the operand being staged: pieces 4 g + j
loff = V("loff")              # (lane & 31) * 4: lse, the planes
tmp = V("tmp", 4)
dQT, Qf, dOf = A("dQ", 128), A("Q", 64), A("dO", 64)

s_kb, s_vb = S_("kb", 2, 4), S_("vb", 2, 2)
s_kstr, s_vstr, s_L, s_nsteps, s_H, s_nt, s_mg_nt, s_mg_H, s_nfull, s_hashalf, s_gstride, s_planeB = (S_(n) for n in (
    "kstr", "vstr", "L", "nsteps", "H", "nt", "mg_nt", "mg_H", "nfull", "hashalf", "gstride", "planeB"))
s_par = S_("par", 2, 2)
s_bid, s_lds = S_("bid"), S_("lds")
s_tm = S_("tm", 2, 2)
s_wave, s_loop = S_("wave"), S_("loop")
s_T = S_("T", 16, 4)           # transient parameters (seam: q, dO, o, lse bases + strides; block start: the planes; epilogue: dq, stride, scale)
s_kt, s_vt = S_("kt", 2, 2), S_("vt", 2, 2)
s_kstep, s_vstep = S_("kstep"), S_("vstep")
s_kdst, s_vdst = S_("kdst"), S_("vdst")
s_t = [S_("t0", 1, 2)] + [S_(f"t{i}") for i in range(1, 6)]
s_kn, s_vn = S_("kn", 2, 2), S_("vn", 2, 2)
s_nbid, s_moden, s_hidx, s_hidxn, s_wg, s_wq = (S_(n) for n in ("nbid", "moden", "hidx", "hidxn", "wg", "wq"))
assert S_.next <= 100, S_.next
assert (s_kb.idx, s_planeB.idx) == (36, 51) and s_t[4].idx % 2 == 0

ABL = 0
v_tl = R("v", 254)


def stamp(idx):
    """timeline builds: s_memtime -> lane idx of v_tl (the half-block program's stamps of the block sit 16 lanes behind the whole-block program's: both stay readable)"""
    if not (ABL & 16):
        return []
    if 3 <= idx < 42 and NQ == 1:
        idx += 16
    return [s_memtime(s_tm), s_waitcnt(lgkmcnt=0), v_writelane_b32(v_tl, s_tm[0], idx)]


def Sblk(buf, q):
    return Sb.sub((buf * 2 + q) * 16, 16)


def dPblk(q):
    return dPb.sub(16 * q, 16)


def dSfr(q, c2):
    return dPblk(q).sub(8 * c2, 4)


def slot(n):
    return Fq.sub(4 * (n % 8), 4)


def dQblk(q, i):
    return dQT.sub((q * 4 + i) * 16, 16)


def Qfr(q, ks):
    return Qf.sub((q * KS + ks) * 4, 4)


def dOfr(q, ks):
    return dOf.sub((q * KS + ks) * 4, 4)


def row_frag(n, tile, ks):
    return [ds_read_b128(slot(n), qaddr, tile + (ks >> 1) * 256 + (ks & 1) * 32)]


def tr_frag(n, tile, i, c2):
    return [ds_read_b64_tr_b16(slot(n).sub(2 * h2, 2), taddr, tile + (4 * c2 + 2 * h2) * PIECE + i * 256) for h2 in range(2)]


def dma_step(stage):
    u = []
    for j in range(2):
        u.append([s_add_u32(M0, s_kdst, stage * STG + j * PIECE), s_nop(0), global_load_lds_dwordx4(dk[j], s_kt)])
    for j in range(2):
        u.append([s_add_u32(M0, s_vdst, stage * STG + j * PIECE), s_nop(0), global_load_lds_dwordx4(dv[j], s_vt)])
    u.append([s_add_u32(s_kt[0], s_kt[0], s_kstep), s_addc_u32(s_kt[1], s_kt[1], 0)])
    u.append([s_add_u32(s_vt[0], s_vt[0], s_vstep), s_addc_u32(s_vt[1], s_vt[1], 0)])
    return u


def _abl(prog):
    out = []
    for ins in prog:
        k = ins.kind
        if (ABL & 1) and k in ("valu", "trans", "dot") and not ins.meta.get("keep"):
            continue
        if (ABL & 2) and k == "lds_rd":
            continue
        if (ABL & 4) and (k in ("dma", "barrier") or (k == "wait" and ins.meta.get("vmcnt") is not None) or
                          (k == "salu" and any(w in ins.writes for w in [("m0", 0)] + s_kt.regs() + s_vt.regs()))):
            continue
        if (ABL & 4) and k == "nop":
            continue
        if (ABL & 8) and k == "mfma":
            continue
        out.append(ins)
    return out


def LA():
    return 7 if NQ == 2 else 4


def n_pref():
    return (LA() + NQ - 1) // NQ


def exp_units(buf, c2):
    """exp2 of the 16-key chunk c2 of S in `buf` (in place).  Chunk 0 of S(t+1) runs under the dQ group of step t - right behind the S(t+1) MFMAs - chunk 1 under the dP
    group of step t+1: 32 quarter-rate instructions under ONE 16-MFMA group saturate the VALU (first timeline: 40.6 cycles per MFMA against the dK / dV program's 36.5)"""
    u = []
    for q in range(NQ):
        s = Sblk(buf, q)
        u += [[v_exp_f32(s[8 * c2 + e], s[8 * c2 + e])] for e in range(8)]
    return u


def ds_units(buf):
    halves = []
    for c2 in range(2):
        u = []
        for q in range(NQ):
            s, d = Sblk(buf, q), dPblk(q)
            u += [[v_pk_mul_f32(d.sub(8 * c2 + 2 * j, 2), s.sub(8 * c2 + 2 * j, 2), d.sub(8 * c2 + 2 * j, 2))] for j in range(4)]
        for q in range(NQ):
            d = dPblk(q)
            u += [[v_cvt_pk_bf16_f32(d[8 * c2 + j], d[8 * c2 + 2 * j], d[8 * c2 + 2 * j + 1])] for j in range(4)]
        halves.append(u)
    return halves


def s_group(G, g0, stage_next, buf_next):
    """the S(t+1) group at MFMA index g0: MFMAs and the K(t+1) row fragments that are not prefetched"""
    g = g0
    for ks in range(KS):
        for q in range(NQ):
            d = Sblk(buf_next, q)
            G.m[g] = v_mfma_f32_32x32x16_bf16(d, slot(8 + ks), Qfr(q, ks), negl.sub(16 * q, 16) if ks == 0 else d)
            g += 1
    for ks in range(KS):        # (at a block start the first of them land ahead of the group's first MFMA: gap -1)
        G.put(max(-1, g0 + ks * NQ - LA()), row_frag(8 + ks, stage_next * STG, ks))


def v_prefetch(stage):
    """the first V row fragments of the step in `stage` (read before that step's dP group starts)"""
    return [row_frag(n, stage * STG + TILE, n) for n in range(n_pref())]


def body(j, variant):
    """step t with t % 4 == j.  variant: 'main'; 'head' (the block's first step: the accumulators start from the constant 0); 'tail0' .. 'tail3' (the block's last four
    steps: tail1 .. tail3 refill the ring with the next block's steps 0 .. 2, tail3 has no S(t+1) and loads the next block's Q / dO / O)"""
    nm = 8 * NQ
    last = variant == "tail3"
    st, stn, rst = j, (j + 1) % NST, (j + PD) % NST
    KT, VT = st * STG, st * STG + TILE
    bt, bn = j & 1, (j + 1) & 1
    G = Gaps(2 * nm if last else 3 * nm)
    gq = nm if last else 2 * nm            # first MFMA of the dQ group
    g = 0
    for ks in range(KS):
        for q in range(NQ):
            d = dPblk(q)
            G.m[g] = v_mfma_f32_32x32x16_bf16(d, slot(ks), dOfr(q, ks), negd.sub(16 * q, 16) if ks == 0 else d)
            g += 1
    g = gq
    for c2 in range(2):
        for i in range(4):
            for q in range(NQ):
                acc = dQblk(q, i)
                G.m[g] = v_mfma_f32_32x32x16_bf16(acc, slot(16 + c2 * 4 + i), dSfr(q, c2), 0 if (variant == "head" and c2 == 0) else acc)
                g += 1
    G.pre += stamp((8 if variant in ("main", "head") else 12) + j)
    G.put(0, [s_waitcnt(vmcnt=4 + 8 * NQ if variant == "tail2" else 4)])      # (tail2: the Q rows' LDS-DMA of the previous step is in the queue behind the ring's)
    G.put(1, [s_barrier()])
    for ks in range(n_pref(), KS):
        G.put(ks * NQ - LA(), row_frag(ks, VT, ks))
    G.spread(exp_units(bt, 1), 0, nm - 1)
    d0, d1 = ds_units(bt)
    if not last:
        s_group(G, nm, stn, bn)
        if NQ == 2:
            G.spread(d0 + d1, 19, 38)
        else:
            G.spread(d0, 12, 15)
            G.spread(d1, 16, 19)
        for c2 in range(2):
            for i in range(4):
                n = 16 + c2 * 4 + i
                G.put(n * NQ - LA(), tr_frag(n, KT, i, c2))
        if variant == "tail1":
            G.put(2 * nm, [s_mov_b64(s_kt, s_kn), s_mov_b64(s_vt, s_vn)])
        G.spread(dma_step(rst), 2 * nm + 1, 2 * nm + (10 if NQ == 2 else 5))
        G.spread(exp_units(bn, 0), 2 * nm + (4 if NQ == 2 else 5), 3 * nm - 1)      # (S(t+1)'s last MFMA is index 2 nm - 1: >= 16 wait states behind it)
        for n, u in enumerate(v_prefetch(stn)):
            G.put(3 * nm + n * NQ - LA(), u)
    else:
        # no S(t+1): the dS arithmetic sits between the two groups (once per block)
        G.put(nm - 1, [s_nop(15)] + [x for u in d0 + d1 for x in u] + [s_nop(1)])
        for c2 in range(2):
            for i in range(4):
                n = 16 + c2 * 4 + i
                G.put(max(0, (n - 8) * NQ - LA()), tr_frag(n, KT, i, c2))
        G.spread(dma_step(rst), nm + 2, nm + (11 if NQ == 2 else 6))      # (behind the dO rows' LDS-DMA: the step's closing wait counts these four)
    return G


def finish(prog, pending):
    prog, pend = auto_waits(prog, pending)
    return (_abl(prog) if ABL else prog), pend


def block_coords(bid, hidx):
    """-> t0 = first query row of the block inside its sequence, t1 = b L, t2 = b H + h, t3 = h * 256 bytes"""
    t0, t1, t2, t3 = s_t[0], s_t[1], s_t[2], s_t[3]
    p = [s_lshr_b32(t0, bid, 3), s_and_b32(t1, bid, 7), s_mul_hi_u32(t2, t0, s_mg_nt), s_mul_i32(t3, t2, s_nt), s_sub_u32(t0, t0, t3),
         s_lshl_b32(t2, t2, 3), s_add_u32(t2, t2, t1),
         s_mul_hi_u32(t1, t2, s_mg_H), s_mul_i32(t3, t1, s_H), s_sub_u32(t3, t2, t3), s_lshl_b32(t3, t3, 8),
         s_mul_i32(t1, t1, s_L), s_lshl_b32(t0, t0, 8)]
    if not (isinstance(hidx, int) and hidx == 0):
        p += [s_lshl_b32(s_t[5], hidx, 7), s_add_u32(t0, t0, s_t[5])]
    return p


def ptr(dst, rows, base, stride):
    return [s_mul_i32(dst[0], rows, stride), s_mul_hi_u32(dst[1], rows, stride), s_add_u32(dst[0], dst[0], s_t[3]), s_addc_u32(dst[1], dst[1], 0),
            s_add_u32(dst[0], dst[0], base[0]), s_addc_u32(dst[1], dst[1], base[1])]


def ptr_inplace(base, rows, stride):
    """base += rows * stride + h * 256 (through t4 : t5)"""
    lo, hi = s_t[4], s_t[5]
    return [s_mul_i32(lo, rows, stride), s_mul_hi_u32(hi, rows, stride), s_add_u32(lo, lo, s_t[3]), s_addc_u32(hi, hi, 0),
            s_add_u32(base[0], base[0], lo), s_addc_u32(base[1], base[1], hi)]


def stream_ptrs(bid, hidx, k, v):
    p = block_coords(bid, hidx)
    return p + ptr(k, s_t[1], s_kb, s_kstr) + ptr(v, s_t[1], s_vb, s_vstr)


def seam_ptrs(bid, hidx, is_half):
    """the operand bases of block (bid, hidx) AT THIS WAVE'S FIRST ROW into s_T: [0:1] q, [2:3] dO, [4:5] o, [6:7] lse, [8] qstr, [9] dostr, [10] ostr.
    is_half: an SGPR that is 1 when the block is a 128-query half (a wave then owns 32 queries), or a bool."""
    p = [s_load_dwords(s_T.sub(0, 8), s_par, 4 * P_Q), s_load_dwords(s_T.sub(8, 4), s_par, 4 * P_QSTR), s_waitcnt(lgkmcnt=0)]
    qb, dob, ob, lseb = (s_T.sub(2 * i, 2) for i in range(4))
    qstr, dostr, ostr = s_T[8], s_T[9], s_T[10]
    if isinstance(is_half, bool):
        p += [s_mov_b32(s_wq, 32 if is_half else 64)]
    else:
        p += [s_lshl_b32(s_wq, is_half, 5), s_sub_u32(s_wq, 64, s_wq)]
    p += block_coords(bid, hidx)
    p += [s_mul_i32(s_t[4], s_wave, s_wq), s_add_u32(s_t[0], s_t[0], s_t[4]), s_add_u32(s_t[1], s_t[1], s_t[0])]      # t0: row in the sequence, t1: global row
    p += ptr_inplace(qb, s_t[1], qstr) + ptr_inplace(dob, s_t[1], dostr) + ptr_inplace(ob, s_t[1], ostr)
    p += [s_mul_i32(s_t[2], s_t[2], s_L), s_add_u32(s_t[2], s_t[2], s_t[0]), s_lshl_b32(s_t[2], s_t[2], 2), s_add_u32(lseb[0], lseb[0], s_t[2]), s_addc_u32(lseb[1], lseb[1], 0)]
    for ins in p:
        ins.meta["keep"] = True
    return p


def stage_dma(base, stride, area_off, ntiles, rewind=False):
    """`ntiles` x 32 rows from `base` (advanced past them) into this wave's staging area at `area_off`, as units: piece 4 g + j = rows 16 g + 4 j .. + 3.
    Uses t0 (LDS destination), t1 (16 rows in bytes) until its last unit has been issued.  rewind (the program of whole blocks staging for a block that may be a
    half, s_wq = 32): the second tile re-reads the first tile's rows instead of running past the wave's 32."""
    pre = [s_lshl_b32(s_t[1], stride, 2), v_mul_lo_u32(da[0], lr, stride), v_add_u32(da[0], da[0], lcs)]
    pre += [v_add_u32(da[j], s_t[1], da[j - 1]) for j in range(1, 4)]
    pre += [s_mul_i32(s_t[0], s_wave, AREA), s_add_u32(s_t[0], s_t[0], s_lds), s_add_u32(s_t[0], s_t[0], A_BASE + area_off), s_lshl_b32(s_t[1], stride, 4)]
    for ins in pre:
        ins.meta["keep"] = True
    units = [pre]
    for g in range(2 * ntiles):
        for j in range(4):
            units.append([s_add_u32(M0, s_t[0], (4 * g + j) * PIECE), s_nop(0), global_load_lds_dwordx4(da[j], base)])
        units.append([s_add_u32(base[0], base[0], s_t[1]), s_addc_u32(base[1], base[1], 0)])
        if rewind and ntiles == 2 and g == 1:
            units.append([s_sub_u32(s_t[2], 64, s_wq), s_mul_i32(s_t[2], s_t[2], stride), s_sub_u32(base[0], base[0], s_t[2]), s_subb_u32(base[1], base[1], 0)])
    return units


def stage_reads(dst, area_off, ntiles):
    """the staged rows as row fragments: dst(q, ks) <- lane (row l & 31 of tile q, 8 columns 16 ks + 8 (l >> 5) ..)"""
    return [[ds_read_b128(dst(q, ks), aaddr, area_off + q * TILE + (ks >> 1) * 256 + (ks & 1) * 32)] for q in range(ntiles) for ks in range(KS)]


def lse_loads():
    return [[global_load_dword(tmp[q], loff, s_T.sub(6, 2), 128 * q)] for q in range(NQ)]


def next_block_choice():
    p = []
    if NQ == 1:
        p += [s_mov_b32(s_moden, 2), s_mov_b32(s_nbid, s_bid), s_mov_b32(s_hidxn, s_hidx)]
    else:
        p += [s_mov_b32(s_hidxn, 0), s_mov_b32(s_moden, 0),
              s_add_u32(s_nbid, s_bid, s_gstride), s_cmp_lt_u32(s_nbid, s_nfull), s_cbranch_scc1("L_np"),
              s_mov_b32(s_moden, 2), s_mov_b32(s_nbid, s_bid), s_cmp_eq_u32(s_hashalf, 0), s_cbranch_scc1("L_np"),
              s_mov_b32(s_moden, 1), s_lshr_b32(s_nbid, s_wg, 4), s_lshl_b32(s_nbid, s_nbid, 3), s_and_b32(s_t[0], s_wg, 7), s_add_u32(s_nbid, s_nbid, s_t[0]),
              s_add_u32(s_nbid, s_nbid, s_nfull), s_lshr_b32(s_hidxn, s_wg, 3), s_and_b32(s_hidxn, s_hidxn, 1),
              label("L_np")]
    return p + stream_ptrs(s_nbid, s_hidxn, s_kn, s_vn)


INPUTS = ["par", "bid", "lds", "tid"]


def entry():
    p = [comment("---- entry: parameters, constants of the wave, the first block's operands and first three steps")]
    raw = lambda t: Inst(t, "raw")
    p += [raw(f"s_mov_b64 {s_par}, %0"), raw(f"s_mov_b32 {s_bid}, %1"), raw(f"s_mov_b32 {s_lds}, %2")]
    tid = tmp[3]
    p += [raw(f"v_mov_b32 {tid}, %3")]
    p += [s_load_dwords(R("s", 36, 16), s_par, 0), s_waitcnt(lgkmcnt=0)]
    if ABL & 16:
        p += [s_load_dwords(s_T.sub(0, 2), s_par, 4 * P_TL), s_waitcnt(lgkmcnt=0), v_mov_b32(v_tl, 0), s_nop(1), v_writelane_b32(v_tl, s_T[0], 62), v_writelane_b32(v_tl, s_T[1], 63)] + stamp(0)
    t = [Fq[i] for i in range(8)]
    p += [s_nop(0), v_lshrrev_b32(t[1], 6, tid), s_nop(0), v_readfirstlane_b32(s_wave, t[1]), v_and_b32(t[0], 63, tid)]
    lane_v, l31, hi = t[0], t[1], t[2]
    p += [v_and_b32(l31, 31, lane_v), v_lshrrev_b32(hi, 5, lane_v)]
    p += [v_lshrrev_b32(t[3], 2, lane_v), v_and_b32(t[3], 3, t[3])]
    p += [s_lshl_b32(s_t[0], s_wave, 3), s_nop(0), v_add_u32(t[3], s_t[0], t[3])]
    p += [v_lshrrev_b32(t[4], 4, lane_v), v_lshlrev_b32(t[4], 2, t[4]), v_and_b32(t[5], 3, lane_v), v_add_u32(t[4], t[4], t[5]), v_lshlrev_b32(t[4], 4, t[4])]
    for j in range(2):
        p += [v_add_u32(t[5], 4 * j, t[3]), v_mul_lo_u32(dk[j], t[5], s_kstr), v_mul_lo_u32(dv[j], t[5], s_vstr)]
        p += [v_add_u32(dk[j], dk[j], t[4]), v_add_u32(dv[j], dv[j], t[4])]
    p += [s_lshl_b32(s_kstep, s_kstr, 5), s_lshl_b32(s_vstep, s_vstr, 5)]
    p += [s_mul_i32(s_t[0], s_wave, 2 * PIECE), s_add_u32(s_kdst, s_lds, s_t[0]), s_add_u32(s_vdst, s_kdst, TILE)]
    p += [s_mov_b32(s_t[0], PIECE), v_lshrrev_b32(t[3], 2, l31), v_mul_lo_u32(t[3], t[3], s_t[0]), v_and_b32(t[4], 3, l31), v_lshlrev_b32(t[4], 6, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_lshlrev_b32(t[4], 4, hi), v_add_u32(t[3], t[3], t[4]), v_add_u32(qaddr, s_lds, t[3])]
    p += [v_mul_lo_u32(t[3], hi, s_t[0]), v_and_b32(t[4], 15, lane_v), v_lshrrev_b32(t[4], 2, t[4]), v_lshlrev_b32(t[4], 6, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_lshrrev_b32(t[4], 4, lane_v), v_and_b32(t[4], 1, t[4]), v_lshlrev_b32(t[4], 5, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_and_b32(t[4], 3, lane_v), v_lshlrev_b32(t[4], 3, t[4]), v_add_u32(t[3], t[3], t[4]), v_add_u32(taddr, s_lds, t[3])]
    # staging: lane constants of the LDS-DMA pieces, the fragment base inside this wave's area, (lane & 31) * 4
    p += [v_lshrrev_b32(lr, 2, lane_v), v_and_b32(lr, 3, lr), v_lshrrev_b32(lcs, 4, lane_v), v_lshlrev_b32(lcs, 2, lcs), v_and_b32(t[5], 3, lane_v), v_add_u32(lcs, lcs, t[5]),
          v_lshlrev_b32(lcs, 4, lcs), s_mul_i32(s_t[0], s_wave, AREA), s_add_u32(s_t[0], s_t[0], A_BASE), s_nop(0), v_add_u32(aaddr, s_t[0], qaddr), v_lshlrev_b32(loff, 2, l31)]
    p += [s_mov_b32(s_wg, s_bid), s_mov_b32(s_hidx, 0)]
    for ins in p:
        ins.meta["keep"] = True
    p += stream_ptrs(s_bid, 0, s_kt, s_vt)
    # ---- the first block (always a whole one): Q, dO and O through the staging area one after the other, lse, the ring's first three steps
    p += seam_ptrs(s_bid, 0, False)
    for u in stage_dma(s_T.sub(0, 2), s_T[8], 0, 2):
        p += u
    for s_ in range(PD):
        for u in dma_step(s_):
            p += u
    p += [s_waitcnt(vmcnt=4 * PD)]
    for u in stage_reads(Qfr, 0, 2):
        p += u
    p += [s_waitcnt(lgkmcnt=0)]
    for u in stage_dma(s_T.sub(2, 2), s_T[9], 0, 2):
        p += u
    for u in lse_loads():
        p += u
    p += [s_waitcnt(vmcnt=0)]
    for u in stage_reads(dOfr, 0, 2):
        p += u
    p += [s_waitcnt(lgkmcnt=0)]
    for u in stage_dma(s_T.sub(4, 2), s_T[10], 0, 2):
        p += u
    p += [s_waitcnt(vmcnt=0)]       # (once per workgroup: the block start's counted wait assumes nothing but stores behind what it needs)
    p += stamp(1)
    return p


def delta_math(q):
    """dlt[q] = this lane's half of rowsum(dO * O) of its query in block q (the O rows sit in the S registers 32 q .. 32 q + 31)"""
    p = [v_mov_b32(dPb[q], 0)]
    for r in range(32):
        p += [v_accvgpr_read_b32(dPb[2 + (r & 1)], dOf[q * 32 + r]), v_dot2c_f32_bf16(dPb[q], Sb[32 * q + r], dPb[2 + (r & 1)])]
    return p


def block_start():
    """per block: what comes next; everything but the previous block's stores has landed (the ring's first steps, the O rows, lse); delta and the planes; the C operands;
    the S(0) group"""
    nm = 8 * NQ
    p = [comment("---- block start"), label("L_block")] + stamp(3)
    p += next_block_choice()
    # everything older than the previous block's stores (8 per 32-row block) has landed: the ring's next steps, the O rows, lse; a workgroup's first block finds the
    # queue drained by the entry
    p += [s_waitcnt(vmcnt=8 * NQ), s_barrier()] + stamp(4)
    p += [s_lshr_b32(s_loop, s_nsteps, 2), s_sub_u32(s_loop, s_loop, 1)]
    # ---- delta = rowsum(dO * O): a lane holds 64 of its query's 128 columns (the other half sits in lane + 32); the O rows wait in the staging area
    dlt = [dPb[0], dPb[1]]
    sw = [dPb[4], dPb[5]]
    for u in stage_reads(lambda q, ks: Sb.sub(32 * q + 4 * ks, 4), 0, NQ):
        p += u
    p += [s_waitcnt(lgkmcnt=0)]
    for q in range(NQ):
        p += delta_math(q)
    p += [s_nop(3)]          # (DOT write -> a different VALU instruction reading it: 3 wait states)
    for q in range(NQ):
        p += [v_mov_b32(sw[q], dlt[q])]
    p += [s_nop(1)] + [v_permlane32_swap_b32(dlt[q], sw[q]) for q in range(NQ)]
    p += [v_add_f32(dlt[q], dlt[q], sw[q]) for q in range(NQ)]
    # ---- the planes delta | -lse | -delta (read by the dK / dV pass), the C operands
    nl, nd = [dPb[6], dPb[7]], [dPb[8], dPb[9]]
    for q in range(NQ):
        p += [v_sub_f32(nl[q], 0, tmp[q]), v_sub_f32(nd[q], 0, dlt[q])]
    p += [s_load_dwords(s_T.sub(12, 2), s_par, 4 * P_DELTA), s_waitcnt(lgkmcnt=0)]
    p += block_coords(s_bid, s_hidx if NQ == 1 else 0)
    pl = s_T.sub(12, 2)
    p += [s_mul_i32(s_t[2], s_t[2], s_L), s_add_u32(s_t[2], s_t[2], s_t[0]), s_mul_i32(s_t[4], s_wave, 32 * NQ), s_add_u32(s_t[2], s_t[2], s_t[4]), s_lshl_b32(s_t[2], s_t[2], 2),
          s_add_u32(pl[0], pl[0], s_t[2]), s_addc_u32(pl[1], pl[1], 0)]
    for k_, vals in enumerate((dlt, nl, nd)):
        if k_:
            p += [s_add_u32(pl[0], pl[0], s_planeB), s_addc_u32(pl[1], pl[1], 0)]
        for q in range(NQ):
            p += [global_store_dword(loff, vals[q], pl, 128 * q)]
    for q in range(NQ):
        for r in range(16):
            p += [v_mov_b32(negl[16 * q + r], nl[q]), v_mov_b32(negd[16 * q + r], nd[q])]
    for ins in p:
        if ins.kind in ("valu", "trans", "dot"):
            ins.meta["keep"] = True
    p += [s_nop(1)]
    # ---- S(0) into buffer 0 (its registers held the O rows of q = 0: consumed above), then the first V fragments of step 0
    G = Gaps(nm)
    s_group(G, 0, 0, 0)
    post = [x for u in v_prefetch(0) for x in u] + [s_nop(15)] + [x for u in exp_units(0, 0) for x in u]      # (chunk 0 of S(0): what a step's dQ group does for the next)
    return p, G, post


def epilogue():
    """dQ^T -> bf16, scaled by ln 2 (the un-folding of the pre-scaled q), stored as whole 128-byte row segments (attn_dkv64.epi_addresses / epi_block); neither the ring
    nor the staging area (the next block's O rows are landing there) is touched"""
    e = [comment("---- epilogue")] + stamp(40)
    e += [s_nop(15), s_nop(15)]
    e += [s_load_dwords(s_T.sub(12, 4), s_par, 4 * P_DQ), s_waitcnt(lgkmcnt=0)]
    dq, dqstr, scale = s_T.sub(12, 2), s_T[14], s_T[15]
    t = [Fq[i] for i in range(6)]
    xb, rdaddr = Fq[6], Fq[7]
    go = [Fq[8 + i] for i in range(4)]
    ta = [Fq[16], Fq[17]]
    vals = [dPb[i] for i in range(4)]
    pk = [dPb[4 + i] for i in range(2)]
    rb = dPb.sub(8, 16)
    e += _epi_addresses(t, E_BASE, xb, rdaddr, [go], [dqstr], 32 * NQ, sregs=(s_t, s_wave, s_lds))
    e += block_coords(s_bid, s_hidx if NQ == 1 else 0)
    e += [s_add_u32(s_t[1], s_t[1], s_t[0])]
    out = s_kn          # (recomputed by the next block start)
    e += ptr(out, s_t[1], dq, dqstr)
    for q in range(NQ):
        e += epi_block(lambda i, q=q: dQblk(q, i), scale, vals, pk, ta, xb, rdaddr, rb, go, out)
        if q == 0 and NQ == 2:
            e += [s_lshl_b32(s_t[2], dqstr, 5), s_add_u32(out[0], out[0], s_t[2]), s_addc_u32(out[1], out[1], 0)]
    e += stamp(41)
    return e


def timeline_store():
    if not (ABL & 16):
        return []
    t = tmp
    p = v_mbcnt_lane_id(t[0])
    p += [s_nop(0)]
    return p + [s_lshl_b32(s_t[4], s_wg, 2), s_add_u32(s_t[4], s_t[4], s_wave), s_lshl_b32(s_t[4], s_t[4], 8), v_lshlrev_b32(t[0], 2, t[0]), v_add_u32(t[0], s_t[4], t[0]),
                v_readlane_b32(s_T[0], v_tl, 62), v_readlane_b32(s_T[1], v_tl, 63), s_nop(4), global_store_dword(t[0], v_tl, s_T.sub(0, 2), 0), s_waitcnt(vmcnt=0)]


def block_program(nq, suf):
    global NQ
    NQ = nq
    nm = 8 * NQ
    counts = {}
    head_code, G0, post = block_start()
    pend = []
    for _ in range(3):
        _, pend = auto_waits(body(0, "main").flat("probe"), pend)
    start, pend_s = finish(G0.flat("S(0)") + post, [])
    assert pend_s == pend, (pend_s, pend)
    prog = head_code + start
    b, pe = finish(body(0, "head").flat("step 0 (head)"), pend)
    assert pe == pend
    prog += b + [s_branch("L_body1"), label("L_loop")]
    for j in range(4):
        G = body(j, "main")
        b, pe = finish(G.flat(f"step j={j}"), pend)
        assert pe == pend, (j, pe, pend)
        if j == 1:
            prog += [label("L_body1")]
        prog += b
        counts[f"main{j}{suf}"] = G.costs()
    prog += [s_sub_u32(s_loop, s_loop, 1), s_cmp_lg_u32(s_loop, 0), s_cbranch_scc1("L_loop")]
    cur = pend
    for j in range(4):
        G = body(j, f"tail{j}")
        post = []
        if j == 1:
            # the next block's operand bases; its Q rows leave for the staging area (behind the ring's refill of this step)
            if NQ == 1:      # (a half block is the last thing a workgroup does: its "next" block is itself, a half)
                G.pre = seam_ptrs(s_nbid, s_hidxn, True) + G.pre
            else:
                G.pre = [s_cmp_eq_u32(s_moden, 1), s_cselect_b32(s_t[5], 1, 0)] + seam_ptrs(s_nbid, s_hidxn, s_t[5]) + G.pre
            G.spread(stage_dma(s_T.sub(0, 2), s_T[8], 0, NQ, rewind=True), 2 * nm + (11 if NQ == 2 else 6), 3 * nm - 1)
        if j == 2:
            # Q rows -> Q registers, behind the block's last S group (their last reader) and this step's refill of the ring (4 LDS-DMAs younger than the rows')
            G.put(2 * nm + (11 if NQ == 2 else 6), [s_waitcnt(vmcnt=4)])
            G.spread(stage_reads(Qfr, 0, NQ), 2 * nm + (11 if NQ == 2 else 6), 3 * nm - 1)
        if j == 3:
            # dO rows leave (the Q rows have been read: drained first), lse; at the end of the step, behind the last dP group, dO rows -> dO registers and the O rows leave:
            # they have the epilogue to land
            G.put(1, [s_waitcnt(lgkmcnt=0)])
            G.spread(stage_dma(s_T.sub(2, 2), s_T[9], 0, NQ, rewind=True) + lse_loads(), 1, nm)
            post = [s_waitcnt(vmcnt=4)] + [x for u in stage_reads(dOfr, 0, NQ) for x in u] + [s_waitcnt(lgkmcnt=0)]
            post += [x for u in stage_dma(s_T.sub(4, 2), s_T[10], 0, NQ, rewind=True) for x in u]
        b, pe = finish(G.flat(f"step tail{j}"), cur)
        if j < 2:
            assert pe == pend, (j, pe, pend)
        cur = pe        # (tail2 ends with the Q rows' reads in the queue: tail3 drains them before the dO rows' LDS-DMA goes out)
        if j == 3:
            b += [s_waitcnt(lgkmcnt=0)] + post
        prog += b
        counts[f"tail{j}{suf}"] = G.costs()
    prog += epilogue()
    NQ = 2

    def ren(ins):
        if ins.kind == "label":
            return label(ins.meta["name"] + suf)
        if ins.kind == "branch" and ins.meta["target"] not in ("L_end", "L_done", "L_block_F", "L_block_H"):
            return {None: s_branch, 0: s_cbranch_scc0, 1: s_cbranch_scc1}[ins.meta["cond"]](ins.meta["target"] + suf)
        return ins
    return [ren(i) for i in prog], counts


def build():
    prog = entry()
    full, counts = block_program(2, "_F")
    half, counts_h = block_program(1, "_H")
    counts.update(counts_h)
    after = [s_cmp_eq_u32(s_moden, 2), s_cbranch_scc1("L_done"), s_mov_b32(s_bid, s_nbid), s_mov_b32(s_hidx, s_hidxn),
             s_cmp_eq_u32(s_moden, 1), s_cbranch_scc1("L_block_H"), s_branch("L_block_F")]
    prog += full + after + half
    prog += [label("L_done"), s_waitcnt(vmcnt=0)] + stamp(42) + timeline_store() + [s_branch("L_end")]
    prog += [label("L_end"), Inst("s_endpgm", "end", final=True)]
    return prog, counts


def emit_one(f, suffix):
    prog, counts = build()
    lines = []
    for ins in prog:
        if ins.kind == "comment":
            continue
        t = ins.text
        if ins.kind == "label":
            t = t[:-1] + "_%=:"
        elif ins.kind == "branch":
            op, tgt = t.split()
            t = f"{op} {tgt}_%="
        elif ins.kind == "end":
            if ins.meta.get("final"):
                continue
            t = "s_branch L_end_%="
        lines.append(t)
    f.write(f"#define UDM_DQ64_ASM{suffix} \\\n")
    for t in lines:
        f.write(f'  "{t}\\n\\t" \\\n')
    f.write('  ""\n')
    return prog, counts


def emit(path, ablations=()):
    global ABL
    with open(path, "w") as f:
        f.write("// GENERATED by asmgen/attn_dq64.py - do not edit.  The whole persistent attention-backward dQ workgroup program as ONE asm statement.\n")
        f.write(f"#define UDM_DQ64_LDS_BYTES {LDS_TOTAL}\n")
        f.write(f"#define UDM_DQ64_PARAM_DWORDS {PARAM_DWORDS}\n")
        clob = [f'"v{i}"' for i in range(255)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(36, 100)] + ['"vcc"', '"scc"', '"m0"', '"memory"']
        f.write("#define UDM_DQ64_CLOBBERS " + ", ".join(clob) + "\n")
        ABL = 0
        prog, counts = emit_one(f, "")
        for a in ablations:
            ABL = a
            emit_one(f, f"_ABL{a}")
        ABL = 0
    return prog, counts


if __name__ == "__main__":
    prog, counts = emit(sys.argv[1] if len(sys.argv) > 1 else "attention_dq64_gen.h", [int(x) for x in sys.argv[2:]])
    probs = lint([i for i in prog if i.kind != "raw"], mfma_states=4)
    print(stats(prog))
    for k, c in counts.items():
        print(k, c)
    for x in probs[:40]:
        print("LINT", x)
    print(len(probs), "lint problems", "| sgprs up to", S_.next - 1, "| vgprs up to", V.next - 1)
