"""Generator of the one-wave-per-SIMD attention backward dK / dV pass for gfx950 (head dim 128, no mask, q pre-scaled, L % 256 == 0).

Replaces the dK / dV half of the backward of `flash_attn_qkvpacked_func` (reference models/dit.py:843) on the headline path; the wave-specialised 8-wave kernel
of attention_dkv_ws.hip keeps every other shape.  One workgroup = 4 waves = one wave per SIMD = 256 keys of one (batch, head); a wave owns TWO 32-key blocks
(f = 0, 1) and the whole 512-register file, so that every Q / dO fragment read from LDS feeds two MFMAs (the 8-wave kernel reads one fragment per MFMA).

    S  = Q K^T  (+ -lse as the C operand of the chain: the accumulator IS s - lse)        A = Q row fragment (LDS), B = K fragment (resident)
    dP = dO V^T (+ -delta as the C operand)                                               A = dO row fragment,      B = V fragment (resident)
    P  = exp2(S), dS = P (dP - delta)                                                     VALU, packed to bf16 (P in its own registers, dS in place over dP)
    dV^T += dO^T P,  dK^T += Q^T dS                                                       A = transposing reads of the SAME LDS tiles, B = P / dS from registers

Registers of a wave:
    a[0:255]    dK^T and dV^T accumulators, [f][32-column group i][16]
    v[0:127]    K and V fragments of the wave's 64 keys (MFMA B operands), [f][k-step][4], loaded once per block
    v[128:191]  S and dP of ONE 32-query step, [f][16] each (VGPRs: the VALU reads them)
    v[192:207]  P as packed bf16, [f][16-query chunk][4]
    v[208:239]  a ring of eight 4-register fragment slots: every LDS fragment is read ~7 MFMAs before its first use and dies two MFMAs later
    v[240:252]  addresses, LDS-DMA offsets (ring and staging area)

LDS: a ring of four stages = 32-query steps, each a Q tile and a dO tile in the piece layout of attn_fwd64.py (4-row pieces of 1040 bytes, [64-byte column chunk]
[row][64 bytes] inside a piece): the row fragments (`ds_read_b128`) and the transposed fragments (`ds_read_b64_tr_b16`) of the same tile are both conflict-free and
every fragment address is one per-lane base + an immediate.  -lse | -delta of a step (the planes the dQ pass leaves behind) travel with it (one 4-byte DMA per lane).

Step t of the loop (64 MFMAs, ONE barrier, no double buffering of S / dP):
    G0  16 MFMAs dP(t)          under them: exp2 of S(t), packing of P(t)
    G1  16 MFMAs dV^T += ..     under them: dS = P dP, packing of dS(t) (in place)
    G2  16 MFMAs dK^T += ..     under them: the -lse(t+1) C operands into the S registers, the LDS-DMA of step t+3
    G3  16 MFMAs S(t+1)         under them: the -delta(t+1) C operands into the dP registers
Every VALU stage sits a full group behind the MFMAs that produce its input and a full group ahead of those that consume its output.

PERSISTENT like the forward: a workgroup walks blocks id, id + grid, ... of its XCD; the last three steps of a block refill the ring with the NEXT block's first steps; when
the blocks behind the whole rounds are half a grid every workgroup ends with one 128-key half block (a wave owns one 32-key block: the NF = 1 program).  The next block's K / V
rows do not come as fragment loads (lane = row: 32 cache lines per instruction, ~200 cycles of a lone wave each) but as coalesced LDS-DMA pieces into a per-wave staging area
behind the ring: K rows leave two steps before the block ends and are read into the K registers at the top of its last step, V rows follow through the same area and are read
under the next block's S(0) group; dK / dV leave through 4 KiB of LDS per wave as whole 128-byte row segments (epi_addresses / epi_block, shared with attn_dq64.py).

`s_waitcnt lgkmcnt` is not written by hand: `auto_waits` walks a straight-line piece, tracks the LDS queue and puts the counted wait in front of the first reader of
every fragment; `isa.lint` checks the software-visible hazards and tests/test_asmgen.py executes the stream on the CPU emulator (late and early memory)."""
import sys
from isa import *   # noqa: F401,F403

D, SUBQ, KS = 128, 32, 8
NST, PD = 4, 3
PIECE = 1040
TILE = 8 * PIECE              # 32 rows
STG = 2 * TILE                # Q tile | dO tile
LD_BASE = NST * STG           # [NST][-lse 32 f32 | -delta 32 f32]
LDS_RING = LD_BASE + NST * 256
# Behind the ring: ONE area of two tiles per wave.  The next block's K rows, then its V rows pass through it as LDS-DMA pieces (coalesced 256-byte rows; a direct
# fragment load touches 32 cache lines per instruction and stalled the issue ~200 cycles each: 32 of them cost 7 k cycles per block, first timeline) and are picked up
# as row fragments (the V rows under the next block's S(0) group: their LDS-DMA has the whole epilogue to land).
A_BASE = LDS_RING
AREA = 2 * TILE
E_BASE = A_BASE + 4 * AREA    # epilogue staging: 4 KiB per wave (32 rows x 64 columns at a time) - NOT inside the area: the next block's V rows wait there through the epilogue
E_STG = 4096
LDS_TOTAL = E_BASE + 4 * E_STG

# ---- the kernel's parameter block (kernarg segment; attention_dkv64.hip declares the same struct): dword offsets
P_Q, P_DO, P_NLSE, P_QSTR, P_DOSTR, P_L, P_NSTEPS, P_H, P_NT, P_MG_NT, P_MG_H, P_NFULL, P_HASHALF, P_GSTRIDE, P_PLANEB = 0, 2, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17
P_K, P_V, P_KSTR, P_VSTR, P_DK, P_DV, P_DKSTR, P_DVSTR, P_SCALE, P_TL = 18, 20, 22, 23, 24, 26, 28, 29, 30, 32
PARAM_DWORDS = 34

NF = 2        # 32-key blocks per wave of the program being generated: 2 (256-key blocks) or 1 (the 128-key half blocks that balance the walk)

V = Alloc("v", 0, 255)     # v255 is left to the compiler (the thread-id operand)
A = Alloc("a", 0, 256)
S_ = Alloc("s", 36, 100)

Kf, Vf = V("Kf", 64), V("Vf", 64)
Sb, dPb = V("S", 32, 2), V("dP", 32, 2)
Ppk = V("Ppk", 16)
Fq = V("Fq", 32)
qaddr, taddr, laddr = V("qaddr"), V("taddr"), V("laddr")
dq, do_ = V("dq", 2), V("do", 2)
dl = V("dl")
aaddr = V("aaddr")            # row-fragment base inside this wave's staging area
da = V("da", 4)               # LDS-DMA source offsets of the operand being staged: pieces 4 g + j (also scratch of the entry: the thread id arrives in da[3])
tmp = da
dKT, dVT = A("dK", 128), A("dV", 128)

# persistent parameters: dwords 0 .. 17 of the block, loaded by two s_load into s36 .. s53 (in this order)
s_qb, s_dob, s_nlse = S_("qb", 2, 4), S_("dob", 2, 2), S_("nlse", 2, 2)
s_qstr, s_dostr, s_L, s_nsteps, s_H, s_nt, s_mg_nt, s_mg_H, s_nfull, s_hashalf, s_gstride, s_planeB = (S_(n) for n in (
    "qstr", "dostr", "L", "nsteps", "H", "nt", "mg_nt", "mg_H", "nfull", "hashalf", "gstride", "planeB"))
s_par = S_("par", 2, 2)
s_bid, s_lds = S_("bid"), S_("lds")
s_tm = S_("tm", 2, 2)
s_T = S_("T", 8, 4)            # transient parameters: k, v, strides (block start / seam) or dk, dv, strides, scale (epilogue)
s_wave = S_("wave")
s_loop = S_("loop")
s_qt, s_dot, s_lt = S_("qt", 2, 2), S_("dot", 2, 2), S_("lt", 2, 2)     # running bases of the refills
s_qstep, s_dostep = S_("qstep"), S_("dostep")
s_qdst, s_odst, s_ldst = S_("qdst"), S_("odst"), S_("ldst")
s_t = [S_("t0", 1, 2)] + [S_(f"t{i}") for i in range(1, 6)]      # (t4 : t5 serve as a pointer pair)
s_qn, s_don, s_ltn = S_("qn", 2, 2), S_("don", 2, 2), S_("ltn", 2, 2)   # the next block's operand bases
s_nbid, s_moden, s_hidx, s_hidxn, s_wg = (S_(n) for n in ("nbid", "moden", "hidx", "hidxn", "wg"))
s_wk = S_("wk")                 # keys per wave of the block whose K / V are being loaded: 64 or 32
assert S_.next <= 100, S_.next
assert (s_qb.idx, s_planeB.idx) == (36, 53)

ABL = 0     # timing-only ablations (WRONG results): 1 = no softmax VALU, 2 = no fragment reads, 4 = no refills / barriers, 8 = no MFMAs; 16 = timeline stamps (correct)
v_tl = R("v", 254)


def stamp(idx):
    """timeline builds: s_memtime -> lane idx of v_tl (the half-block program's stamps of the block sit 16 lanes behind the whole-block program's: both stay readable)"""
    if not (ABL & 16):
        return []
    if 3 <= idx < 42 and NF == 1:
        idx += 16
    return [s_memtime(s_tm), s_waitcnt(lgkmcnt=0), v_writelane_b32(v_tl, s_tm[0], idx)]


# ---- named pieces of the register file -------------------------------------------------------------------------------------------------
def Kfr(f, ks):
    return Kf.sub((f * KS + ks) * 4, 4)


def Vfr(f, ks):
    return Vf.sub((f * KS + ks) * 4, 4)


def Sblk(f):
    return Sb.sub(16 * f, 16)


def dPblk(f):
    return dPb.sub(16 * f, 16)


def Pfr(f, c2):
    return Ppk.sub((f * 2 + c2) * 4, 4)


def dSfr(f, c2):
    """packed bf16 dS of 16 queries: written IN PLACE over the first four of the eight fp32 values it was made from"""
    return dPblk(f).sub(8 * c2, 4)


def slot(n):
    return Fq.sub(4 * (n % 8), 4)


def dKblk(f, i):
    return dKT.sub((f * 4 + i) * 16, 16)


def dVblk(f, i):
    return dVT.sub((f * 4 + i) * 16, 16)


# ---- LDS reads ----------------------------------------------------------------------------------------------------------------------------
def row_frag(n, tile, ks):
    """k-step ks of the row fragments of the 32 x 128 tile at byte `tile`: lane (row l & 31, 8 columns 16 ks + 8 (l >> 5) ..)"""
    return [ds_read_b128(slot(n), qaddr, tile + (ks >> 1) * 256 + (ks & 1) * 32)]


def tr_frag(n, tile, i, c2):
    """transposed fragment: lane (column 32 i + (l & 31)), 8 rows 16 c2 + 4 (l >> 5) + {0..3, 8..11} - the k order of the accumulator registers 8 c2 .. 8 c2 + 7"""
    return [ds_read_b64_tr_b16(slot(n).sub(2 * h2, 2), taddr, tile + (4 * c2 + 2 * h2) * PIECE + i * 256) for h2 in range(2)]


def c_loads(blk, stage, which):
    """-lse (which = 0) / -delta (1) of the step in `stage` -> all 16 registers of an accumulator block: register 4 rg + e of half h belongs to query 8 rg + 4 h + e"""
    return [ds_read_b128(blk.sub(4 * rg, 4), laddr, stage * 256 + which * 128 + 32 * rg) for rg in range(4)]


# ---- LDS-DMA of one step -----------------------------------------------------------------------------------------------------------------
def dma_step(stage):
    """this wave's five pieces of a step (two of Q, two of dO, -lse | -delta) and the advance of the running bases.  Units (glued sequences)."""
    u = []
    for j in range(2):
        u.append([s_add_u32(M0, s_qdst, stage * STG + j * PIECE), s_nop(0), global_load_lds_dwordx4(dq[j], s_qt)])
    for j in range(2):
        u.append([s_add_u32(M0, s_odst, stage * STG + j * PIECE), s_nop(0), global_load_lds_dwordx4(do_[j], s_dot)])
    u.append([s_add_u32(M0, s_ldst, stage * 256), s_nop(0), global_load_lds_dword(dl, s_lt)])
    u.append([s_add_u32(s_qt[0], s_qt[0], s_qstep), s_addc_u32(s_qt[1], s_qt[1], 0)])
    u.append([s_add_u32(s_dot[0], s_dot[0], s_dostep), s_addc_u32(s_dot[1], s_dot[1], 0)])
    u.append([s_add_u32(s_lt[0], s_lt[0], 128), s_addc_u32(s_lt[1], s_lt[1], 0)])
    return u


# ---- gap lists ------------------------------------------------------------------------------------------------------------------------------
class Gaps:
    """MFMAs with the instructions issued behind each of them (gap g = behind MFMA g; `pre` = ahead of the first)"""

    def __init__(self, n):
        self.m = [None] * n
        self.f = [[] for _ in range(n)]
        self.pre = []

    def put(self, g, insts):
        (self.pre if g < 0 else self.f[min(g, len(self.f) - 1)]).extend(insts)

    def spread(self, units, g0, g1):
        """units (instruction lists that stay together) in order, evenly over gaps g0 .. g1"""
        n = len(units)
        for k, u in enumerate(units):
            self.put(g0 + (k * (g1 - g0 + 1)) // n, u)

    def flat(self, note):
        out = [comment(note)] + list(self.pre)
        for m, f in zip(self.m, self.f):
            out.append(m)
            out += f
        return out

    def costs(self):
        c = lambda i: {"trans": 1.5, "dma": 1.5, "label": 0.0, "comment": 0.0}.get(i.kind, 1.0)
        return [round(sum(c(i) for i in f), 1) for f in self.f]


def auto_waits(prog, pending):
    """Straight-line piece -> the same piece with `s_waitcnt lgkmcnt(n)` in front of the first instruction that touches a register an outstanding LDS read
    writes.  `pending`: the LDS queue on entry (register sets, oldest first).  Returns (piece, queue on exit).  LDS operations return in order, so waiting for
    one drains everything older."""
    out, q = [], [set(p) for p in pending]
    for ins in prog:
        if ins.kind == "wait" and ins.meta.get("lgkmcnt") is not None:
            n = ins.meta["lgkmcnt"]
            q = q[len(q) - n:] if n < len(q) else q
            out.append(ins)
            continue
        touched = set(ins.reads) | set(ins.writes)
        hit = max((k for k, regs in enumerate(q) if regs & touched), default=None)
        if hit is not None:
            n = len(q) - 1 - hit
            out.append(s_waitcnt(lgkmcnt=min(n, 15)))
            q = q[len(q) - min(n, 15):] if min(n, 15) > 0 else []
        if ins.kind in ("barrier",) and False:
            pass
        out.append(ins)
        if ins.kind in ("lds_rd", "smem"):
            q.append(set(ins.writes))
        elif ins.kind == "lds_wr":
            q.append(set())
    return out, [sorted(x) for x in q]


def _abl(prog):
    out = []
    for ins in prog:
        k = ins.kind
        if (ABL & 1) and k in ("valu", "trans") and not ins.meta.get("keep"):
            continue
        if (ABL & 2) and k == "lds_rd":
            continue
        if (ABL & 4) and (k in ("dma", "barrier") or (k == "wait" and ins.meta.get("vmcnt") is not None) or
                          (k == "salu" and any(w in ins.writes for w in [("m0", 0)] + s_qt.regs() + s_dot.regs() + s_lt.regs()))):
            continue
        if (ABL & 4) and k == "nop":
            continue
        if (ABL & 8) and k == "mfma":
            continue
        out.append(ins)
    return out


# ---- one step ---------------------------------------------------------------------------------------------------------------------------------
def LA():
    """how many MFMAs ahead of its first use a fragment is read"""
    return 7 if NF == 2 else 4


def n_pref():
    """fragments of a group whose reads fall into the previous group"""
    return (LA() + NF - 1) // NF


def softmax_units():
    """exp2 of S(t) and the packing of P(t), in the order the dV MFMAs need them: per 16-query chunk c2, per key block f: 8 exp, then 4 packs"""
    halves = []
    for c2 in range(2):
        u = []
        for f in range(NF):
            s = Sblk(f)
            u += [[v_exp_f32(s[8 * c2 + e], s[8 * c2 + e])] for e in range(8)]
        for f in range(NF):
            s = Sblk(f)
            u += [[v_cvt_pk_bf16_f32(Pfr(f, c2)[j], s[8 * c2 + 2 * j], s[8 * c2 + 2 * j + 1])] for j in range(4)]
        halves.append(u)
    return halves


def ds_units():
    """dS = P (dP - delta): two products per instruction, then packed in place"""
    halves = []
    for c2 in range(2):
        u = []
        for f in range(NF):
            s, d = Sblk(f), dPblk(f)
            u += [[v_pk_mul_f32(d.sub(8 * c2 + 2 * j, 2), s.sub(8 * c2 + 2 * j, 2), d.sub(8 * c2 + 2 * j, 2))] for j in range(4)]
        for f in range(NF):
            d = dPblk(f)
            u += [[v_cvt_pk_bf16_f32(d[8 * c2 + j], d[8 * c2 + 2 * j], d[8 * c2 + 2 * j + 1])] for j in range(4)]
        halves.append(u)
    return halves


def s_group(G, g0, stage_next, prefetch_do=True):
    """the S(t+1) group at MFMA index g0: MFMAs, the Q row fragments that are not prefetched, the -delta(t+1) C operands, the first dO row fragments of the next step"""
    nm = 8 * NF
    QTn, OTn = stage_next * STG, stage_next * STG + TILE
    g = g0
    for ks in range(KS):
        for f in range(NF):
            G.m[g] = v_mfma_f32_32x32x16_bf16(Sblk(f), slot(24 + ks), Kfr(f, ks), Sblk(f))
            g += 1
    for ks in range(n_pref(), KS):
        G.put(g0 + ks * NF - LA(), row_frag(24 + ks, QTn, ks))
    # -delta: the dP registers hold dS(t) until the last dK MFMA (index g0 - 1) has been issued
    dl_ = [c_loads(dPblk(f), stage_next, 1) for f in range(NF)]
    G.spread([[x] for f in range(NF) for x in dl_[f]], g0 + (2 if NF == 2 else 1), g0 + (9 if NF == 2 else 4))
    if prefetch_do:
        for n in range(n_pref()):
            G.put(g0 + nm + n * NF - LA(), row_frag(n, OTn, n))


def body(j, variant):
    """step t with t % 4 == j.  variant: 'main'; 'head' (the block's first step: the accumulators start from the constant 0); 'tail0' .. 'tail3' (the block's last four
    steps: tail1 .. tail3 refill the ring with the next block's steps 0 .. 2, tail3 has no S(t+1) and loads the next block's K / V fragments)"""
    nm = 8 * NF
    last = variant == "tail3"
    st, stn, rst = j, (j + 1) % NST, (j + PD) % NST
    QT, OT = st * STG, st * STG + TILE
    G = Gaps(3 * nm if last else 4 * nm)
    # ---------------- MFMAs of G0 .. G2
    g = 0
    for ks in range(KS):
        for f in range(NF):
            G.m[g] = v_mfma_f32_32x32x16_bf16(dPblk(f), slot(ks), Vfr(f, ks), dPblk(f))
            g += 1
    for c2 in range(2):
        for i in range(4):
            for f in range(NF):
                acc = dVblk(f, i)
                G.m[g] = v_mfma_f32_32x32x16_bf16(acc, slot(8 + c2 * 4 + i), Pfr(f, c2), 0 if (variant == "head" and c2 == 0) else acc)
                g += 1
    for c2 in range(2):
        for i in range(4):
            for f in range(NF):
                acc = dKblk(f, i)
                G.m[g] = v_mfma_f32_32x32x16_bf16(acc, slot(16 + c2 * 4 + i), dSfr(f, c2), 0 if (variant == "head" and c2 == 0) else acc)
                g += 1
    # ---------------- G0: wait + barrier, dO row fragments, softmax
    G.pre += stamp((8 if variant in ("main", "head") else 12) + j)       # timeline builds: the top of the step (drains the LDS queue: the queue state stays the loop's)
    G.put(0, [s_waitcnt(vmcnt=5 + 8 * NF if variant == "tail2" else 5)])      # (tail2: the K rows' LDS-DMA of the previous step is in the queue behind the ring's)
    G.put(1, [s_barrier()])
    for ks in range(n_pref(), KS):
        G.put(ks * NF - LA(), row_frag(ks, OT, ks))
    h0, h1 = softmax_units()
    if NF == 2:
        G.spread(h0, 3, 11)
        G.spread(h1, 12, 21)
    else:
        G.spread(h0, 2, 5)
        G.spread(h1, 6, 10)
    # transposed fragments of dO (dV group) and Q (dK group)
    for c2 in range(2):
        for i in range(4):
            n = 8 + c2 * 4 + i
            G.put(n * NF - LA(), tr_frag(n, OT, i, c2))
            n = 16 + c2 * 4 + i
            G.put(n * NF - LA(), tr_frag(n, QT, i, c2))
    # ---------------- G1: dS
    d0, d1 = ds_units()
    if NF == 2:
        G.spread(d0 + d1, 19, 36)
    else:
        G.spread(d0, 10, 14)
        G.spread(d1, 15, 18)
    # ---------------- G2: refill of step t + 3, the -lse(t+1) C operands, the first Q(t+1) row fragments
    if variant == "tail1":
        G.put(2 * nm, [s_mov_b64(s_qt, s_qn), s_mov_b64(s_dot, s_don), s_mov_b64(s_lt, s_ltn)])
    G.spread(dma_step(rst), 2 * nm + (8 if NF == 2 else 4), 3 * nm - 1)
    if not last:
        ll = [c_loads(Sblk(f), stn, 0) for f in range(NF)]
        G.spread([[x] for f in range(NF) for x in ll[f]], 2 * nm + (5 if NF == 2 else 2), 2 * nm + (12 if NF == 2 else 4))
        for n in range(n_pref()):
            G.put(3 * nm + n * NF - LA(), row_frag(24 + n, stn * STG, n))
        s_group(G, 3 * nm, stn)
    return G


def finish(prog, pending):
    prog, pend = auto_waits(prog, pending)
    return (_abl(prog) if ABL else prog), pend


# ---- block id -> pointers (SALU only; needs (B H) % 8 == 0) ------------------------------------------------------------------------------------
def block_coords(bid, hidx):
    """-> t0 = first key row of the block inside its sequence, t1 = b L, t2 = b H + h, t3 = h * 256 bytes"""
    t0, t1, t2, t3 = s_t[0], s_t[1], s_t[2], s_t[3]
    p = [s_lshr_b32(t0, bid, 3), s_and_b32(t1, bid, 7), s_mul_hi_u32(t2, t0, s_mg_nt), s_mul_i32(t3, t2, s_nt), s_sub_u32(t0, t0, t3),
         s_lshl_b32(t2, t2, 3), s_add_u32(t2, t2, t1),
         s_mul_hi_u32(t1, t2, s_mg_H), s_mul_i32(t3, t1, s_H), s_sub_u32(t3, t2, t3), s_lshl_b32(t3, t3, 8),
         s_mul_i32(t1, t1, s_L), s_lshl_b32(t0, t0, 8)]
    if not (isinstance(hidx, int) and hidx == 0):
        p += [s_lshl_b32(s_t[5], hidx, 7), s_add_u32(t0, t0, s_t[5])]
    return p


def ptr(dst, rows, base, stride):
    """dst = base + rows * stride + h * 256   (rows: an SGPR; t3 = h * 256 from block_coords)"""
    return [s_mul_i32(dst[0], rows, stride), s_mul_hi_u32(dst[1], rows, stride), s_add_u32(dst[0], dst[0], s_t[3]), s_addc_u32(dst[1], dst[1], 0),
            s_add_u32(dst[0], dst[0], base[0]), s_addc_u32(dst[1], dst[1], base[1])]


def stream_ptrs(bid, hidx, q, do, lt):
    """the Q / dO / -lse bases of the block's (batch, head): query row 0"""
    p = block_coords(bid, hidx)
    p += ptr(q, s_t[1], s_qb, s_qstr) + ptr(do, s_t[1], s_dob, s_dostr)
    p += [s_mul_i32(s_t[2], s_t[2], s_L), s_lshl_b32(s_t[2], s_t[2], 2), s_add_u32(lt[0], s_nlse[0], s_t[2]), s_addc_u32(lt[1], s_nlse[1], 0)]
    return p


def seam_ptrs(bid, hidx, is_half):
    """the K / V bases of block (bid, hidx) AT THIS WAVE'S FIRST KEY ROW into s_T: [0:1] k, [2:3] v, [4] kstr, [5] vstr.
    is_half: an SGPR that is 1 when the block is a 128-key half (a wave then owns 32 keys), or a bool."""
    p = [s_load_dwords(s_T.sub(0, 4), s_par, 4 * P_K), s_load_dwords(s_T.sub(4, 2), s_par, 4 * P_KSTR), s_waitcnt(lgkmcnt=0)]
    k, v, kstr, vstr = s_T.sub(0, 2), s_T.sub(2, 2), s_T[4], s_T[5]
    if isinstance(is_half, bool):
        p += [s_mov_b32(s_wk, 32 if is_half else 64)]
    else:
        p += [s_lshl_b32(s_wk, is_half, 5), s_sub_u32(s_wk, 64, s_wk)]
    p += block_coords(bid, hidx)
    p += [s_mul_i32(s_t[4], s_wave, s_wk), s_add_u32(s_t[1], s_t[1], s_t[0]), s_add_u32(s_t[1], s_t[1], s_t[4])]      # this wave's first key row (global row index)
    for base, stride in ((k, kstr), (v, vstr)):
        p += [s_mul_i32(s_t[4], s_t[1], stride), s_mul_hi_u32(s_t[5], s_t[1], stride), s_add_u32(s_t[4], s_t[4], s_t[3]), s_addc_u32(s_t[5], s_t[5], 0),
              s_add_u32(base[0], base[0], s_t[4]), s_addc_u32(base[1], base[1], s_t[5])]
    for ins in p:
        ins.meta["keep"] = True
    return p


def stage_dma(base, stride, ntiles, rewind=False):
    """`ntiles` x 32 rows from `base` (advanced past them) into this wave's staging area, as units: piece 4 g + j = rows 16 g + 4 j .. + 3; lane i of a piece reads row
    (i >> 2) & 3, 16 bytes at (4 (i >> 4) + (i & 3)) * 16.  Uses t0 (LDS destination), t1 (16 rows in bytes) until its last unit has been issued.  rewind (the program
    of whole blocks staging for a block that may be a half, s_wk = 32): the second tile re-reads the first tile's rows instead of running past the wave's 32."""
    pre = v_mbcnt_lane_id(da[3])
    pre += [v_lshrrev_b32(da[2], 2, da[3]), v_and_b32(da[2], 3, da[2]), v_lshrrev_b32(da[1], 4, da[3]), v_lshlrev_b32(da[1], 2, da[1]), v_and_b32(da[3], 3, da[3]),
            v_add_u32(da[1], da[1], da[3]), v_lshlrev_b32(da[1], 4, da[1]), v_mul_lo_u32(da[0], da[2], stride), v_add_u32(da[0], da[0], da[1]), s_lshl_b32(s_t[1], stride, 2)]
    pre += [v_add_u32(da[j], s_t[1], da[j - 1]) for j in range(1, 4)]
    pre += [s_mul_i32(s_t[0], s_wave, AREA), s_add_u32(s_t[0], s_t[0], s_lds), s_add_u32(s_t[0], s_t[0], A_BASE), s_lshl_b32(s_t[1], stride, 4)]
    for ins in pre:
        ins.meta["keep"] = True
    units = [pre]
    for g in range(2 * ntiles):
        for j in range(4):
            units.append([s_add_u32(M0, s_t[0], (4 * g + j) * PIECE), s_nop(0), global_load_lds_dwordx4(da[j], base)])
        units.append([s_add_u32(base[0], base[0], s_t[1]), s_addc_u32(base[1], base[1], 0)])
        if rewind and ntiles == 2 and g == 1:
            units.append([s_sub_u32(s_t[2], 64, s_wk), s_mul_i32(s_t[2], s_t[2], stride), s_sub_u32(base[0], base[0], s_t[2]), s_subb_u32(base[1], base[1], 0)])
    return units


def stage_reads(dst, ntiles):
    """the staged rows as row fragments: dst(f, ks) <- lane (row l & 31 of tile f, 8 columns 16 ks + 8 (l >> 5) ..)"""
    return [[ds_read_b128(dst(f, ks), aaddr, f * TILE + (ks >> 1) * 256 + (ks & 1) * 32)] for f in range(ntiles) for ks in range(KS)]


def next_block_choice():
    """What this workgroup does after the current block (s_moden: 0 = a whole block, 1 = its 128-key half, 2 = nothing) - the current block again when there is none
    (its prefetches then re-read valid memory and are never used) - and that block's Q / dO / -lse bases."""
    p = []
    if NF == 1:     # a half block is always the last thing a workgroup does
        p += [s_mov_b32(s_moden, 2), s_mov_b32(s_nbid, s_bid), s_mov_b32(s_hidxn, s_hidx)]
    else:
        p += [s_mov_b32(s_hidxn, 0), s_mov_b32(s_moden, 0),
              s_add_u32(s_nbid, s_bid, s_gstride), s_cmp_lt_u32(s_nbid, s_nfull), s_cbranch_scc1("L_np"),
              s_mov_b32(s_moden, 2), s_mov_b32(s_nbid, s_bid), s_cmp_eq_u32(s_hashalf, 0), s_cbranch_scc1("L_np"),
              # workgroup 8 a + x (x = its XCD) takes half a & 1 of block nfull + 8 (a >> 1) + x: the block id stays congruent to the XCD
              s_mov_b32(s_moden, 1), s_lshr_b32(s_nbid, s_wg, 4), s_lshl_b32(s_nbid, s_nbid, 3), s_and_b32(s_t[0], s_wg, 7), s_add_u32(s_nbid, s_nbid, s_t[0]),
              s_add_u32(s_nbid, s_nbid, s_nfull), s_lshr_b32(s_hidxn, s_wg, 3), s_and_b32(s_hidxn, s_hidxn, 1),
              label("L_np")]
    return p + stream_ptrs(s_nbid, s_hidxn, s_qn, s_don, s_ltn)


# ---- entry --------------------------------------------------------------------------------------------------------------------------------------
INPUTS = ["par", "bid", "lds", "tid"]


def entry():
    p = [comment("---- entry: parameters, constants of the wave, the first block's K / V and first three steps")]
    raw = lambda t: Inst(t, "raw")
    p += [raw(f"s_mov_b64 {s_par}, %0"), raw(f"s_mov_b32 {s_bid}, %1"), raw(f"s_mov_b32 {s_lds}, %2")]
    tid = tmp[3]
    p += [raw(f"v_mov_b32 {tid}, %3")]
    p += [s_load_dwords(R("s", 36, 16), s_par, 0), s_load_dwords(R("s", 52, 2), s_par, 64), s_waitcnt(lgkmcnt=0)]
    if ABL & 16:
        p += [s_load_dwords(s_T.sub(0, 2), s_par, 4 * P_TL), s_waitcnt(lgkmcnt=0), v_mov_b32(v_tl, 0), s_nop(1), v_writelane_b32(v_tl, s_T[0], 62), v_writelane_b32(v_tl, s_T[1], 63)] + stamp(0)
    t = [Fq[i] for i in range(8)]
    p += [s_nop(0), v_lshrrev_b32(t[1], 6, tid), s_nop(0), v_readfirstlane_b32(s_wave, t[1]), v_and_b32(t[0], 63, tid)]
    lane_v, l31, hi = t[0], t[1], t[2]
    p += [v_and_b32(l31, 31, lane_v), v_lshrrev_b32(hi, 5, lane_v)]
    # ---- LDS-DMA source offsets: lane i -> column chunk c = i >> 4, row r = (i >> 2) & 3, 16-byte slot s = i & 3 of a piece's four rows;
    #      piece 2 wave + j of the tile: byte offset = (4 (2 wave + j) + r) * stride + (4 c + s) * 16
    p += [v_lshrrev_b32(t[3], 2, lane_v), v_and_b32(t[3], 3, t[3])]
    p += [s_lshl_b32(s_t[0], s_wave, 3), s_nop(0), v_add_u32(t[3], s_t[0], t[3])]                      # 8 wave + r
    p += [v_lshrrev_b32(t[4], 4, lane_v), v_lshlrev_b32(t[4], 2, t[4]), v_and_b32(t[5], 3, lane_v), v_add_u32(t[4], t[4], t[5]), v_lshlrev_b32(t[4], 4, t[4])]
    for j in range(2):
        p += [v_add_u32(t[5], 4 * j, t[3]), v_mul_lo_u32(dq[j], t[5], s_qstr), v_mul_lo_u32(do_[j], t[5], s_dostr)]
        p += [v_add_u32(dq[j], dq[j], t[4]), v_add_u32(do_[j], do_[j], t[4])]
    # -lse | -delta: lane i < 32 reads -lse[row i], lane i >= 32 reads -delta[row i - 32] (the plane behind)
    p += [v_lshlrev_b32(dl, 2, l31), v_mul_lo_u32(t[5], hi, s_planeB), v_add_u32(dl, dl, t[5])]
    p += [s_lshl_b32(s_qstep, s_qstr, 5), s_lshl_b32(s_dostep, s_dostr, 5)]
    p += [s_mul_i32(s_t[0], s_wave, 2 * PIECE), s_add_u32(s_qdst, s_lds, s_t[0]), s_add_u32(s_odst, s_qdst, TILE), s_add_u32(s_ldst, s_lds, LD_BASE)]
    # ---- fragment read bases
    #   rows: piece = l31 >> 2, row = l31 & 3:  lds + piece * 1040 + row * 64 + hi * 16
    p += [s_mov_b32(s_t[0], PIECE), v_lshrrev_b32(t[3], 2, l31), v_mul_lo_u32(t[3], t[3], s_t[0]), v_and_b32(t[4], 3, l31), v_lshlrev_b32(t[4], 6, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_lshlrev_b32(t[4], 4, hi), v_add_u32(t[3], t[3], t[4]), v_add_u32(qaddr, s_lds, t[3])]
    #   transposed: piece = hi (+ 4 c2 + 2 h2), row = (lane & 15) >> 2, 16-column half (lane >> 4) & 1, 4 columns (lane & 3)
    p += [v_mul_lo_u32(t[3], hi, s_t[0]), v_and_b32(t[4], 15, lane_v), v_lshrrev_b32(t[4], 2, t[4]), v_lshlrev_b32(t[4], 6, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_lshrrev_b32(t[4], 4, lane_v), v_and_b32(t[4], 1, t[4]), v_lshlrev_b32(t[4], 5, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_and_b32(t[4], 3, lane_v), v_lshlrev_b32(t[4], 3, t[4]), v_add_u32(t[3], t[3], t[4]), v_add_u32(taddr, s_lds, t[3])]
    #   -lse / -delta: lds + LD_BASE + hi * 16
    p += [v_lshlrev_b32(t[3], 4, hi), v_add_u32(t[3], s_lds, t[3]), v_add_u32(laddr, LD_BASE, t[3])]
    p += [s_mul_i32(s_t[0], s_wave, AREA), s_add_u32(s_t[0], s_t[0], A_BASE), s_nop(0), v_add_u32(aaddr, s_t[0], qaddr)]     # the fragment base inside this wave's area
    p += [s_mov_b32(s_wg, s_bid), s_mov_b32(s_hidx, 0)]
    for ins in p:
        ins.meta["keep"] = True
    # ---- the first block (always a whole one): K and V through the staging area into their registers, its first three steps
    p += stream_ptrs(s_bid, 0, s_qt, s_dot, s_lt)
    p += seam_ptrs(s_bid, 0, False)
    for u in stage_dma(s_T.sub(0, 2), s_T[4], 2):
        p += u
    for s in range(PD):
        for u in dma_step(s):
            p += u
    p += [s_waitcnt(vmcnt=3 * 5)]
    for u in stage_reads(Kfr, 2):
        p += u
    p += [s_waitcnt(lgkmcnt=0)]
    for u in stage_dma(s_T.sub(2, 2), s_T[5], 2):
        p += u
    p += [s_waitcnt(vmcnt=0)]       # (once per workgroup: the block start's counted wait assumes nothing but stores behind what it needs)
    p += stamp(1)
    return p


def block_start():
    """per block: what comes next, everything in flight has landed (the previous block's stores, this block's K / V, the ring's first steps), then the S(0) group"""
    nm = 8 * NF
    p = [comment("---- block start"), label("L_block")] + stamp(3)
    p += next_block_choice()
    # everything older than the previous block's stores (8 per 32-row block and output) has landed: the next steps of the ring, the V rows; a workgroup's first block
    # finds the queue drained by the entry
    p += [s_waitcnt(vmcnt=16 * NF), s_barrier()] + stamp(4)
    # trips through the four steady-state steps: nsteps / 4 - 1 (>= 1: the launcher takes L >= 256)
    p += [s_lshr_b32(s_loop, s_nsteps, 2), s_sub_u32(s_loop, s_loop, 1)]
    pre = []
    for f in range(NF):
        pre += c_loads(Sblk(f), 0, 0)
    for n in range(n_pref()):
        pre += row_frag(24 + n, 0, n)
    G = Gaps(nm)
    s_group(G, 0, 0)
    # (s_group indexes its gaps from the group's first MFMA: the fragments before gap 0 are the prefetched ones)
    G.pre = pre + G.pre
    # the V rows (staged under the previous block's last step, or by the entry) -> V registers: first read by the dP group of step 0.  Issued ahead of the group's last
    # Q fragment, so that the wait for that fragment drains them (the LDS queue at the loop's entry stays the loop's)
    G.spread(stage_reads(Vfr, NF), 0, (2 * 7 - LA() - 1) if NF == 2 else (7 - LA() - 1))
    return p, G


def epi_addresses(t, e_base, xb, rdaddr, gos, strides, rows_per_wave, sregs=None):
    """Addresses of the row-store epilogue (shared with attn_dq64.py).  A wave holds X^T of 32 rows (lane & 31 = row, registers walk the columns); it goes out 64 columns
    at a time through 4 KiB of LDS owned by the wave (rows of 128 bytes, 16-byte slots XOR-ed with the row) as whole 128-byte row segments.
    t: 6 scratch VGPRs; xb: write base (slot sl at xb ^ (sl << 4)); rdaddr: read-back base (instruction k at + k * 1024: rows 8 k + (lane >> 3), slot lane & 7);
    gos[n][k]: global offsets of output n, rows wave * rows_per_wave + 8 k + (lane >> 3) (+ (lane & 7) * 16).  Uses s_t[0 .. 2].
    sregs: (s_t, s_wave, s_lds) of the calling generator (default: this module's)."""
    s_t, s_wave, s_lds = sregs if sregs is not None else (globals()["s_t"], globals()["s_wave"], globals()["s_lds"])
    lane_v, l31, hi, g8 = t[0], t[1], t[2], t[3]
    e = v_mbcnt_lane_id(lane_v)
    e += [v_and_b32(l31, 31, lane_v), v_lshrrev_b32(hi, 5, lane_v)]
    e += [s_lshl_b32(s_t[0], s_wave, 12), s_add_u32(s_t[0], s_t[0], s_lds), s_add_u32(s_t[0], s_t[0], e_base)]
    e += [v_lshlrev_b32(xb, 7, l31), v_and_b32(t[4], 7, l31), v_lshlrev_b32(t[4], 4, t[4]), v_add_u32(xb, xb, t[4]), v_lshlrev_b32(t[4], 3, hi), v_add_u32(xb, xb, t[4]),
          v_add_u32(xb, s_t[0], xb)]
    e += [v_lshrrev_b32(g8, 3, lane_v), v_and_b32(t[4], 7, lane_v), v_xor_b32(t[5], t[4], g8), v_lshlrev_b32(t[5], 4, t[5]), v_lshlrev_b32(rdaddr, 7, g8), v_add_u32(rdaddr, rdaddr, t[5]),
          v_add_u32(rdaddr, s_t[0], rdaddr)]
    e += [s_mul_i32(s_t[1], s_wave, rows_per_wave), s_nop(0), v_add_u32(t[5], s_t[1], g8), v_lshlrev_b32(t[4], 4, t[4])]
    for go, stride in zip(gos, strides):
        e += [v_mul_lo_u32(go[0], t[5], stride), v_add_u32(go[0], go[0], t[4]), s_lshl_b32(s_t[2], stride, 3), s_nop(0)]
        e += [v_add_u32(go[k], s_t[2], go[k - 1]) for k in range(1, 4)]
    return e


def epi_block(acc, scale, vals, pk, ta, xb, rdaddr, rb, go, base):
    """one 32-row x 128-column block: acc(i) = the 16 accumulator registers of column group i; scale: an SGPR or None; vals: 4, pk: 2 (an even pair), ta: 2, rb: 16 VGPRs"""
    e = []
    n = 0
    for ch in range(2):
        for i in (2 * ch, 2 * ch + 1):
            for rg in range(4):
                o = acc(i)
                e += [v_accvgpr_read_b32(vals[k], o[rg * 4 + k]) for k in range(4)]
                if scale is not None:
                    e += [v_mul_f32(vals[k], scale, vals[k]) for k in range(4)]
                e += [v_cvt_pk_bf16_f32(pk[0], vals[0], vals[1]), v_cvt_pk_bf16_f32(pk[1], vals[2], vals[3])]
                e += [v_xor_b32(ta[n & 1], ((i & 1) * 4 + rg) << 4, xb), ds_write_b64(ta[n & 1], R("v", pk[0].idx, 2), 0)]
                n += 1
        for k in range(4):
            e += [ds_read_b128(rb.sub(k * 4, 4), rdaddr, k * 1024)]
        e += [s_waitcnt(lgkmcnt=0)]
        for k in range(4):
            e += [global_store_dwordx4(go[k], rb.sub(k * 4, 4), base, ch * 128)]
        e += [s_nop(1)]
    return e


def epilogue():
    """dK^T, dV^T -> bf16 (dK scaled by ln 2: the un-folding of the pre-scaled q), stored as whole 128-byte row segments (epi_addresses / epi_block).  Neither the ring (it
    holds the next block's first steps) nor the staging area (the next block's V rows) is touched."""
    e = [comment("---- epilogue")] + stamp(40)
    e += [s_nop(15), s_nop(15)]
    e += [s_load_dwords(s_T, s_par, 4 * P_DK), s_waitcnt(lgkmcnt=0)]
    dk, dv, dkstr, dvstr, scale = s_T.sub(0, 2), s_T.sub(2, 2), s_T[4], s_T[5], s_T[6]
    t = [Fq[i] for i in range(6)]
    xb, rdaddr = Fq[6], Fq[7]
    gos = [[Fq[8 + i] for i in range(4)], [Fq[12 + i] for i in range(4)]]
    ta = [Fq[16], Fq[17]]
    vals = [dPb[i] for i in range(4)]
    pk = [dPb[4 + i] for i in range(2)]
    rb = dPb.sub(8, 16)
    e += epi_addresses(t, E_BASE, xb, rdaddr, gos, [dkstr, dvstr], 32 * NF)
    e += block_coords(s_bid, s_hidx if NF == 1 else 0)
    e += [s_add_u32(s_t[1], s_t[1], s_t[0])]
    ok, ov = s_qn, s_don      # (recomputed by the next block start)
    e += ptr(ok, s_t[1], dk, dkstr) + ptr(ov, s_t[1], dv, dvstr)
    for which, (acc_of, base, stride, sc) in enumerate(((dKblk, ok, dkstr, scale), (dVblk, ov, dvstr, None))):
        for f in range(NF):
            e += epi_block(lambda i, f=f, acc_of=acc_of: acc_of(f, i), sc, vals, pk, ta, xb, rdaddr, rb, gos[which], base)
            if f == 0 and NF == 2:   # the f = 1 key rows: + 32 rows
                e += [s_lshl_b32(s_t[2], stride, 5), s_add_u32(base[0], base[0], s_t[2]), s_addc_u32(base[1], base[1], 0)]
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


def block_program(nf, suf):
    """the program of one block - whole (nf = 2) or half (nf = 1) - from L_block to the end of its epilogue; labels get `suf`"""
    global NF
    NF = nf
    nm = 8 * NF
    counts = {}
    head_code, G0 = block_start()
    # the LDS queue at a step boundary is the same for every step: find it as the fixed point of a main step
    pend = []
    for _ in range(3):
        _, pend = auto_waits(body(0, "main").flat("probe"), pend)
    start, pend_s = finish(G0.flat("S(0)"), [])
    assert pend_s == pend, (pend_s, pend)
    prog = head_code + start
    # first pass: the head step (accumulators from 0), then steps 1 .. 3 of the loop body; later passes: steps 0 .. 3
    Gh = body(0, "head")
    b, pe = finish(Gh.flat("step 0 (head)"), pend)
    assert pe == pend
    prog += b + [s_branch("L_body1")]
    prog += [label("L_loop")]
    for j in range(4):
        G = body(j, "main")
        b, pe = finish(G.flat(f"step j={j}"), pend)
        assert pe == pend, (j, pe, pend)
        if j == 1:
            prog += [label("L_body1")]
        prog += b
        counts[f"main{j}{suf}"] = G.costs()
    prog += [s_sub_u32(s_loop, s_loop, 1), s_cmp_lg_u32(s_loop, 0), s_cbranch_scc1("L_loop")]
    for j in range(4):
        G = body(j, f"tail{j}")
        post = []
        if j == 1:
            # the next block's K / V bases, its K rows on their way into the staging area TWO steps before they are picked up (at the top of the last step, when the K
            # registers are free): an LDS-DMA issued one step ahead was still in flight (timeline: the wait at the top of the last step cost 2-3 k cycles)
            if NF == 1:      # (a half block is the last thing a workgroup does: its "next" block is itself, a half)
                G.pre = seam_ptrs(s_nbid, s_hidxn, True) + G.pre
            else:
                G.pre = [s_cmp_eq_u32(s_moden, 1), s_cselect_b32(s_t[5], 1, 0)] + seam_ptrs(s_nbid, s_hidxn, s_t[5]) + G.pre
            G.spread(stage_dma(s_T.sub(0, 2), s_T[4], NF, rewind=True), 3 * nm, 4 * nm - 1)      # (behind the step's own refill of the ring)
        if j == 3:
            # K rows -> K registers (their last reader was S(nsteps - 1)); then the V rows into the same area
            G.pre = [s_waitcnt(vmcnt=5)] + G.pre          # (everything but the previous step's refill of the ring)
            G.spread(stage_reads(Kfr, NF), 0, 5 if NF == 2 else 2)
            G.put(6 if NF == 2 else 3, [s_waitcnt(lgkmcnt=0)])
            G.spread(stage_dma(s_T.sub(2, 2), s_T[5], NF, rewind=True), 6 if NF == 2 else 3, 2 * nm - 1)
            # (the V rows are picked up under the next block's S(0) group: block_start)
        b, pe = finish(G.flat(f"step tail{j}"), pend)
        if j < 3:
            assert pe == pend, (j, pe, pend)
        else:
            b += [s_waitcnt(lgkmcnt=0)] + post
        prog += b
        counts[f"tail{j}{suf}"] = G.costs()
    prog += epilogue()
    NF = 2

    def ren(ins):
        if ins.kind == "label":
            return label(ins.meta["name"] + suf)
        if ins.kind == "branch" and ins.meta["target"] not in ("L_end", "L_done", "L_block_F", "L_block_H"):
            return {None: s_branch, 0: s_cbranch_scc0, 1: s_cbranch_scc1}[ins.meta["cond"]](ins.meta["target"] + suf)
        return ins
    return [ren(i) for i in prog], counts


def build():
    """entry; whole blocks bid, bid + grid, ... < nfull; then, when the launcher says so, ONE half block; out"""
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
    f.write(f"#define UDM_DKV64_ASM{suffix} \\\n")
    for t in lines:
        f.write(f'  "{t}\\n\\t" \\\n')
    f.write('  ""\n')
    return prog, counts


def emit(path, ablations=()):
    global ABL
    with open(path, "w") as f:
        f.write("// GENERATED by asmgen/attn_dkv64.py - do not edit.  The whole persistent attention-backward dK / dV workgroup program as ONE asm statement.\n")
        f.write(f"#define UDM_DKV64_LDS_BYTES {LDS_TOTAL}\n")
        f.write(f"#define UDM_DKV64_PARAM_DWORDS {PARAM_DWORDS}\n")
        clob = [f'"v{i}"' for i in range(255)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(36, 100)] + ['"vcc"', '"scc"', '"m0"', '"memory"']
        f.write("#define UDM_DKV64_CLOBBERS " + ", ".join(clob) + "\n")
        ABL = 0
        prog, counts = emit_one(f, "")
        for a in ablations:
            ABL = a
            emit_one(f, f"_ABL{a}")
        ABL = 0
    return prog, counts


if __name__ == "__main__":
    prog, counts = emit(sys.argv[1] if len(sys.argv) > 1 else "attention_dkv64_gen.h", [int(x) for x in sys.argv[2:]])
    probs = lint([i for i in prog if i.kind != "raw"], mfma_states=4)
    print(stats(prog))
    for k, c in counts.items():
        print(k, c)
    for x in probs[:40]:
        print("LINT", x)
    print(len(probs), "lint problems", "| sgprs up to", S_.next - 1, "| vgprs up to", V.next - 1)
