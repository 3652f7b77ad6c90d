"""Generator of the 64-query-per-wave attention forward for gfx950 (head dim 128, no mask, L % 256 == 0).

Replaces `flash_attn_qkvpacked_func` (reference models/dit.py:843) on the headline path.  One workgroup = 4 waves = one wave per SIMD = 256
queries of one (batch, head); a wave owns TWO 32-query blocks and the whole 512-register file, so that every K / V fragment read from LDS
feeds two MFMAs (the 8-wave kernel of attention.hip reads one fragment per MFMA and is LDS-read bound at 0.35 of the matrix peak).

Registers of a wave (fixed, named here):
    a[0:127]    O^T accumulators, [block q][32-column group i][16]
    a[128:191]  Q fragments (MFMA B operand of the score MFMAs), [q][k-step][4]
    a[192:255]  K fragments of ONE key tile (MFMA A operand), [32-key block f][k-step][4]: read from LDS during the previous PV phase
    v[0:127]    S^T of two tiles (score MFMAs write VGPRs: the softmax reads them with the VALU), [buffer][q][f][16]
    v[128:191]  V^T fragments of one key tile, [16-key chunk][32-column group][4]: read (transposing) during the score phase
    v[192:223]  P^T as packed bf16 (MFMA B operand of the PV MFMAs), [q][chunk][4]
    v[224:254]  addresses, LDS-DMA offsets, softmax statistics

LDS (132 096 bytes): a ring of four K tiles and four V tiles.  A 64 x 128 tile is sixteen 1-KiB DMA pieces of four rows each, stored as
[64-byte column chunk c][row r][64 bytes] inside a piece.  A K fragment (`ds_read_b128`, lane = key row) then sits at
lane_base + c * 256 + (k-step & 1) * 32 with K pieces 1040 bytes apart (the 16 pad bytes rotate consecutive pieces over the banks), and a
V^T fragment half (`ds_read_b64_tr_b16`, 4 keys x 16 columns per 16 lanes) reads one contiguous 256-byte bank row per half-wave: both are
conflict-free AND every fragment address is ONE per-lane base register plus an immediate - no XOR swizzle arithmetic in the loop.

PERSISTENT: a workgroup walks blocks id, id + grid, ... (block -> (batch, head, 256-query tile) as in attention.hip: an XCD works through its
(batch, head) pairs one after the other).  With one wave per SIMD nothing hides a block's prologue, so the tile stream simply continues across
the seam: the last four tiles of a block refill the rings with the NEXT block's K(0..3) / V(0..2), its Q fragments are loaded into the (then
idle) Q registers under the block's last PV phase, and O leaves through a small dedicated LDS staging area (4 KiB per wave) so the rings stay
live.  Measured before this (one block per workgroup): 21 k of a block's 75 k cycles were prologue + epilogue.

Tile loop (one `s_barrier` per 64-key tile, counted `vmcnt`: refills issued two to three tiles ahead are never drained):
    phase A(t)   32 MFMAs S(t+1) = K(t+1) Q^T   under them: rest of softmax(t) (exp2, row sums, bf16 packing), the 32 V(t) fragment reads
    phase B(t)   32 MFMAs O^T += V(t)^T P(t)^T  under them: K(t+2) fragment reads, LDS-DMA of K(t+4) / V(t+3), running maximum of S(t+1), the
                 lazy-rescale decision (attention.hip's rule), first part of softmax(t+1)
Everything between two MFMAs is written out here, in order; `isa.lint` checks the software-visible hazards and `isa.run_workgroup`
executes the stream on the CPU (tests/test_asmgen.py).
"""
import sys
from isa import *   # noqa: F401,F403

D, BKV, KS = 128, 64, 8
NST = 4
K_PIECE, V_PIECE = 1040, 1024
K_TILE, V_TILE = 16 * K_PIECE, 16 * V_PIECE
LDS_K0, LDS_V0 = 0, NST * K_TILE
LDS_BYTES = LDS_V0 + NST * V_TILE
WAVE_OUT = 16384     # epilogue staging per wave (inside the K ring)

NQ = 2        # 32-query blocks per wave of the program being generated: 2 (256-query blocks) or 1 (the 128-query half blocks that balance the walk)
SUF = ""      # label suffix of that program


def nm():
    """MFMAs per phase"""
    return 16 * NQ


def G(x):
    """a gap index written for the 32-MFMA phases of the full program, scaled to this program's phase length"""
    return -1 if x < 0 else min(nm() - 1, x * nm() // 32)


V = Alloc("v", 0, 255)     # v255 is left to the compiler (the thread-id operand)
A = Alloc("a", 0, 256)
S_ = Alloc("s", 36, 100)

Sbuf = [V("Sa", 64), V("Sb", 64)]
Vf = V("Vf", 64)
negm = [V("negm0", 16), V("negm1", 16)]   # -mc[q] in all 16 registers: the C operand of a score chain's first MFMA, so the accumulator holds s - mc
kaddr, vaddr = V("kaddr"), V("vaddr")
dk, dv = V("dk", 4), V("dv", 4)
mc = [V("mc0"), V("mc1")]                 # reference exponent per query (base 2; q carries log2(e) / sqrt(D))
eight = V("eight")
ls = [[V("l00"), V("l01")], [V("l10"), V("l11")]]
mx = [V("mx0"), V("mx1")]
ta = [V("ta0"), V("ta1")]      # cross-half maximum (mloc) of the tile under decision
alpha = [V("al0"), V("al1")]
tmp = V("tmp", 6)
tb = [tmp[4], tmp[5]]
Oacc = A("O", 128)
Qf = A("Q", 64)
Ka = A("K", 64)

# inputs (copied from the asm statement's operands, in this order)
s_qb, s_kb, s_vb, s_ob, s_lseb = S_("qb", 2, 2), S_("kb", 2, 2), S_("vb", 2, 2), S_("ob", 2, 2), S_("lseb", 2, 2)
s_qstr, s_kstr, s_vstr, s_ostr = S_("qstr"), S_("kstr"), S_("vstr"), S_("ostr")
s_L, s_nkv, s_H, s_nt, s_mg_nt, s_mg_H, s_nfull, s_hashalf, s_lds, s_bid, s_gstride = (S_(n) for n in ("L", "nkv", "H", "nt", "mg_nt", "mg_H", "nfull", "hashalf", "lds", "bid", "gstride"))
INPUTS = ["qb", "kb", "vb", "ob", "lseb", "qstr", "kstr", "vstr", "ostr", "L", "nkv", "H", "nt", "mg_nt", "mg_H", "nfull", "hashalf", "lds", "bid", "gstride", "tid", "tl"]
# working
s_wave = S_("wave")
s_kt, s_vt = S_("kt", 2, 2), S_("vt", 2, 2)       # running tile bases of the refills
s_kstep, s_vstep = S_("kstep"), S_("vstep")       # bytes per 64-row tile
s_kdst, s_vdst = S_("kdst"), S_("vdst")           # LDS destination of this wave's first piece in stage 0
s_dec = [S_("dec0", 2, 2), S_("dec1", 2, 2)]
s_flag = [S_("flag0")]
s_loop, s_ret = S_("loop"), S_("ret")
s_t = [S_(f"t{i}") for i in range(6)]
s_t0, s_t1 = s_t[0], s_t[1]
s_qn, s_kn, s_vn = S_("qn", 2, 2), S_("kn", 2, 2), S_("vn", 2, 2)   # next block's operands (the current block's output pointers are recomputed in its epilogue)
s_relax, s_nbid = S_("relax"), S_("nbid")
# the walk: blocks bid, bid + grid, ... below `nfull` are whole 256-query blocks; when the remainder is exactly half a grid every workgroup finishes with ONE
# 128-query half (see next_block_ptrs) - 640 blocks on 256 CUs are 2.5 per workgroup instead of 3 for half of them and 2 for the rest
s_wg, s_hidx, s_hidxn, s_mode, s_moden, s_wqn = (S_(n) for n in ("wg", "hidx", "hidxn", "mode", "moden", "wqn"))
s_o, s_lse = s_dec[0], s_dec[1]      # the epilogue's output pointers live in the decision's SGPR pairs (idle from the block's last tile on)
s_tm = S_("tm", 2, 2)                # timeline builds: s_memtime lands here
v_tl = R("v", 254)             # timeline builds only (ABL & 16): cycle stamps of this wave in lanes 0..61, the output pointer in lanes 62, 63


def stamp(idx):
    """timeline builds: s_memtime at a point where lgkmcnt is drained anyway -> lane idx of v_tl"""
    if not (ABL & 16):
        return []
    return [s_memtime(s_tm), s_waitcnt(lgkmcnt=0), v_writelane_b32(v_tl, s_tm[0], idx)]

LDS_STG = LDS_BYTES            # O staging: 4 KiB per wave behind the rings
LDS_TOTAL = LDS_BYTES + 4 * 4096


def Sblk(buf, q, f):
    return Sbuf[buf].sub((q * 2 + f) * 16, 16)


def Vfr(cc, i):
    return Vf.sub((cc * 4 + i) * 4, 4)


def Pfr(buf, q, cc):
    """packed bf16 P^T of 16 keys: written IN PLACE over the first four of the eight fp32 scores it was made from"""
    return Sblk(buf, q, cc >> 1).sub(8 * (cc & 1), 4)


def Oblk(q, i):
    return Oacc.sub((q * 4 + i) * 16, 16)


def Qfr(q, ks):
    return Qf.sub((q * KS + ks) * 4, 4)


def Kfr(f, ks):
    return Ka.sub((f * KS + ks) * 4, 4)


# ---- pieces of the instruction stream -------------------------------------------------------------------------------------------------
def k_reads(stage):
    """the 16 K fragments of the tile in `stage` -> Ka, in the order the score MFMAs use them"""
    out = []
    for f in range(2):
        for ks in range(KS):
            out.append(ds_read_b128(Kfr(f, ks), kaddr, stage * K_TILE + f * 8 * K_PIECE + (ks >> 1) * 256 + (ks & 1) * 32))
    return out


def v_reads(stage):
    """the 16 V^T fragments (two transposing reads each) of the tile in `stage` -> Vf, in the order the PV MFMAs use them"""
    out = []
    for cc in range(4):
        for i in range(4):
            for h2 in range(2):
                out.append(ds_read_b64_tr_b16(Vfr(cc, i).sub(2 * h2, 2), vaddr, stage * V_TILE + (cc * 4 + 2 * h2) * V_PIECE + i * 256))
    return out


def dma_k(stage):
    out = []
    for j in range(4):
        out += [s_add_u32(M0, s_kdst, stage * K_TILE + j * K_PIECE), s_nop(0), global_load_lds_dwordx4(dk[j], s_kt)]
    out += [s_add_u32(s_kt[0], s_kt[0], s_kstep), s_addc_u32(s_kt[1], s_kt[1], 0)]
    return out


def dma_v(stage):
    out = []
    for j in range(4):
        out += [s_add_u32(M0, s_vdst, stage * V_TILE + j * V_PIECE), s_nop(0), global_load_lds_dwordx4(dv[j], s_vt)]
    out += [s_add_u32(s_vt[0], s_vt[0], s_vstep), s_addc_u32(s_vt[1], s_vt[1], 0)]
    return out


def score_mfmas(buf):
    """S[buf] = K Q^T from Ka / Qf: (f, ks, q) order - consecutive MFMAs are independent, each K fragment feeds two"""
    out = []
    for f in range(2):
        for ks in range(KS):
            for q in range(NQ):
                d = Sblk(buf, q, f)
                out.append(v_mfma_f32_32x32x16_bf16(d, Kfr(f, ks), Qfr(q, ks), negm[q] if ks == 0 else d))
    return out


def pv_mfmas(buf):
    out = []
    for cc in range(4):
        for i in range(4):
            for q in range(NQ):
                out.append(v_mfma_f32_32x32x16_bf16(Oblk(q, i), Vfr(cc, i), Pfr(buf, q, cc), Oblk(q, i)))
    return out


def max_ops(buf):
    """running maximum of the 32 scores a lane holds per query block: 16 v_max3 per block; the f = 0 halves (finished first) first, the two
    blocks alternating (independent dependency chains)"""
    out = []
    for f in range(2):
        chains = []
        for q in range(NQ):
            s = Sblk(buf, q, f)
            ops = []
            if f == 0:
                ops.append(v_max3_f32(mx[q], s[0], s[1], s[2]))
                rest = list(range(3, 16))
            else:
                rest = list(range(16))
            while rest:
                x = rest.pop(0)
                y = rest.pop(0) if rest else x
                ops.append(v_max3_f32(mx[q], mx[q], s[x], s[y]))
            chains.append(ops)
        for ops in zip(*chains):
            out += list(ops)
    return out


def decide_ops(tag):
    """cross-half maximum of the accumulators (= s - mc); the reference exponent moves iff some lane of the wave saw more than 2^8 above it (attention.hip's rule)"""
    out = []
    for q in range(NQ):
        out += [v_mov_b32(ta[q], mx[q]), v_mov_b32(tb[q], mx[q])]
    out += [s_nop(1 if NQ == 1 else 0)] + [v_permlane32_swap_b32(ta[q], tb[q]) for q in range(NQ)]
    out += [v_max_f32(ta[q], ta[q], tb[q]) for q in range(NQ)]
    out += [v_cmp_gt_f32(s_dec[q], ta[q], eight) for q in range(NQ)]
    out += [s_or_b64(s_dec[0], s_dec[0], s_dec[NQ - 1])]          # SCC = some lane moves
    out += [s_cbranch_scc1(f"L_move_{tag}"), label(f"L_moved_{tag}")]
    return out


def move_block(tag, buf, set_flag=True):
    """out of line: every lane of both blocks raises its reference exponent by d = max(s_max - mc, 0): the row sums are rescaled now, the scores of the tile
    under decision (still accumulators: nothing of it is exponentiated yet) drop by d, the C-operand registers follow, O^T is rescaled at the end of the PV
    phase (P(t), still being accumulated, is relative to the OLD exponent).  set_flag = False: the block's first tile (O^T = 0, l = 0)."""
    out = [label(f"L_move_{tag}"), s_nop(1)]
    d = [tmp[0], tmp[1]]
    for q in range(NQ):
        out += [v_max_f32(d[q], 0, ta[q])]
    for q in range(NQ):
        out += [v_sub_f32(tmp[2 + q], 0, d[q])]
    for q in range(NQ):
        out += [v_exp_f32(alpha[q], tmp[2 + q])]
    for q in range(NQ):
        out += [v_add_f32(mc[q], mc[q], d[q])]
    for q in range(NQ):
        out += [v_mul_f32(ls[q][0], ls[q][0], alpha[q]), v_mul_f32(ls[q][1], ls[q][1], alpha[q]), v_sub_f32(tmp[2 + q], 0, mc[q])]
    for q in range(NQ):
        for r in range(16):
            out += [v_mov_b32(negm[q][r], tmp[2 + q])]
        for f in range(2):
            blk = Sblk(buf, q, f)
            for r in range(16):
                out += [v_sub_f32(blk[r], blk[r], d[q])]
    if set_flag:
        out += [s_mov_b32(s_flag[0], 1)]
    out += [s_branch(f"L_moved_{tag}")]
    return out


def rescale_o_block():
    """out of line, shared: O^T *= alpha (accumulators live in AGPRs: read, multiply, write back); returns through s_ret"""
    out = [label("L_rescale"), s_nop(15), s_nop(15)]
    for q in range(NQ):
        for base in range(0, 64, 4):
            regs = [Oacc.sub(q * 64 + base + e, 1) for e in range(4)]
            out += [v_accvgpr_read_b32(tmp[e], regs[e]) for e in range(4)]
            out += [v_mul_f32(tmp[e], tmp[e], alpha[q]) for e in range(4)]
            out += [v_accvgpr_write_b32(regs[e], tmp[e]) for e in range(4)]
    out += [s_mov_b32(s_flag[0], 0), s_nop(3)]
    for k in range(7):
        out += [s_cmp_eq_u32(s_ret, k), s_cbranch_scc1(f"L_rescaled_{k}")]
    out += [s_endpgm()]
    return out


def softmax_group(buf, q, cc):
    """8 accumulators (s - mc) of block q, key chunk cc: p = exp2(.), row sums (two chains), bf16 P packed over the first four of them"""
    f, r0 = cc >> 1, 8 * (cc & 1)
    s = Sblk(buf, q, f)
    out = [v_exp_f32(s[r0 + e], s[r0 + e]) for e in range(8)]
    out += [v_add_f32(ls[q][e & 1], ls[q][e & 1], s[r0 + e]) for e in range(8)]
    out += [v_cvt_pk_bf16_f32(s[r0 + j], s[r0 + 2 * j], s[r0 + 2 * j + 1]) for j in range(4)]
    return out


def groups():
    """softmax groups of a tile in the order the PV MFMAs need their P"""
    return [(q, cc) for cc in range(4) for q in range(NQ)]


def n_early():
    """groups of softmax(t+1) done in phase B(t); the rest in phase A(t+1)"""
    return 3 if NQ == 2 else 2


CAP = 5.5          # issue budget behind one MFMA (single-issue slots of ~4 cycles under its 32; MI355X_MICROARCH.md: <= 5 hidden per gap)
COST = {"trans": 1.5, "dma": 1.5, "label": 0.0, "comment": 0.0}


def spread(mfmas, streams, note, cap=None):
    return schedule_gaps(mfmas, streams, note, CAP if cap is None else cap, COST)


ABL = 0   # timing-only ablations (WRONG results): 1 = no softmax VALU in the loop, 2 = no fragment reads, 4 = no refills / waits / barriers, 8 = no MFMAs;
          # 16 = timeline stamps (correct results)


def _abl(prog):
    out = []
    for ins in prog:
        k = ins.kind
        if (ABL & 1) and k in ("valu", "trans", "permlane", "valu_sgpr") and not ins.meta.get("keep"):
            continue
        if (ABL & 1) and k == "branch" and "L_move_" in ins.text:
            continue
        if (ABL & 2) and k == "lds_rd":
            continue
        if (ABL & 4) and (k in ("dma", "barrier", "vmem_ld") or (k == "wait" and ins.meta.get("vmcnt") is not None) or
                          (k == "salu" and any(w in ins.writes for w in [("m0", 0)] + s_kt.regs() + s_vt.regs()))):
            continue
        if (ABL & 4) and k == "nop":
            continue
        if (ABL & 8) and k == "mfma":
            continue
        out.append(ins)
    return out


def body(j, variant, tag, vm_wait):
    prog, ca, cb = _body(j, variant, tag, vm_wait)
    return (_abl(prog) if ABL else prog), ca, cb


def q_loads(src, scratch):
    """the 16 Q fragments of the block at `src` (its first query row, head offset applied) -> Qf.  lane (query l31 of block q, half hi) reads 16 bytes
    per k-step at (wave * W + q * 32 + l31) * qstr + hi * 16 + ks * 32, W = s_wqn = 64 rows per wave (32 for a half block: its q = 1 fragments
    re-read the q = 0 rows and are never used); scratch: 4 free VGPRs"""
    t = scratch
    p = v_mbcnt_lane_id(t[0])
    p += [v_and_b32(t[1], 31, t[0]), v_lshrrev_b32(t[2], 5, t[0]), s_mul_i32(s_t[4], s_wave, s_wqn), s_sub_u32(s_t[5], s_wqn, 32), s_mul_i32(s_t[5], s_t[5], s_qstr)]
    p += [v_add_u32(t[1], s_t[4], t[1]), v_lshlrev_b32(t[2], 4, t[2]), v_mul_lo_u32(t[1], t[1], s_qstr)]
    p += [v_add_u32(t[0], t[1], t[2]), s_nop(0), v_add_u32(t[3], s_t[5], t[0])]
    loads = [global_load_dwordx4(Qfr(q, ks), t[0] if q == 0 else t[3], src, ks * 32) for q in range(2) for ks in range(KS)]
    for ins in p:
        ins.meta["keep"] = True      # (address arithmetic: the timing-only ablation that drops the softmax's VALU work must not drop these)
    return p, loads


def cond_wait(tag, relaxed, strict=12):
    """block-start tiles: the first trip of a block that follows another one sees the seam's Q loads and O stores in the vector-memory queue"""
    return [s_cmp_eq_u32(s_relax, 1), s_cbranch_scc1(f"L_wr_{tag}"), s_waitcnt(vmcnt=strict), s_branch(f"L_wj_{tag}"), label(f"L_wr_{tag}"), s_waitcnt(vmcnt=relaxed),
            label(f"L_wj_{tag}")]


def _body(j, variant, tag, vm_wait):
    """tile t with t % 4 == j.  variant 'main', or the last four tiles of a block, whose refills fetch the NEXT block's first tiles:
    'tail0' (t = nkv-4: K'(0), own V(nkv-1)), 'tail1' (K'(1), V'(0)), 'tail2' (K'(2), V'(1); no K reads - the scores of tile nkv-1 were the
    block's last; the next block's Q fragments are loaded under its PV phase), 'tail3' (K'(3), V'(2); no score MFMAs, no softmax of a next tile)"""
    cur, nxt = j & 1, (j + 1) & 1
    last = variant == "tail3"
    prog = [comment(f"---- tile body j={j} {variant}")] + stamp(8 + 3 * tag)
    if variant == "tail0":
        prog += [s_mov_b64(s_kt, s_kn)]
    if variant == "tail1":
        prog += [s_mov_b64(s_vt, s_vn)]
    # ---------------- phase A
    if variant == "main" and j == 0:
        wait_bar = cond_wait(tag, 18) + [s_barrier()]
    elif variant == "main" and j == 1:
        wait_bar = cond_wait(tag, 26) + [s_mov_b32(s_relax, 0), s_barrier()]
    else:
        wait_bar = [s_waitcnt(vmcnt=vm_wait), s_barrier()]
    fin = []
    for (q, cc) in groups()[n_early():]:
        fin += softmax_group(cur, q, cc)
    vr = v_reads(j)
    if not last:
        a, ca = spread(score_mfmas(nxt), [(wait_bar, G(1), G(1)), (vr, G(1) + 1, G(28)), (fin, -1, G(31))], "phase A")
    else:
        a, ca = [comment("phase A (last tile: no scores)")] + fin[:8 * NQ] + wait_bar + vr + fin[8 * NQ:], []
    prog += a
    prog += [s_waitcnt(lgkmcnt=0)] + stamp(9 + 3 * tag)
    # ---------------- phase B
    streams = []
    if variant in ("main", "tail0", "tail1"):
        streams.append((k_reads((j + 2) % 4), -1, G(20)))
    dma = dma_k(j) + dma_v((j + 3) % 4)
    streams.append((dma, G(1), G(14) if variant == "tail2" else G(31)))
    # the next block's Q fragments: a load instruction touches 32 rows (64 cache lines) - sixteen of them back to back stalled the issue for 3.2 k cycles
    # (timeline), so they go out one per four MFMAs over the last two PV phases.  Offsets live in two registers of the S buffer whose tile is finished.
    if variant == "tail2":     # (registers 60..63 of an S buffer never hold P: stale exponentials of a finished tile)
        qa, ql = q_loads(s_qn, [Sbuf[cur][60 + i] for i in range(4)])
        streams.append((qa + ql[:8], G(15), G(31)))
    if variant == "tail3":
        _, ql = q_loads(s_qn, [Sbuf[nxt][60 + i] for i in range(4)])      # (tail2's S buffer = this tile's `nxt`: never written in the last tile)
        streams.append((ql[8:], G(4), G(31)))
        streams.append((block_ptrs(s_bid, s_hidx if NQ == 1 else 0, o=s_o, lse=s_lse), G(1), G(31)))     # (no decision in the last tile: its SGPR pairs are free)
    if not last:
        soft = max_ops(nxt) + decide_ops(tag)
        early = []
        for (q, cc) in groups()[:n_early()]:
            early += softmax_group(nxt, q, cc)
        streams.append((soft, G(1), G(14)))       # (the f = 0 halves of S(t+1) were finished half a phase ago; the f = 1 maxima come 16 ops later)
        streams.append((early, G(14) + 1, G(31)))
    # (a half block has the same fragment reads, refills and half the softmax behind half the MFMAs: its PV phase is issue bound, spread evenly)
    b, cb = spread(pv_mfmas(cur), streams, "phase B", cap=None if NQ == 2 else 8.5)
    prog += b
    if not last:
        prog += [s_waitcnt(lgkmcnt=0)] + stamp(10 + 3 * tag)
        prog += [s_cmp_lg_u32(s_flag[0], 0), s_cbranch_scc0(f"L_rescaled_{tag}"), s_mov_b32(s_ret, tag), s_branch("L_rescale"), label(f"L_rescaled_{tag}")]
    return prog, ca, cb


def block_ptrs(bid, hidx, q=None, k=None, v=None, o=None, lse=None):
    """block id (+ 128-row half index: an SGPR, or 0) -> (batch, head, first query row) -> the pointers asked for (SALU only).  Needs (B H) % 8 == 0."""
    t0, t1, t2, t3 = s_t[0], s_t[1], s_t[2], s_t[3]
    p = [s_lshr_b32(t0, bid, 3), s_and_b32(t1, bid, 7), s_mul_hi_u32(t2, t0, s_mg_nt), s_mul_i32(t3, t2, s_nt), s_sub_u32(t0, t0, t3),   # t0 = tile
         s_lshl_b32(t2, t2, 3), s_add_u32(t2, t2, t1),                                                                                    # t2 = b H + h
         s_mul_hi_u32(t1, t2, s_mg_H), s_mul_i32(t3, t1, s_H), s_sub_u32(t3, t2, t3), s_lshl_b32(t3, t3, 8),                              # t1 = b, t3 = h * 256 bytes
         s_mul_i32(t1, t1, s_L), s_lshl_b32(t0, t0, 8)]                                                                                   # t1 = b L, t0 = tile * 256
    if not (isinstance(hidx, int) and hidx == 0):
        p += [s_lshl_b32(s_t[5], hidx, 7), s_add_u32(t0, t0, s_t[5])]                                                                     # + half * 128 rows
    for dst, base, stride in ((k, s_kb, s_kstr), (v, s_vb, s_vstr)):
        if dst is not None:
            p += [s_mul_i32(dst[0], t1, stride), s_mul_hi_u32(dst[1], t1, stride), s_add_u32(dst[0], dst[0], t3), s_addc_u32(dst[1], dst[1], 0),
                  s_add_u32(dst[0], dst[0], base[0]), s_addc_u32(dst[1], dst[1], base[1])]
    if lse is not None:
        p += [s_mul_i32(t2, t2, s_L), s_add_u32(t2, t2, t0), s_lshl_b32(t2, t2, 2), s_add_u32(lse[0], s_lseb[0], t2), s_addc_u32(lse[1], s_lseb[1], 0)]
    p += [s_add_u32(t1, t1, t0)]
    for dst, base, stride in ((q, s_qb, s_qstr), (o, s_ob, s_ostr)):
        if dst is not None:
            p += [s_mul_i32(dst[0], t1, stride), s_mul_hi_u32(dst[1], t1, stride), s_add_u32(dst[0], dst[0], t3), s_addc_u32(dst[1], dst[1], 0),
                  s_add_u32(dst[0], dst[0], base[0]), s_addc_u32(dst[1], dst[1], base[1])]
    return p


def next_block_ptrs():
    """What this workgroup does after the current block (s_moden: 0 = a whole block, 1 = its 128-query half, 2 = nothing) and that block's operand pointers - the
    current block's own when there is none (its prefetches then re-read valid memory and are never used)."""
    p = []
    if NQ == 1:     # a half block is always the last thing a workgroup does
        p += [s_mov_b32(s_moden, 2), s_mov_b32(s_nbid, s_bid), s_mov_b32(s_hidxn, s_hidx), s_mov_b32(s_wqn, 32)]
    else:
        p += [s_mov_b32(s_hidxn, 0), s_mov_b32(s_wqn, 64), s_mov_b32(s_moden, 0),
              s_add_u32(s_nbid, s_bid, s_gstride), s_cmp_lt_u32(s_nbid, s_nfull), s_cbranch_scc1("L_np"),
              s_mov_b32(s_moden, 2), s_mov_b32(s_nbid, s_bid), s_cmp_eq_u32(s_hashalf, 0), s_cbranch_scc1("L_np"),
              # workgroup 8 a + x (x = its XCD) takes half a & 1 of block nfull + 8 (a >> 1) + x: the block id stays congruent to the XCD, whose L2 then holds its K / V
              s_mov_b32(s_moden, 1), s_mov_b32(s_wqn, 32), s_lshr_b32(s_nbid, s_wg, 4), s_lshl_b32(s_nbid, s_nbid, 3), s_and_b32(s_t[0], s_wg, 7), s_add_u32(s_nbid, s_nbid, s_t[0]),
              s_add_u32(s_nbid, s_nbid, s_nfull), s_lshr_b32(s_hidxn, s_wg, 3), s_and_b32(s_hidxn, s_hidxn, 1),
              label("L_np")]
    return p + block_ptrs(s_nbid, s_hidxn, q=s_qn, k=s_kn, v=s_vn)


def entry():
    p = [comment("---- entry: constants of the wave, first block's loads")]
    raw = lambda t: Inst(t, "raw")
    regs = [s_qb, s_kb, s_vb, s_ob, s_lseb, s_qstr, s_kstr, s_vstr, s_ostr, s_L, s_nkv, s_H, s_nt, s_mg_nt, s_mg_H, s_nfull, s_hashalf, s_lds, s_bid, s_gstride]
    assert [S_.names[n] for n in INPUTS[:-2]] == regs
    for i, r in enumerate(regs):
        p += [raw(f"s_mov_b{64 if r.n == 2 else 32} {r}, %{i}")]
    tid = tmp[0]
    p += [raw(f"v_mov_b32 {tid}, %{len(regs)}")]
    if ABL & 64:
        p += [s_branch("L_end")]
    if ABL & 16:
        p += [raw(f"s_mov_b64 {s_dec[0]}, %{len(regs) + 1}"), v_mov_b32(v_tl, 0), s_nop(1), v_writelane_b32(v_tl, s_dec[0][0], 62), v_writelane_b32(v_tl, s_dec[0][1], 63)] + stamp(0)
    t = tmp
    p += [s_nop(0), v_lshrrev_b32(t[1], 6, tid), s_nop(0), v_readfirstlane_b32(s_wave, t[1]), v_and_b32(t[0], 63, tid)]
    lane_v, l31, hi = t[0], t[1], t[2]
    p += [v_and_b32(l31, 31, lane_v), v_lshrrev_b32(hi, 5, lane_v)]
    # ---- LDS-DMA source offsets: lane i -> column chunk c = i >> 4, row r = (i >> 2) & 3, 16-byte slot s = i & 3 of the piece's four rows
    #      byte offset = (16 wave + 4 j + r) * stride + (4 c + s) * 16
    p += [v_lshrrev_b32(t[3], 2, lane_v), v_and_b32(t[3], 3, t[3])]                 # r
    p += [s_lshl_b32(s_t0, s_wave, 4), s_nop(0), v_add_u32(t[3], s_t0, t[3])]         # 16 wave + r
    p += [v_lshrrev_b32(t[4], 4, lane_v), v_lshlrev_b32(t[4], 2, t[4]), v_and_b32(t[5], 3, lane_v), v_add_u32(t[4], t[4], t[5]), v_lshlrev_b32(t[4], 4, t[4])]
    for j in range(4):
        p += [v_add_u32(t[5], 4 * j, t[3]), v_mul_lo_u32(dk[j], t[5], s_kstr), v_mul_lo_u32(dv[j], t[5], s_vstr)]
        p += [v_add_u32(dk[j], dk[j], t[4]), v_add_u32(dv[j], dv[j], t[4])]
    p += [s_lshl_b32(s_kstep, s_kstr, 6), s_lshl_b32(s_vstep, s_vstr, 6)]
    p += [s_mul_i32(s_t0, s_wave, 4 * K_PIECE), s_add_u32(s_kdst, s_lds, s_t0), s_add_u32(s_kdst, s_kdst, LDS_K0)]
    p += [s_mul_i32(s_t0, s_wave, 4 * V_PIECE), s_add_u32(s_vdst, s_lds, s_t0), s_add_u32(s_vdst, s_vdst, LDS_V0)]
    # ---- fragment read bases
    #   K: piece = l31 >> 2 (+ 8 f), row = l31 & 3:  lds + K0 + piece * 1040 + row * 64 + hi * 16
    p += [s_mov_b32(s_t0, K_PIECE), v_lshrrev_b32(t[3], 2, l31), v_mul_lo_u32(t[3], t[3], s_t0), v_and_b32(t[4], 3, l31), v_lshlrev_b32(t[4], 6, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_lshlrev_b32(t[4], 4, hi), v_add_u32(t[3], t[3], t[4]), v_add_u32(kaddr, s_lds, t[3])]
    if LDS_K0:
        p += [v_add_u32(kaddr, LDS_K0, kaddr)]
    #   V: piece = hi (+ 4 cc + 2 h2), key row = (lane & 15) >> 2, 16-column half g1 = (lane >> 4) & 1, 4 columns (lane & 3):
    #      lds + V0 + hi * 1024 + row * 64 + g1 * 32 + (lane & 3) * 8
    p += [v_lshlrev_b32(t[3], 10, hi), v_and_b32(t[4], 15, lane_v), v_lshrrev_b32(t[4], 2, t[4]), v_lshlrev_b32(t[4], 6, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_lshrrev_b32(t[4], 4, lane_v), v_and_b32(t[4], 1, t[4]), v_lshlrev_b32(t[4], 5, t[4]), v_add_u32(t[3], t[3], t[4]),
          v_and_b32(t[4], 3, lane_v), v_lshlrev_b32(t[4], 3, t[4]), v_add_u32(t[3], t[3], t[4]), v_add_u32(t[3], s_lds, t[3]), v_add_u32(vaddr, LDS_V0, t[3])]
    p += [s_mov_b32(s_flag[0], 0), s_mov_b32(s_relax, 0), v_mov_b32(eight, 8.0), s_mov_b32(s_wg, s_bid), s_mov_b32(s_hidx, 0), s_mov_b32(s_wqn, 64)]
    # ---- first block (always a whole one: the launcher balances with halves only behind at least one whole block per workgroup): its pointers, Q, and the ring
    #      as if its tiles -4 .. -1 had run: K0, K1, V0, K2, V1, K3, V2
    p += block_ptrs(s_bid, 0, q=s_qn, k=s_kt, v=s_vt)
    if ABL & 32:   # debug: dump the first block's scalars into the LSE tensor (lane i of wave 0 stores SGPR 36 + i) and stop
        dump = v_mbcnt_lane_id(t[0])
        dump += [v_lshlrev_b32(t[1], 2, t[0]), s_lshl_b32(s_t0, s_bid, 8), s_nop(0), v_add_u32(t[1], s_t0, t[1]), v_mov_b32(t[2], 0), s_nop(1)]
        for i in range(64):
            dump += [v_writelane_b32(t[2], R("s", 36 + i), i)]
        dump += [global_store_dword(t[1], t[2], s_lseb, 0), s_waitcnt(vmcnt=0), s_branch("L_end")]
        p += dump
    qa, ql = q_loads(s_qn, [Sbuf[0][i] for i in range(4)])
    p += qa + ql
    p += dma_k(0) + dma_k(1) + dma_v(0) + dma_k(2) + dma_v(1) + dma_k(3) + dma_v(2)
    p += stamp(1)
    p += [s_waitcnt(vmcnt=0)]      # once per workgroup; later blocks find their first tiles prefetched
    p += stamp(2)
    return p


def block_start():
    """per block: the next block's pointers, S(0) with the zeroing of O^T under its MFMAs, first reference exponent, K(1) fragments"""
    p = [comment("---- block start"), label("L_block")] + stamp(3)
    p += next_block_ptrs()
    # the seam's loads have landed - the last Q loads are the youngest of them - only the previous block's 18 stores may still fly; fresh block: nothing does
    p += [s_cmp_eq_u32(s_relax, 1), s_cbranch_scc0("L_bs_fresh"), s_waitcnt(vmcnt=18), label("L_bs_fresh")] + stamp(4) + [s_barrier()]
    p += k_reads(0) + [s_waitcnt(lgkmcnt=0)] + stamp(5)
    for q in range(NQ):
        p += [v_mov_b32(mc[q], 0)] + [v_mov_b32(negm[q][r], 0) for r in range(16)]
    p += [s_nop(1)]
    zero = [v_accvgpr_write_b32(Oacc[r], 0) for r in range(64 * NQ)]
    for q in range(NQ):
        zero += [v_mov_b32(ls[q][0], 0), v_mov_b32(ls[q][1], 0)]
    sm, _ = spread(score_mfmas(0), [(zero, 0, nm() - 1)], "S(0)")
    p += sm
    p += k_reads(1)
    p += [s_nop(7)]
    # first reference exponent: the common decision with mc = 0 (scores within 2^8 of zero keep it there); nothing to rescale yet
    p += max_ops(0) + decide_ops(7)
    for (q, cc) in groups()[:n_early()]:
        p += softmax_group(0, q, cc)
    p += [s_waitcnt(lgkmcnt=0)] + stamp(6)
    # trips of the four steady-state bodies: nkv / 4 - 1 (>= 1: the launcher takes L >= 512)
    p += [s_lshr_b32(s_loop, s_nkv, 2), s_sub_u32(s_loop, s_loop, 1)]
    return p


def epilogue():
    """O^T / l -> bf16, 64 columns at a time through this wave's 4 KiB staging area (rows of 128 bytes, 16-byte slots XOR-ed with the row), stored as
    whole 128-byte row segments; LSE.  The rings are NOT touched: they already hold the next block's first tiles."""
    e = [comment("---- epilogue")] + stamp(40)
    t = tmp
    e += [s_nop(15)]    # the last PV MFMAs have written O^T
    inv = [alpha[0], alpha[1]]
    for q in range(NQ):
        e += [v_add_f32(ls[q][0], ls[q][0], ls[q][1])]
    for q in range(NQ):
        e += [v_mov_b32(ta[q], ls[q][0]), v_mov_b32(tb[q], ls[q][0])]
    e += [s_nop(1)] + [v_permlane32_swap_b32(ta[q], tb[q]) for q in range(NQ)]
    e += [v_add_f32(ta[q], ta[q], tb[q]) for q in range(NQ)] + [s_nop(0)]
    e += [v_rcp_f32(inv[q], ta[q]) for q in range(NQ)] + [v_log_f32(tb[q], ta[q]) for q in range(NQ)] + [s_nop(0)]
    e += [v_add_f32(tb[q], mc[q], tb[q]) for q in range(NQ)]
    lane_v, l31, hi = mx[0], mx[1], ta[0]
    e += v_mbcnt_lane_id(lane_v)
    e += [v_and_b32(l31, 31, lane_v), v_lshrrev_b32(hi, 5, lane_v)]
    e += [s_lshl_b32(s_t0, s_wave, 4 + NQ), s_nop(0), v_add_u32(t[0], s_t0, l31), v_lshlrev_b32(t[0], 2, t[0])]      # (a wave's rows: 32 NQ)
    e += [global_store_dword(t[0], tb[q], s_lse, 128 * q) for q in range(NQ)]
    # staging write address: row r = l31: X = stg + r * 128 + ((r & 7) << 4) + hi * 8; slot sl = (i & 1) * 4 + rg at X ^ (sl << 4)
    xb = t[1]
    e += [s_lshl_b32(s_t0, s_wave, 12), s_add_u32(s_t0, s_t0, s_lds), s_add_u32(s_t0, s_t0, LDS_STG)]
    e += [v_lshlrev_b32(xb, 7, l31), v_and_b32(t[2], 7, l31), v_lshlrev_b32(t[2], 4, t[2]), v_add_u32(xb, xb, t[2]), v_lshlrev_b32(t[2], 3, hi), v_add_u32(xb, xb, t[2]),
          v_add_u32(xb, s_t0, xb)]
    # read-back address: instruction k reads rows 8 k + (lane >> 3), slot lane & 7:  stg + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4) + k * 1024
    g8, rdaddr = ta[1], t[4]
    e += [v_lshrrev_b32(g8, 3, lane_v), v_and_b32(t[2], 7, lane_v), v_xor_b32(t[3], t[2], g8), v_lshlrev_b32(t[3], 4, t[3]), v_lshlrev_b32(rdaddr, 7, g8), v_add_u32(rdaddr, rdaddr, t[3]),
          v_add_u32(rdaddr, s_t0, rdaddr)]
    # global offsets: (wave * 32 NQ + 8 k + (lane >> 3)) * ostr + (lane & 7) * 16   (+ q * 32 rows, + ch * 128 bytes as immediate)
    go = [negm[0][0], negm[0][1], negm[0][2], negm[0][3]]      # (the next block start rewrites the C-operand registers)
    e += [s_lshl_b32(s_t1, s_wave, 4 + NQ), s_nop(0), v_add_u32(t[3], s_t1, g8), v_mul_lo_u32(t[3], t[3], s_ostr), v_lshlrev_b32(t[2], 4, t[2]), v_add_u32(go[0], t[3], t[2])]
    e += [s_lshl_b32(s_t1, s_ostr, 3), s_nop(0)]
    for k in range(1, 4):
        e += [v_add_u32(go[k], s_t1, go[k - 1])]
    e += [s_lshl_b32(s_t1, s_ostr, 5)]
    pk = Vf
    n = 0
    for q in range(NQ):
        for ch in range(2):
            for i in (2 * ch, 2 * ch + 1):
                for rg in range(4):
                    o = Oblk(q, i)
                    vals = [Sbuf[0][(n * 4 + k) % 64] for k in range(4)]
                    e += [v_accvgpr_read_b32(vals[k], o[rg * 4 + k]) for k in range(4)]
                    e += [v_mul_f32(vals[k], vals[k], inv[q]) for k in range(4)]
                    dst = pk.sub((n % 16) * 2, 2)
                    e += [v_cvt_pk_bf16_f32(dst[0], vals[0], vals[1]), v_cvt_pk_bf16_f32(dst[1], vals[2], vals[3])]
                    e += [v_xor_b32(t[2 + (n & 1)], ((i & 1) * 4 + rg) << 4, xb)]
                    e += [ds_write_b64(t[2 + (n & 1)], dst, 0)]
                    n += 1
            rb = Sbuf[1]
            for k in range(4):
                e += [ds_read_b128(rb.sub(k * 4, 4), rdaddr, k * 1024)]
            e += [s_waitcnt(lgkmcnt=0)]
            for k in range(4):
                e += [global_store_dwordx4(go[k], rb.sub(k * 4, 4), s_o, ch * 128)]
            e += [s_nop(1)]
        if q == 0 and NQ == 2:
            for k in range(4):
                e += [v_add_u32(go[k], s_t1, go[k])]
    e += stamp(41)
    return e


def timeline_store():
    if not (ABL & 16):
        return []
    t = tmp
    p = v_mbcnt_lane_id(t[0])
    # [workgroup][wave][64]
    p += [s_nop(0)]
    return p + [Inst("s_nop 0", "nop", count=1)] + _tl_tail(t)


def _tl_tail(t):
    return [s_lshl_b32(s_t[4], s_wg, 2), s_add_u32(s_t[4], s_t[4], s_wave), s_lshl_b32(s_t[4], s_t[4], 8), v_lshlrev_b32(t[0], 2, t[0]), v_add_u32(t[0], s_t[4], t[0]),
            v_readlane_b32(s_dec[0][0], v_tl, 62), v_readlane_b32(s_dec[0][1], v_tl, 63), s_nop(4), global_store_dword(t[0], v_tl, s_dec[0], 0), s_waitcnt(vmcnt=0)]


def block_program(nq, suf):
    """the program of one block - whole (nq = 2) or half (nq = 1) - from L_block to the end of its epilogue, and its out-of-line pieces; labels get `suf`"""
    global NQ
    NQ = nq
    counts = {}
    prog = block_start()
    prog += [label("L_loop")]
    for j in range(4):
        b, ca, cb = body(j, "main", j, 12)
        prog += b
        counts[f"main{j}{suf}"] = (ca, cb)
    prog += [s_sub_u32(s_loop, s_loop, 1), s_cmp_lg_u32(s_loop, 0), s_cbranch_scc1("L_loop")]
    # the block's last four tiles; in flight behind what each needs: 12, 12, 12 (stricter than the 16 possible) and, for the last one, the two refill
    # groups + the first 8 Q loads issued since V(nkv-1) = 24
    for j, (variant, vmw) in enumerate([("tail0", 12), ("tail1", 12), ("tail2", 12), ("tail3", 24)]):
        b, ca, cb = body(j, variant, 4 + j, vmw)
        prog += b
        counts[variant + suf] = (ca, cb)
    prog += epilogue()
    side = []
    for tag in range(7):       # bodies 0-3 decide about tile t+1 in S buffer (j+1) & 1; the three tails 4-6 likewise
        side += move_block(tag, (tag + 1) & 1)
    side += move_block(7, 0, set_flag=False)
    side += rescale_o_block()
    NQ = 2

    def ren(ins):
        if ins.kind == "label":
            return label(ins.meta["name"] + suf)
        if ins.kind == "branch" and ins.meta["target"] not in ("L_end", "L_done", "L_block_F", "L_block_H"):
            return {None: s_branch, 0: s_cbranch_scc0, 1: s_cbranch_scc1}[ins.meta["cond"]](ins.meta["target"] + suf)
        return ins
    return [ren(i) for i in prog], [ren(i) for i in side], counts


def build():
    """entry; whole blocks bid, bid + grid, ... < nfull; then, when the launcher says so, ONE half block; out"""
    prog = entry()
    full, side_f, counts = block_program(2, "_F")
    half, side_h, counts_h = block_program(1, "_H")
    counts.update(counts_h)
    after = [s_cmp_eq_u32(s_moden, 2), s_cbranch_scc1("L_done"), s_mov_b32(s_bid, s_nbid), s_mov_b32(s_hidx, s_hidxn), s_mov_b32(s_relax, 1),
             s_cmp_eq_u32(s_moden, 1), s_cbranch_scc1("L_block_H"), s_branch("L_block_F")]
    prog += full + after + half
    prog += [label("L_done"), s_waitcnt(vmcnt=0)] + stamp(42) + timeline_store() + [s_branch("L_end")]
    prog += side_f + side_h
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
                continue                 # the enclosing kernel function ends the program
            t = "s_branch L_end_%="     # the rescale block's unreachable fall-through
        lines.append(t)
    f.write(f"#define UDM_FWD64_ASM{suffix} \\\n")
    for t in lines:
        f.write(f'  "{t}\\n\\t" \\\n')
    f.write('  ""\n')
    return prog, counts


def emit(path, ablations=()):
    global ABL
    with open(path, "w") as f:
        f.write("// GENERATED by asmgen/attn_fwd64.py - do not edit.  The whole persistent attention-forward workgroup program as ONE asm statement.\n")
        f.write(f"#define UDM_FWD64_LDS_BYTES {LDS_TOTAL}\n")
        # (v254 = timeline stamps in the diagnostic build; v255 is the compiler's: the thread-id operand)
        clob = [f'"v{i}"' for i in range(255)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(36, 100)] + ['"vcc"', '"scc"', '"m0"', '"memory"']
        f.write("#define UDM_FWD64_CLOBBERS " + ", ".join(clob) + "\n")
        ABL = 0
        prog, counts = emit_one(f, "")
        for a in ablations:
            ABL = a
            emit_one(f, f"_ABL{a}")
        ABL = 0
    return prog, counts


if __name__ == "__main__":
    prog, counts = emit(sys.argv[1] if len(sys.argv) > 1 else "attention_fwd64_gen.h", [int(x) for x in sys.argv[2:]])
    probs = lint([i for i in prog if i.kind != "raw"])
    print(stats(prog))
    for k, (ca, cb) in counts.items():
        print(k, "A", ca, "B", cb)
    for x in probs[:40]:
        print("LINT", x)
    print(len(probs), "lint problems", "| sgprs up to", S_.next - 1, "| vgprs up to", V.next - 1)
