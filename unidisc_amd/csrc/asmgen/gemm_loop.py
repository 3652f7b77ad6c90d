"""Generator of the K loop of the one-wave-per-SIMD bf16 GEMM (gemm_quad.hip) as ONE hand-scheduled asm statement.

The C++ kernel keeps everything around the loop (tile order, epilogues, split-K, paired problems); this statement replaces its prologue + K loop when
the K-tile count is even and >= 4.  Tile (64 FM) x 256 x 64, four waves as 2 x 2, a wave owns (32 FM) x 128 outputs = FM x 4 accumulators of
v_mfma_f32_32x32x16_bf16.  The accumulators are OPERANDS of the statement (the compiler keeps rows 0-3 in the accumulator file and row 4 in arch VGPRs,
exactly as the C++ loop had them); everything else lives in fixed registers declared as clobbers.

    forms      NT: C = A B^T, both operands K-contiguous (128-byte LDS rows, XOR-swizzled 16-byte k-slots, ds_read_b128 fragments)
               (NN / TN: B / both operands K-major, gathered with ds_read_b64_tr_b16 - same schedule, other address arithmetic)
    LDS        [A stage 0][A stage 1][B stage 0][B stage 1], the C++ kernel's own image (source-side swizzle of the LDS-DMA pieces)
    pipeline   tile t is computed from stage t & 1 in four k-steps of FM x 4 MFMAs; the fragments of k-step s + 1 are read into the other register buffer
               under the MFMAs of k-step s with COUNTED lgkmcnt waits in front of the first MFMA that needs each of them; ONE barrier per tile, BND MFMAs
               into k-step 3 (all fragment reads of the stage are done, this wave's pieces of tile t + 1 have landed); behind it the stage is refilled with
               tile t + 2: N1 pieces under the rest of k-step 3, the others under k-steps 0-1 of the next tile - every piece is in flight for at least two
               k-steps before the boundary that waits for it.
`isa.lint` checks the software-visible hazards, `emu_gemm.py` executes the statement on the CPU emulator against numpy (tests/test_asmgen.py).
"""
import re
import sys
from isa import *   # noqa: F401,F403

BK, FN = 64, 4
CAP = 4.0
import os
SNAKE = int(os.environ.get("UDM_GEMMLOOP_SNAKE", "1"))   # rows alternate direction: exactly one operand changes from an MFMA to the next (A-B measured +0.6..2.5 %)
COST = {"dma": 1.5, "label": 0.0, "comment": 0.0, "need": 0.0}


class Gemm:
    def __init__(self, FM, mode=0, timeline=False):
        assert FM in (3, 4, 5) and mode == 0
        self.FM, self.mode, self.timeline = FM, mode, timeline
        self.A_BYTES, self.B_BYTES = 64 * FM * BK * 2, 256 * BK * 2
        self.A_PW, self.B_PW = self.A_BYTES // 4096, self.B_BYTES // 4096      # 1 KiB pieces per wave and K tile
        self.LOADS = self.A_PW + self.B_PW
        self.B0 = 2 * self.A_BYTES
        self.LDS_BYTES = 2 * (self.A_BYTES + self.B_BYTES)
        self.NMF = FM * FN
        self.BND = FN
        self.N1 = 6                                   # pieces of tile t + 2 under the rest of k-step 3
        V = Alloc("v", 0, 192)
        S = Alloc("s", 36, 72)
        self.V, self.S = V, S
        self.fa = [[V(f"fa{b}{i}", 4) for i in range(FM)] for b in range(2)]
        self.fb = [[V(f"fb{b}{j}", 4) for j in range(FN)] for b in range(2)]
        self.offa = [V(f"offa{j}") for j in range(self.A_PW)]
        self.offb = [V(f"offb{j}") for j in range(self.B_PW)]
        self.addr_a = [V(f"aa{k}") for k in range(4)]
        self.addr_b = [V(f"ab{k}") for k in range(4)]
        self.tmp = [V(f"t{k}") for k in range(8)]
        self.v_last = V.next - 1
        self.s_asrc, self.s_bsrc = S("asrc", 2, 2), S("bsrc", 2, 2)
        self.s_lda, self.s_ldb, self.s_lds, self.s_nk = S("lda"), S("ldb"), S("lds"), S("nk")
        self.s_wave, self.s_dsta, self.s_dstb, self.s_loop = S("wave"), S("dsta"), S("dstb"), S("loop")
        self.s_t = [S(f"t{k}") for k in range(4)]
        self.INPUTS = ["asrc", "bsrc", "lda", "ldb", "lds", "nk"]
        if timeline:   # diagnostic build: cycles of the whole loop, and of the boundary's two waits, summed over the tiles -> 4 dwords per wave at `tl`
            self.s_tl = S("tl", 2, 2)
            self.s_tm = [S(f"tm{k}", 2, 2) for k in range(3)]
            self.s_sum = [S(f"sum{k}") for k in range(3)]       # vmcnt wait, barrier wait, start stamp
            self.INPUTS.append("tl")
        self.s_last = S.next - 1

    # ---- registers of the accumulators in the EMULATED program; the emitted text names them %0 .. %(4 FM - 1)
    def acc(self, i, j):
        if i < 4:
            return R("a", (i * FN + j) * 16, 16)
        return R("v", 192 + j * 16, 16)

    # ---- pieces of the stream
    def frag_reads(self, st, kk, buf):
        """the FN + FM fragments of k-step kk of the tile in stage st -> register buffer buf: B first (every MFMA row needs all of them), then A row by row"""
        out = []
        for n in range(FN):
            out.append(ds_read_b128(self.fb[buf][n], self.addr_b[kk], st * self.B_BYTES + n * 4096))
        for m in range(self.FM):
            out.append(ds_read_b128(self.fa[buf][m], self.addr_a[kk], st * self.A_BYTES + m * 4096))
        return out

    def dma_piece(self, st, j):
        if j < self.A_PW:
            return [s_add_u32(M0, self.s_dsta, st * self.A_BYTES + j * 1024), s_nop(0), global_load_lds_dwordx4(self.offa[j], self.s_asrc)]
        j -= self.A_PW
        return [s_add_u32(M0, self.s_dstb, st * self.B_BYTES + j * 1024), s_nop(0), global_load_lds_dwordx4(self.offb[j], self.s_bsrc)]

    def advance_src(self):
        return [s_add_u32(self.s_asrc[0], self.s_asrc[0], BK * 2), s_addc_u32(self.s_asrc[1], self.s_asrc[1], 0),
                s_add_u32(self.s_bsrc[0], self.s_bsrc[0], BK * 2), s_addc_u32(self.s_bsrc[1], self.s_bsrc[1], 0)]

    def mfmas(self, buf):
        """the FM x 4 MFMAs of one k-step, each with the fragments it is the first to need"""
        out = []
        for i in range(self.FM):
            js = list(range(FN)) if (i % 2 == 0 or not SNAKE) else list(range(FN - 1, -1, -1))     # snake: exactly ONE operand changes from an MFMA to the next
            for n, j in enumerate(js):
                pre = []
                if i == 0:
                    pre.append(need(self.fb[buf][j]))
                if n == 0:
                    pre.append(need(self.fa[buf][i]))
                out.append(pre + [v_mfma_f32_32x32x16_bf16(self.acc(i, j), self.fa[buf][i], self.fb[buf][j], self.acc(i, j))])
        return out

    def body(self, ST, variant):
        """one K tile from stage ST.  variant 'loop' (a next tile and a refill exist), 'prelast' (tile nk - 2: nothing left to refill), 'last'"""
        FMN = self.NMF
        prog = [comment(f"---- K tile, stage {ST}, {variant}")]
        part2 = [x for j in range(self.N1, self.LOADS) for x in self.dma_piece(ST ^ 1, j)]     # rest of tile t + 1
        part1 = [x for j in range(self.N1) for x in self.dma_piece(ST, j)]                         # first pieces of tile t + 2
        half = (len(part2) // 3 + 1) // 2 * 3
        for kk in range(4):
            buf = kk & 1
            m = self.mfmas(buf)
            streams = []
            if kk < 3:
                streams.append((self.frag_reads(ST, kk + 1, buf ^ 1), 0, FMN - 2))
                if variant != "last" and kk < 2:
                    streams.append((part2[:half] if kk == 0 else part2[half:], 0, FMN - 1))
            else:
                if variant != "last":
                    # boundary: every fragment of this stage is in registers, this wave's pieces of the next tile have landed; behind the barrier every
                    # wave's have, and the stage just read may be refilled
                    if self.timeline:
                        z = [s_waitcnt(lgkmcnt=0)]
                        m[self.BND] = ([need_all(), s_memtime(self.s_tm[0])] + z + [s_waitcnt(vmcnt=0), s_memtime(self.s_tm[1])] + z + [s_barrier(), s_memtime(self.s_tm[2])] + z +
                                       [s_sub_u32(self.s_t[3], self.s_tm[1][0], self.s_tm[0][0]), s_add_u32(self.s_sum[0], self.s_sum[0], self.s_t[3]),
                                        s_sub_u32(self.s_t[3], self.s_tm[2][0], self.s_tm[1][0]), s_add_u32(self.s_sum[1], self.s_sum[1], self.s_t[3])] + m[self.BND])
                    else:
                        m[self.BND] = [need_all(), s_waitcnt(vmcnt=0), s_barrier()] + m[self.BND]
                    streams.append((self.frag_reads(ST ^ 1, 0, 0), self.BND, FMN - 2))
                    if variant == "loop":
                        streams.append((self.advance_src() + part1, self.BND, FMN - 1))
            out, _ = schedule_gaps(m, streams, f"k-step {kk}", CAP, COST)
            prog += out
        return prog

    def entry(self, first_input):
        """operands -> fixed registers, per-lane offsets and addresses, tile 0 + the first pieces of tile 1 in flight, tile 0 landed, its first fragments
        requested"""
        raw = lambda t: Inst(t, "raw")
        p = [comment("---- entry")]
        for k, name in enumerate(self.INPUTS):
            r = self.S.names[name]
            p.append(raw(f"s_mov_b{64 if r.n == 2 else 32} {r}, %{first_input + k}"))
        t = self.tmp
        tid = t[0]
        p += [raw(f"v_mov_b32 {tid}, %{first_input + len(self.INPUTS)}"), s_nop(0)]
        p += [v_lshrrev_b32(t[1], 6, tid), s_nop(0), v_readfirstlane_b32(self.s_wave, t[1]), v_and_b32(t[0], 63, tid)]      # t0 = lane
        lane = t[0]
        l31, hi, lrow, lslot, sw = t[1], t[2], t[3], t[4], t[5]
        p += [v_and_b32(l31, 31, lane), v_lshrrev_b32(hi, 5, lane), v_lshrrev_b32(lrow, 3, lane), v_and_b32(lslot, 7, lane)]
        # ---- LDS-DMA sources: piece j of this wave = tile rows r = (wave PW + j) 8 + lrow, LDS slot lslot holds k-slot lslot ^ ((r >> 1) & 7):
        #      offset = r ld + ((lslot ^ ((lrow >> 1) | (((wave PW + j) & 1) << 2))) << 4)        (ld in bytes)
        for offs, PW, ld in ((self.offa, self.A_PW, self.s_lda), (self.offb, self.B_PW, self.s_ldb)):
            p += [s_mul_i32(self.s_t[0], self.s_wave, PW)]
            for j in range(PW):
                p += [s_add_u32(self.s_t[1], self.s_t[0], j), s_and_b32(self.s_t[2], self.s_t[1], 1), s_lshl_b32(self.s_t[2], self.s_t[2], 2), s_lshl_b32(self.s_t[1], self.s_t[1], 3),
                      v_lshrrev_b32(t[6], 1, lrow), v_or_b32(t[6], self.s_t[2], t[6]), v_xor_b32(t[6], lslot, t[6]), v_lshlrev_b32(t[6], 4, t[6]),
                      v_add_u32(t[7], self.s_t[1], lrow), v_mul_lo_u32(t[7], t[7], ld), v_add_u32(offs[j], t[7], t[6])]
        p += [s_mul_i32(self.s_t[0], self.s_wave, self.A_PW * 1024), s_add_u32(self.s_dsta, self.s_lds, self.s_t[0])]
        p += [s_mul_i32(self.s_t[0], self.s_wave, self.B_PW * 1024), s_add_u32(self.s_dstb, self.s_lds, self.s_t[0]), s_add_u32(self.s_dstb, self.s_dstb, self.B0)]
        # ---- fragment addresses: row (wm 32 FM + l31) resp. (wn 128 + l31) of the stage, 128-byte rows, k-slot (2 kk + hi) ^ ((l31 >> 1) & 7)
        p += [s_lshr_b32(self.s_t[0], self.s_wave, 1), s_mul_i32(self.s_t[0], self.s_t[0], 32 * self.FM * 128), s_add_u32(self.s_t[0], self.s_t[0], self.s_lds)]     # wm
        p += [s_and_b32(self.s_t[1], self.s_wave, 1), s_lshl_b32(self.s_t[1], self.s_t[1], 14), s_add_u32(self.s_t[1], self.s_t[1], self.s_lds), s_add_u32(self.s_t[1], self.s_t[1], self.B0)]
        p += self.frag_addresses(lane, l31, hi, sw, t)
        # ---- tile 0, first pieces of tile 1
        for j in range(self.LOADS):
            p += self.dma_piece(0, j)
        p += self.advance_src()
        for j in range(self.N1):
            p += self.dma_piece(1, j)
        p += [s_waitcnt(vmcnt=self.N1), s_barrier()]
        if self.timeline:
            p += [s_mov_b32(self.s_sum[0], 0), s_mov_b32(self.s_sum[1], 0), s_memtime(self.s_tm[0]), s_waitcnt(lgkmcnt=0), s_mov_b32(self.s_sum[2], self.s_tm[0][0])]
        p += self.first_reads()
        p += [s_sub_u32(self.s_loop, self.s_nk, 2), s_lshr_b32(self.s_loop, self.s_loop, 1)]
        return p

    def timeline_store(self):
        if not self.timeline:
            return []
        t = self.tmp
        return [s_memtime(self.s_tm[0]), s_waitcnt(lgkmcnt=0), s_sub_u32(self.s_t[3], self.s_tm[0][0], self.s_sum[2]),
                v_mov_b32(t[0], self.s_t[3]), v_mov_b32(t[1], self.s_sum[0]), v_mov_b32(t[2], self.s_sum[1]), v_mov_b32(t[3], self.s_nk), v_mov_b32(t[4], 0), s_nop(1),
                global_store_dwordx4(t[4], R("v", t[0].idx, 4), self.s_tl, 0), s_waitcnt(vmcnt=0)]

    def frag_addresses(self, lane, l31, hi, sw, t):
        """per-lane fragment addresses (s_t[0] / s_t[1] hold the wave's A / B row bases): row l31, k-slot (2 kk + hi) ^ ((l31 >> 1) & 7) of 128-byte rows"""
        p = [v_lshrrev_b32(sw, 1, l31), v_and_b32(sw, 7, sw), v_lshlrev_b32(t[6], 7, l31)]
        for kk in range(4):
            p += [v_add_u32(t[7], 2 * kk, hi), v_xor_b32(t[7], t[7], sw), v_lshlrev_b32(t[7], 4, t[7]), v_add_u32(t[7], t[7], t[6]),
                  v_add_u32(self.addr_a[kk], self.s_t[0], t[7]), v_add_u32(self.addr_b[kk], self.s_t[1], t[7])]
        return p

    def first_reads(self):
        return self.frag_reads(0, 0, 0)

    def build(self):
        """the whole statement: entry, (nk - 2) / 2 trips of two tiles, the last two tiles"""
        first_input = 4 * self.FM
        nfr = FN + self.FM
        start_q = [ins.writes for ins in self.frag_reads(0, 0, 0)]
        prog = self.entry(first_input)
        prog = resolve_needs(prog, [])[0]
        loop = [label("L_loop")] + self.body(0, "loop") + self.body(1, "loop")
        loop, q = resolve_needs(loop, start_q)
        assert [sorted(x) for x in q] == [sorted(x) for x in start_q], "the loop must leave the fragment queue as it found it"
        prog += loop + [s_sub_u32(self.s_loop, self.s_loop, 1), s_cmp_lg_u32(self.s_loop, 0), s_cbranch_scc1("L_loop")]
        tail = self.body(0, "prelast") + self.body(1, "last")
        tail, q = resolve_needs(tail, start_q)
        assert q == []
        prog += tail + [s_nop(15), s_nop(15)]      # the last MFMAs' results -> the epilogue's reads (wait states the compiler does not know it owes)
        prog += self.timeline_store()
        assert nfr <= 15
        return prog

    # ---- emission
    def asm_text(self, prog):
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

            def sub(mo):
                kind, lo, hi = mo.group(1), int(mo.group(2)), int(mo.group(3))
                if hi - lo == 15 and kind == "a" and lo % 16 == 0:
                    return f"%{lo // 16}"
                if hi - lo == 15 and kind == "v" and lo >= 192:
                    return f"%{16 + (lo - 192) // 16}"
                return mo.group(0)
            t = re.sub(r"\b([av])\[(\d+):(\d+)\]", sub, t)
            lines.append(t)
        return lines

    def clobbers(self):
        return [f'"v{i}"' for i in range(self.v_last + 1)] + [f'"s{i}"' for i in range(36, self.s_last + 1)] + ['"vcc"', '"scc"', '"m0"', '"memory"']


class Gemm16(Gemm):
    """The same K loop on v_mfma_f32_16x16x32_bf16 (round 5: on random bf16 operands the matrix pipe sustains 2.07 PF in this form against 1.85 PF in the 32x32x16
    form - experiments/ubench/mfma_power.hip - and the loop is power bound).  A 32 x 32 accumulator block is four 16 x 16 sub-blocks of 4 registers; the accumulators
    are PINNED (physical-register constraints: rows 0-3 in a[0:255], row 4 in v[192:255]) so that the text can name sub-ranges.  Per 32-k step a wave reads 2 FM + 8
    fragments of 1 KiB (the same bytes per flop as before) for 16 FM MFMAs: the B fragments double-buffered, an A fragment re-read into its own registers as soon as
    its row's last MFMA has issued; two k-steps per K tile, the boundary at the top of the second one, all of tile t + 2's refills behind it."""

    def __init__(self, FM, timeline=False):
        Gemm.__init__(self, FM, 0, timeline)
        V = Alloc("v", 0, 192)
        self.V = V
        self.fa = [V(f"fa{r}", 4) for r in range(2 * FM)]                     # single buffer: 16-row fragments of the current k-step
        self.fb = [[V(f"fb{b}{c}", 4) for c in range(8)] for b in range(2)]
        self.offa = [V(f"offa{j}") for j in range(self.A_PW)]
        self.offb = [V(f"offb{j}") for j in range(self.B_PW)]
        self.addr_a = [V(f"aa{k}") for k in range(2)]
        self.addr_b = [V(f"ab{k}") for k in range(2)]
        self.tmp = [V(f"t{k}") for k in range(8)]
        self.v_last = V.next - 1
        self.NMF = 2 * FM * 8
        self.N1 = self.LOADS

    def acc16(self, ri, cj):
        return self.acc(ri >> 1, cj >> 1).sub((((ri & 1) * 2) + (cj & 1)) * 4, 4)

    def a_read(self, st, kk2, ri):
        return ds_read_b128(self.fa[ri], self.addr_a[kk2], st * self.A_BYTES + ri * 2048)

    def b_reads(self, st, kk2, buf):
        return [ds_read_b128(self.fb[buf][c], self.addr_b[kk2], st * self.B_BYTES + c * 2048) for c in range(8)]

    def mfmas16(self, buf):
        out = []
        for ri in range(2 * self.FM):
            cs = list(range(8)) if ri % 2 == 0 else list(range(7, -1, -1))     # snake: one operand changes per MFMA
            for n, cj in enumerate(cs):
                pre = []
                if ri == 0:
                    pre.append(need(self.fb[buf][cj]))
                if n == 0:
                    pre.append(need(self.fa[ri]))
                d = self.acc16(ri, cj)
                out.append(pre + [v_mfma_f32_16x16x32_bf16(d, self.fa[ri], self.fb[buf][cj], d)])
        return out

    def body(self, ST, variant):
        prog = [comment(f"---- K tile, stage {ST}, {variant} (16x16x32)")]
        n = self.NMF
        for kk2 in range(2):
            m = self.mfmas16(kk2)
            streams = []
            nxt_st, nxt_kk = (ST, 1) if kk2 == 0 else (ST ^ 1, 0)
            have_next = kk2 == 0 or variant != "last"
            if kk2 == 1 and variant != "last":
                # boundary: every fragment of this stage is in registers (the second k-step's were requested under the first), this wave's pieces of the next tile
                # have landed; behind the barrier the stage may be refilled and the next tile read
                if self.timeline:
                    z = [s_waitcnt(lgkmcnt=0)]
                    m[0] = ([need_all(), s_memtime(self.s_tm[0])] + z + [s_waitcnt(vmcnt=0), s_memtime(self.s_tm[1])] + z + [s_barrier(), s_memtime(self.s_tm[2])] + z +
                            [s_sub_u32(self.s_t[3], self.s_tm[1][0], self.s_tm[0][0]), s_add_u32(self.s_sum[0], self.s_sum[0], self.s_t[3]),
                             s_sub_u32(self.s_t[3], self.s_tm[2][0], self.s_tm[1][0]), s_add_u32(self.s_sum[1], self.s_sum[1], self.s_t[3])] + m[0])
                else:
                    m[0] = [need_all(), s_waitcnt(vmcnt=0), s_barrier()] + m[0]
                if variant == "loop":
                    streams.append((self.advance_src() + [x for j in range(self.LOADS) for x in self.dma_piece(ST, j)], 0, n - 9))
            if have_next:
                streams.append((self.b_reads(nxt_st, nxt_kk, kk2 ^ 1), 0, n - 2))
                for ri in range(2 * self.FM):     # an A fragment's registers are free once its row's last MFMA has issued
                    streams.append(([self.a_read(nxt_st, nxt_kk, ri)], ri * 8 + 7, min(n - 1, ri * 8 + 15)))
            out, _ = schedule_gaps(m, streams, f"k-step {kk2}", 2.5, COST)
            prog += out
        return prog

    def frag_addresses(self, lane, l31, hi, sw, t):
        """row (lane & 15) of a 16-row fragment, k-slot (4 kk2 + (lane >> 4)) ^ ((lane & 15) >> 1): 16 consecutive lanes cover all 64 banks once"""
        r16, kg = t[1], t[2]
        p = [v_and_b32(r16, 15, lane), v_lshrrev_b32(kg, 4, lane), v_lshrrev_b32(sw, 1, r16), v_lshlrev_b32(t[6], 7, r16)]
        for kk2 in range(2):
            p += [v_add_u32(t[7], 4 * kk2, kg), v_xor_b32(t[7], t[7], sw), v_lshlrev_b32(t[7], 4, t[7]), v_add_u32(t[7], t[7], t[6]),
                  v_add_u32(self.addr_a[kk2], self.s_t[0], t[7]), v_add_u32(self.addr_b[kk2], self.s_t[1], t[7])]
        return p

    def all_first(self):
        return self.b_reads(0, 0, 0) + [self.a_read(0, 0, ri) for ri in range(2 * self.FM)]

    def first_reads(self):
        """the first k-step's fragments, requested in the ORDER the loop's own end-of-tile prefetch uses (the counted waits of every trip assume that order)"""
        byw = {tuple(ins.writes): ins for ins in self.all_first()}
        return [byw[tuple(w)] for w in self.loop_queue_order]

    def build(self):
        first_input = 4 * self.FM
        _, q0 = resolve_needs(self.body(0, "loop") + self.body(1, "loop"), [ins.writes for ins in self.all_first()])
        self.loop_queue_order = q0        # the order in which a tile's last k-step requests the next tile's first fragments
        prog = resolve_needs(self.entry(first_input), [])[0]
        loop, q = resolve_needs([label("L_loop")] + self.body(0, "loop") + self.body(1, "loop"), q0)
        assert q == q0, "the loop must leave the fragment queue as it found it"
        prog += loop + [s_sub_u32(self.s_loop, self.s_loop, 1), s_cmp_lg_u32(self.s_loop, 0), s_cbranch_scc1("L_loop")]
        tail, q = resolve_needs(self.body(0, "prelast") + self.body(1, "last"), q0)
        assert q == []
        prog += tail + [s_nop(15), s_nop(15)] + self.timeline_store()
        return prog

    def asm_text(self, prog):
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
            lines.append(t)
        return lines


def need(reg):
    """marker: the next instruction reads `reg`, written by an LDS read that may still be in flight (resolved into a counted s_waitcnt by resolve_needs)"""
    return Inst(f"; need {reg}", "need", regs=reg.regs())


def need_all():
    return Inst("; need all", "need", regs=None)


def resolve_needs(seq, queue):
    """Replace the `need` markers of a straight-line sequence by `s_waitcnt lgkmcnt(n)` with the LARGEST n that is safe: LDS reads return in order, so a
    fragment is complete once at most (reads issued after it) are outstanding.  `queue` = write sets of the reads in flight on entry, oldest first."""
    q = [list(x) for x in queue]
    out = []
    for ins in seq:
        if ins.kind == "need":
            regs = ins.meta["regs"]
            if regs is None:
                p = len(q) - 1
            else:
                p = max([k for k, w in enumerate(q) if any(r in w for r in regs)], default=-1)
            if p >= 0:
                n = len(q) - 1 - p
                out.append(s_waitcnt(lgkmcnt=min(n, 15)))
                q = q[p + 1:] if n <= 15 else q[len(q) - 15:]
            continue
        if ins.kind == "wait" and ins.meta.get("lgkmcnt") is not None:
            n = ins.meta["lgkmcnt"]
            q = q[len(q) - n:] if n < len(q) else q
        if ins.kind == "lds_rd":
            q.append(list(ins.writes))
        out.append(ins)
    return out, q


def emit(path, fms=(4, 5), timeline=False, mf16=False):
    with open(path, "w") as f:
        f.write("// GENERATED by asmgen/gemm_loop.py - do not edit.  Prologue + K loop of gemm_quad_kernel as one asm statement per tile height.\n")
        progs = {}
        for FM in fms:
            g = Gemm(FM, timeline=timeline)
            prog = g.build()
            progs[FM] = (g, prog)
            if timeline:
                f.write("#define UDM_QUADLOOP_TIMELINE 1\n")
            f.write(f"#define UDM_QUADLOOP_NT{FM}_CLOBBERS " + ", ".join(g.clobbers()) + "\n")
            f.write(f"#define UDM_QUADLOOP_NT{FM}_ASM \\\n")
            for t in g.asm_text(prog):
                f.write(f'  "{t}\\n\\t" \\\n')
            f.write('  ""\n')
        for FM in (fms if mf16 else ()):      # measured and not shipped (`make UDM_QUADLOOP=mf16`): the same loop on v_mfma_f32_16x16x32_bf16, accumulators pinned
            g = Gemm16(FM, timeline=timeline)
            prog = g.build()
            progs[(FM, 16)] = (g, prog)
            f.write(f"#define UDM_QUADLOOP16_NT{FM}_CLOBBERS " + ", ".join(g.clobbers()) + "\n")
            f.write(f"#define UDM_QUADLOOP16_NT{FM}_ASM \\\n")
            for t in g.asm_text(prog):
                f.write(f'  "{t}\\n\\t" \\\n')
            f.write('  ""\n')
    return progs


if __name__ == "__main__":
    progs = emit(sys.argv[1] if len(sys.argv) > 1 else "gemm_loop_gen.h", timeline="timeline" in sys.argv[2:], mf16="mf16" in sys.argv[2:])
    for FM, (g, prog) in progs.items():
        probs = lint([i for i in prog if i.kind != "raw"])
        print("FM", FM, stats(prog), "| vgprs up to", g.v_last, "sgprs up to", g.s_last)
        for x in probs[:20]:
            print("LINT", x)
        print(len(probs), "lint problems")
