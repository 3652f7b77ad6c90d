"""Tiny gfx950 assembly toolkit for the hand-scheduled attention kernels: instruction builders, a hazard / wait-count lint and a
functional emulator (one workgroup, wave64) so that a generated instruction stream is checked on the CPU before it ever meets a GPU.

Scope: exactly the opcodes the generators in this directory emit.  Every builder returns an `Inst` that knows its text, the single
registers it reads and writes (for the lint) and how to execute itself on an emulated wave (`emu`).

Memory model of the emulator (what makes a missing wait visible as a wrong RESULT instead of a lucky pass):
  * LDS reads (`ds_read_*`) poison their destination at issue and deliver at the `s_waitcnt lgkmcnt` that covers them;
  * vector-memory operations (LDS-DMA, global loads / stores) take effect at the covering `s_waitcnt vmcnt` in mode "late"
    (read-after-write hazards) or at issue in mode "early" (write-after-read hazards: a refill that lands while somebody still reads);
  * `s_barrier` is a rendezvous of the workgroup's waves; it does not drain anything.
Run both modes; a correct stream gives the same answer in both.
"""
import numpy as np

POISON = np.uint32(0x7FC0DEAD)


class R:
    """Register range: kind 'v' | 'a' | 's', first index, length."""

    def __init__(self, kind, idx, n=1):
        self.kind, self.idx, self.n = kind, idx, n

    def __getitem__(self, k):
        if isinstance(k, slice):
            start = k.start or 0
            stop = self.n if k.stop is None else k.stop
            assert 0 <= start < stop <= self.n
            return R(self.kind, self.idx + start, stop - start)
        assert 0 <= k < self.n, (k, self.n)
        return R(self.kind, self.idx + k, 1)

    def sub(self, start, n):
        assert 0 <= start and start + n <= self.n, (start, n, self.n)
        return R(self.kind, self.idx + start, n)

    def regs(self):
        return [(self.kind, self.idx + i) for i in range(self.n)]

    def __str__(self):
        if self.kind in ("m0", "vcc", "scc"):
            return self.kind
        if self.n == 1:
            return f"{self.kind}{self.idx}"
        return f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    __repr__ = __str__


class Alloc:
    def __init__(self, kind, start, limit):
        self.kind, self.next, self.limit = kind, start, limit
        self.names = {}

    def __call__(self, name, n=1, align=1):
        self.next = (self.next + align - 1) // align * align
        r = R(self.kind, self.next, n)
        self.next += n
        assert self.next <= self.limit, f"out of {self.kind} registers at {name}"
        self.names[name] = r
        return r


M0 = R("m0", 0)
VCC = R("vcc", 0)
SCC = R("scc", 0)


class Inst:
    __slots__ = ("text", "kind", "reads", "writes", "emu", "meta")

    def __init__(self, text, kind, reads=(), writes=(), emu=None, **meta):
        self.text, self.kind, self.emu, self.meta = text, kind, emu, meta
        self.reads = [x for r in reads if isinstance(r, R) for x in r.regs()]
        self.writes = [x for r in writes if isinstance(r, R) for x in r.regs()]

    def __repr__(self):
        return self.text


# ------------------------------------------------------------------------------------------------------------------------------------
# emulator state
# ------------------------------------------------------------------------------------------------------------------------------------
class Wave:
    def __init__(self, wg, wid):
        self.wg, self.wid = wg, wid
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.m0 = 0
        self.scc = 0
        self.vcc = np.zeros(64, bool)
        self.vmq = []      # pending vector-memory operations (closures), oldest first
        self.lgq = []      # pending LDS reads
        self.pc = 0
        self.done = False

    def rf(self, r):   # view of a register range, [n, 64] uint32
        if r.kind == "v":
            return self.v[r.idx:r.idx + r.n]
        if r.kind == "a":
            return self.a[r.idx:r.idx + r.n]
        raise ValueError(r)

    def f32(self, r):
        return self.rf(r).view(np.float32)

    def sget(self, x):   # scalar operand -> python int (uint32)
        if isinstance(x, R):
            if x.kind == "s":
                return int(self.s[x.idx])
            if x.kind == "m0":
                return int(self.m0)
            raise ValueError(x)
        return int(x) & 0xFFFFFFFF

    def sget64(self, x):
        return int(self.s[x.idx]) | (int(self.s[x.idx + 1]) << 32)

    def sset(self, r, val):
        if r.kind == "m0":
            self.m0 = int(val) & 0xFFFFFFFF
        else:
            self.s[r.idx] = np.uint32(int(val) & 0xFFFFFFFF)

    def vsrc_f(self, x):   # VALU float source: VGPR row, SGPR or literal -> array[64] / scalar float32
        if isinstance(x, R):
            if x.kind in "va":
                return self.f32(x)[0]
            return np.uint32(self.sget(x)).view(np.float32)
        if isinstance(x, float):
            return np.float32(x)
        return np.uint32(int(x) & 0xFFFFFFFF).view(np.float32)

    def vsrc_u(self, x):
        if isinstance(x, R):
            if x.kind in "va":
                return self.rf(x)[0]
            return np.uint32(self.sget(x))
        if isinstance(x, float):
            return np.float32(x).view(np.uint32)
        return np.uint32(int(x) & 0xFFFFFFFF)


class Workgroup:
    """LDS + global memory (named numpy byte arrays addressed through fake 64-bit pointers)."""

    def __init__(self, lds_bytes=160 * 1024, mode="late"):
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.mode = mode
        self.bufs = []   # (base, ndarray uint8)
        self.next_base = 0x100000

    def add_buffer(self, arr):
        flat = arr.view(np.uint8).reshape(-1)
        base = self.next_base
        self.bufs.append((base, flat))
        self.next_base = (base + flat.size + 0xFFFFF) // 0x100000 * 0x100000 + 0x100000
        return base

    def gmem(self, addr, nbytes):
        for base, flat in self.bufs:
            if base <= addr and addr + nbytes <= base + flat.size:
                return flat, addr - base
        raise IndexError(f"global access out of bounds: {addr:#x} +{nbytes}")

    def gread(self, addrs, nbytes):   # addrs: [64] int -> [64, nbytes] uint8
        out = np.zeros((len(addrs), nbytes), np.uint8)
        for i, ad in enumerate(addrs):
            flat, off = self.gmem(int(ad), nbytes)
            out[i] = flat[off:off + nbytes]
        return out

    def gwrite(self, addrs, data):
        for i, ad in enumerate(addrs):
            flat, off = self.gmem(int(ad), data.shape[1])
            flat[off:off + data.shape[1]] = data[i]


def _bf16_round(x):   # float32 array -> uint16 bits (round to nearest even, NaN kept)
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    nan = np.isnan(np.ascontiguousarray(x, np.float32))
    r = np.where(nan, np.uint32(0x7FC0), r)
    return r.astype(np.uint32)


def _bf16_to_f32(bits16):
    return (bits16.astype(np.uint32) << 16).view(np.float32)


# ------------------------------------------------------------------------------------------------------------------------------------
# builders
# ------------------------------------------------------------------------------------------------------------------------------------
def _imm(x):
    if isinstance(x, R):
        return str(x)
    if isinstance(x, float):
        if x in (0.0, 0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0):
            return repr(x)
        return hex(int(np.float32(x).view(np.uint32)))
    return str(x) if -16 <= x <= 64 else hex(x & 0xFFFFFFFF)


def comment(text):
    return Inst(f"; {text}", "comment")


def label(name):
    return Inst(f"{name}:", "label", name=name)


def s_nop(n):
    return Inst(f"s_nop {n}", "nop", count=n + 1)


def s_waitcnt(vmcnt=None, lgkmcnt=None):
    parts = []
    if vmcnt is not None:
        parts.append(f"vmcnt({vmcnt})")
    if lgkmcnt is not None:
        parts.append(f"lgkmcnt({lgkmcnt})")

    def emu(w):
        if vmcnt is not None:
            while len(w.vmq) > vmcnt:
                w.vmq.pop(0)()
        if lgkmcnt is not None:
            while len(w.lgq) > lgkmcnt:
                w.lgq.pop(0)()
    return Inst("s_waitcnt " + " ".join(parts), "wait", emu=emu, vmcnt=vmcnt, lgkmcnt=lgkmcnt)


def s_barrier():
    return Inst("s_barrier", "barrier")


def s_setprio(n):
    return Inst(f"s_setprio {n}", "salu")


def s_endpgm():
    return Inst("s_endpgm", "end")


def s_branch(name):
    return Inst(f"s_branch {name}", "branch", target=name, cond=None)


def s_cbranch_scc1(name):
    return Inst(f"s_cbranch_scc1 {name}", "branch", reads=[SCC], target=name, cond=1)


def s_cbranch_scc0(name):
    return Inst(f"s_cbranch_scc0 {name}", "branch", reads=[SCC], target=name, cond=0)


def _salu2(op, fn, sets_scc=None):
    def build(dst, a, b):
        def emu(w):
            x, y = w.sget(a), w.sget(b)
            res, scc = fn(x, y)
            w.sset(dst, res)
            if scc is not None:
                w.scc = int(scc)
        wr = [dst] + ([SCC] if sets_scc else [])
        return Inst(f"{op} {dst}, {_imm(a)}, {_imm(b)}", "salu", reads=[a, b], writes=wr, emu=emu)
    return build


s_add_u32 = _salu2("s_add_u32", lambda x, y: ((x + y) & 0xFFFFFFFF, (x + y) >> 32), True)
s_sub_u32 = _salu2("s_sub_u32", lambda x, y: ((x - y) & 0xFFFFFFFF, x < y), True)
s_mul_i32 = _salu2("s_mul_i32", lambda x, y: ((x * y) & 0xFFFFFFFF, None))
s_mul_hi_u32 = _salu2("s_mul_hi_u32", lambda x, y: ((x * y) >> 32, None))
s_lshl_b32 = _salu2("s_lshl_b32", lambda x, y: ((x << (y & 31)) & 0xFFFFFFFF, ((x << (y & 31)) & 0xFFFFFFFF) != 0), True)
s_lshr_b32 = _salu2("s_lshr_b32", lambda x, y: (x >> (y & 31), (x >> (y & 31)) != 0), True)
s_and_b32 = _salu2("s_and_b32", lambda x, y: (x & y, (x & y) != 0), True)
s_or_b32 = _salu2("s_or_b32", lambda x, y: (x | y, (x | y) != 0), True)


def s_addc_u32(dst, a, b):
    def emu(w):
        t = w.sget(a) + w.sget(b) + w.scc
        w.sset(dst, t)
        w.scc = t >> 32
    return Inst(f"s_addc_u32 {dst}, {_imm(a)}, {_imm(b)}", "salu", reads=[a, b, SCC], writes=[dst, SCC], emu=emu)


def s_subb_u32(dst, a, b):
    def emu(w):
        t = w.sget(a) - w.sget(b) - w.scc
        w.sset(dst, t)
        w.scc = int(t < 0)
    return Inst(f"s_subb_u32 {dst}, {_imm(a)}, {_imm(b)}", "salu", reads=[a, b, SCC], writes=[dst, SCC], emu=emu)


def s_mov_b32(dst, a):
    return Inst(f"s_mov_b32 {dst}, {_imm(a)}", "salu", reads=[a], writes=[dst], emu=lambda w: w.sset(dst, w.sget(a)))


def s_cselect_b32(dst, a, b):
    return Inst(f"s_cselect_b32 {dst}, {_imm(a)}, {_imm(b)}", "salu", reads=[a, b, SCC], writes=[dst], emu=lambda w: w.sset(dst, w.sget(a) if w.scc else w.sget(b)))


def s_mov_b64(dst, a):
    def emu(w):
        if isinstance(a, R):
            w.s[dst.idx], w.s[dst.idx + 1] = w.s[a.idx], w.s[a.idx + 1]
        else:
            w.s[dst.idx], w.s[dst.idx + 1] = np.uint32(a & 0xFFFFFFFF), np.uint32((a >> 32) & 0xFFFFFFFF)
    return Inst(f"s_mov_b64 {dst}, {_imm(a)}", "salu", reads=[a], writes=[dst], emu=emu)


def _scmp(op, fn):
    def build(a, b):
        def emu(w):
            w.scc = int(fn(w.sget(a), w.sget(b)))
        return Inst(f"{op} {_imm(a)}, {_imm(b)}", "salu", reads=[a, b], writes=[SCC], emu=emu)
    return build


s_cmp_lt_u32 = _scmp("s_cmp_lt_u32", lambda x, y: x < y)
s_cmp_lg_u32 = _scmp("s_cmp_lg_u32", lambda x, y: x != y)
s_cmp_eq_u32 = _scmp("s_cmp_eq_u32", lambda x, y: x == y)
s_cmp_ge_u32 = _scmp("s_cmp_ge_u32", lambda x, y: x >= y)


def s_cmp_lg_u64(a, b):
    def emu(w):
        w.scc = int(w.sget64(a) != (b if not isinstance(b, R) else w.sget64(b)))
    return Inst(f"s_cmp_lg_u64 {a}, {_imm(b)}", "salu", reads=[a, b], writes=[SCC], emu=emu)


def s_or_b64(dst, a, b):
    def emu(w):
        t = w.sget64(a) | w.sget64(b)
        w.s[dst.idx], w.s[dst.idx + 1] = np.uint32(t & 0xFFFFFFFF), np.uint32(t >> 32)
        w.scc = int(t != 0)
    return Inst(f"s_or_b64 {dst}, {a}, {b}", "salu", reads=[a, b], writes=[dst, SCC], emu=emu)


# ---- VALU -----------------------------------------------------------------------------------------------------------------------
def _f(x):
    return np.asarray(x, np.float32)


def _valu(op, kind, dst, srcs, fn, text=None):
    def emu(w):
        with np.errstate(all="ignore"):
            res = np.asarray(fn(w, *srcs))
        if res.dtype != np.uint32:
            res = np.ascontiguousarray(np.broadcast_to(res.astype(np.float32), (64,))).view(np.uint32)
        w.rf(dst)[0][:] = res
    return Inst(text or f"{op} {dst}, " + ", ".join(_imm(s) for s in srcs), kind, reads=srcs, writes=[dst], emu=emu)


def v_mov_b32(dst, a):
    return _valu("v_mov_b32", "valu", dst, [a], lambda w, a: np.broadcast_to(w.vsrc_u(a), (64,)).copy())


def v_mbcnt_lane_id(dst):
    """v_mbcnt_lo + v_mbcnt_hi with full masks: the lane index (two instructions, returned as a list)"""
    def emu(w):
        w.rf(dst)[0][:] = np.arange(64, dtype=np.uint32)
    return [Inst(f"v_mbcnt_lo_u32_b32 {dst}, -1, 0", "valu", writes=[dst], emu=lambda w: None),
            Inst(f"v_mbcnt_hi_u32_b32 {dst}, -1, {dst}", "valu", reads=[dst], writes=[dst], emu=emu)]


def v_add_u32(dst, a, b):
    return _valu("v_add_u32", "valu", dst, [a, b], lambda w, a, b: (w.vsrc_u(a).astype(np.uint64) + w.vsrc_u(b)).astype(np.uint32))


def v_mul_lo_u32(dst, a, b):
    return _valu("v_mul_lo_u32", "valu", dst, [a, b], lambda w, a, b: (w.vsrc_u(a).astype(np.uint64) * w.vsrc_u(b)).astype(np.uint32))


def v_lshlrev_b32(dst, sh, a):
    return _valu("v_lshlrev_b32", "valu", dst, [sh, a], lambda w, sh, a: (w.vsrc_u(a).astype(np.uint64) << (int(w.vsrc_u(sh)) & 31)).astype(np.uint32))


def v_lshrrev_b32(dst, sh, a):
    return _valu("v_lshrrev_b32", "valu", dst, [sh, a], lambda w, sh, a: (w.vsrc_u(a) >> np.uint32(int(w.vsrc_u(sh)) & 31)).astype(np.uint32))


def v_and_b32(dst, a, b):
    return _valu("v_and_b32", "valu", dst, [a, b], lambda w, a, b: (w.vsrc_u(a) & w.vsrc_u(b)).astype(np.uint32))


def v_or_b32(dst, a, b):
    return _valu("v_or_b32", "valu", dst, [a, b], lambda w, a, b: (w.vsrc_u(a) | w.vsrc_u(b)).astype(np.uint32))


def v_xor_b32(dst, a, b):
    return _valu("v_xor_b32", "valu", dst, [a, b], lambda w, a, b: (w.vsrc_u(a) ^ w.vsrc_u(b)).astype(np.uint32))


def v_mul_f32(dst, a, b):
    return _valu("v_mul_f32", "valu", dst, [a, b], lambda w, a, b: _f(w.vsrc_f(a)) * _f(w.vsrc_f(b)))


def v_add_f32(dst, a, b):
    return _valu("v_add_f32", "valu", dst, [a, b], lambda w, a, b: _f(w.vsrc_f(a)) + _f(w.vsrc_f(b)))


def v_sub_f32(dst, a, b):
    return _valu("v_sub_f32", "valu", dst, [a, b], lambda w, a, b: _f(w.vsrc_f(a)) - _f(w.vsrc_f(b)))


def v_max_f32(dst, a, b):
    return _valu("v_max_f32", "valu", dst, [a, b], lambda w, a, b: np.fmax(_f(w.vsrc_f(a)), _f(w.vsrc_f(b))))


def v_max3_f32(dst, a, b, c):
    return _valu("v_max3_f32", "valu", dst, [a, b, c], lambda w, a, b, c: np.fmax(np.fmax(_f(w.vsrc_f(a)), _f(w.vsrc_f(b))), _f(w.vsrc_f(c))))


def v_fma_f32(dst, a, b, c, neg_c=False):
    def fn(w, a, b, c):
        x = _f(w.vsrc_f(a)).astype(np.float64) * _f(w.vsrc_f(b)).astype(np.float64)
        z = _f(w.vsrc_f(c)).astype(np.float64)
        return (x - z if neg_c else x + z).astype(np.float32)
    text = f"v_fma_f32 {dst}, {_imm(a)}, {_imm(b)}, {'-' if neg_c else ''}{_imm(c)}"
    return _valu("v_fma_f32", "valu", dst, [a, b, c], fn, text=text)


def v_pk_mul_f32(dst, a, b):
    """dst[0:1] = a[0:1] * b[0:1] (two fp32 products per instruction; even-aligned register pairs)"""
    assert dst.n == 2 and a.n == 2 and b.n == 2 and dst.idx % 2 == 0 and a.idx % 2 == 0 and b.idx % 2 == 0

    def emu(w):
        with np.errstate(all="ignore"):
            res = (w.f32(a) * w.f32(b)).astype(np.float32)
        w.f32(dst)[:] = res
    return Inst(f"v_pk_mul_f32 {dst}, {a}, {b}", "valu", reads=[a, b], writes=[dst], emu=emu)


def v_dot2c_f32_bf16(dst, a, b):
    """dst += a.bf16[0] * b.bf16[0] + a.bf16[1] * b.bf16[1]   (fp32 accumulate)"""
    def fn(w, d, a, b):
        ua, ub = w.vsrc_u(a), w.vsrc_u(b)
        lo = _bf16_to_f32(ua & 0xFFFF).astype(np.float64) * _bf16_to_f32(ub & 0xFFFF).astype(np.float64)
        hi = _bf16_to_f32(ua >> 16).astype(np.float64) * _bf16_to_f32(ub >> 16).astype(np.float64)
        return (w.f32(d)[0].astype(np.float64) + lo + hi).astype(np.float32)
    # kind "dot": gfx940+ needs 3 wait states between a DOT instruction's write and any OTHER kind of VALU instruction that reads it (a chain of the same DOT opcode
    # accumulating into its own destination needs none) - the GPU read a stale accumulator one product short when a v_mov followed the last v_dot2c directly
    return _valu("v_dot2c_f32_bf16", "dot", dst, [dst, a, b], fn, text=f"v_dot2c_f32_bf16 {dst}, {_imm(a)}, {_imm(b)}")


def v_exp_f32(dst, a):
    return _valu("v_exp_f32", "trans", dst, [a], lambda w, a: np.exp2(_f(w.vsrc_f(a))).astype(np.float32))


def v_log_f32(dst, a):
    return _valu("v_log_f32", "trans", dst, [a], lambda w, a: np.log2(_f(w.vsrc_f(a))).astype(np.float32))


def v_rcp_f32(dst, a):
    return _valu("v_rcp_f32", "trans", dst, [a], lambda w, a: (np.float32(1.0) / _f(w.vsrc_f(a))).astype(np.float32))


def v_cvt_pk_bf16_f32(dst, a, b):
    def fn(w, a, b):
        lo = _bf16_round(np.broadcast_to(_f(w.vsrc_f(a)), (64,)))
        hi = _bf16_round(np.broadcast_to(_f(w.vsrc_f(b)), (64,)))
        return (lo | (hi << 16)).astype(np.uint32)
    return _valu("v_cvt_pk_bf16_f32", "valu", dst, [a, b], fn)


def v_accvgpr_read_b32(dst, a):
    return _valu("v_accvgpr_read_b32", "valu", dst, [a], lambda w, a: w.rf(a)[0].copy())


def v_accvgpr_write_b32(dst, a):
    return _valu("v_accvgpr_write_b32", "valu", dst, [a], lambda w, a: np.broadcast_to(w.vsrc_u(a), (64,)).copy())


def v_permlane32_swap_b32(vdst, vsrc):
    """lanes 32-63 of vdst swap with lanes 0-31 of vsrc"""
    def emu(w):
        d, s = w.rf(vdst)[0], w.rf(vsrc)[0]
        t = d[32:].copy()
        d[32:] = s[:32]
        s[:32] = t
    return Inst(f"v_permlane32_swap_b32 {vdst}, {vsrc}", "permlane", reads=[vdst, vsrc], writes=[vdst, vsrc], emu=emu)


def v_cmp_gt_f32(sdst, a, b):
    """sdst (SGPR pair) = per-lane a > b"""
    def emu(w):
        with np.errstate(all="ignore"):
            m = np.broadcast_to(_f(w.vsrc_f(a)) > _f(w.vsrc_f(b)), (64,))
        bits = 0
        for i in range(64):
            if m[i]:
                bits |= 1 << i
        w.s[sdst.idx], w.s[sdst.idx + 1] = np.uint32(bits & 0xFFFFFFFF), np.uint32(bits >> 32)
    return Inst(f"v_cmp_gt_f32 {sdst}, {_imm(a)}, {_imm(b)}", "valu_sgpr", reads=[a, b], writes=[sdst], emu=emu)


def s_memtime(sdst):
    def emu(w):
        w.wg.clock = getattr(w.wg, "clock", 0) + 1
        w.s[sdst.idx], w.s[sdst.idx + 1] = np.uint32(w.wg.clock), np.uint32(0)
        w.lgq.append(lambda: None)
    return Inst(f"s_memtime {sdst}", "smem", writes=[sdst], emu=emu)


def v_writelane_b32(vdst, ssrc, lane):
    def emu(w):
        w.rf(vdst)[0][lane] = np.uint32(w.sget(ssrc))
    return Inst(f"v_writelane_b32 {vdst}, {ssrc}, {lane}", "valu", reads=[ssrc, vdst], writes=[vdst], emu=emu)


def v_readlane_b32(sdst, a, lane):
    return Inst(f"v_readlane_b32 {sdst}, {a}, {lane}", "readlane", reads=[a], writes=[sdst], emu=lambda w: w.sset(sdst, int(w.rf(a)[0][lane])))


def v_readfirstlane_b32(sdst, a):
    return Inst(f"v_readfirstlane_b32 {sdst}, {a}", "readlane", reads=[a], writes=[sdst], emu=lambda w: w.sset(sdst, int(w.rf(a)[0][0])))


def s_load_dwords(sdst, sbase, offset):
    """s_load_dwordx{2,4,8,16}: sdst.n dwords from the 64-bit address in sbase + offset (scalar memory: out of order among themselves - wait lgkmcnt(0))"""
    assert sdst.n in (2, 4, 8, 16) and sdst.idx % min(sdst.n, 4) == 0 and offset % 4 == 0

    def emu(w):
        flat, off = w.wg.gmem(w.sget64(sbase) + offset, 4 * sdst.n)
        vals = flat[off:off + 4 * sdst.n].view(np.uint32).copy()

        def commit():
            w.s[sdst.idx:sdst.idx + sdst.n] = vals
        w.s[sdst.idx:sdst.idx + sdst.n] = POISON
        w.lgq.append(commit)
    return Inst(f"s_load_dwordx{sdst.n} {sdst}, {sbase}, {hex(offset)}", "smem", reads=[sbase], writes=[sdst], emu=emu)


# ---- MFMA -----------------------------------------------------------------------------------------------------------------------
def _unpack_bf16x8(regs4):   # [4, 64] uint32 -> [64 lanes, 8] float32
    lo = _bf16_to_f32((regs4 & 0xFFFF).astype(np.uint32))
    hi = _bf16_to_f32((regs4 >> 16).astype(np.uint32))
    out = np.empty((64, 8), np.float32)
    out[:, 0::2] = lo.T
    out[:, 1::2] = hi.T
    return out


_ROW = np.array([(r & 3) + 8 * (r >> 2) for r in range(16)])


def v_mfma_f32_32x32x16_bf16(d, a, b, c):
    """D[i][j] = sum_k A[i][k] B[k][j] + C[i][j];  A: lane (i = l & 31, k = 8 (l >> 5) + e),  B: lane (j = l & 31, k = 8 (l >> 5) + e),
    C / D: lane (j = l & 31), register r -> i = (r & 3) + 8 (r >> 2) + 4 (l >> 5).   c may be the constant 0."""
    assert d.n == 16 and a.n == 4 and b.n == 4 and (c == 0 or c.n == 16)

    def emu(w):
        A = _unpack_bf16x8(w.rf(a))   # [lane, e]
        B = _unpack_bf16x8(w.rf(b))
        Am = np.zeros((32, 16), np.float32)
        Bm = np.zeros((16, 32), np.float32)
        for h in range(2):
            Am[:, 8 * h:8 * h + 8] = A[32 * h:32 * h + 32]
            Bm[8 * h:8 * h + 8, :] = B[32 * h:32 * h + 32].T
        with np.errstate(all="ignore"):
            Dm = Am.astype(np.float64) @ Bm.astype(np.float64)
        out = np.zeros((16, 64), np.float32)
        for h in range(2):
            out[:, 32 * h:32 * h + 32] = Dm[_ROW + 4 * h, :]
        if c != 0:
            with np.errstate(all="ignore"):
                out = (out.astype(np.float64) + w.f32(c).astype(np.float64)).astype(np.float32)
        w.f32(d)[:] = out
    return Inst(f"v_mfma_f32_32x32x16_bf16 {d}, {a}, {b}, {_imm(c)}", "mfma", reads=[a, b] + ([c] if c != 0 else []), writes=[d], emu=emu,
                acc_chain=(c != 0 and c.kind == d.kind and c.idx == d.idx))


def v_mfma_f32_16x16x32_bf16(d, a, b, c):
    """D[i][j] (16 x 16) = sum_{k < 32} A[i][k] B[k][j] + C[i][j];  A: lane (i = l & 15, k = 8 (l >> 4) + e),  B: lane (j = l & 15, k = 8 (l >> 4) + e),
    C / D: lane (j = l & 15), register r -> i = 4 (l >> 4) + r.   c may be the constant 0."""
    assert d.n == 4 and a.n == 4 and b.n == 4 and (c == 0 or c.n == 4)

    def emu(w):
        A = _unpack_bf16x8(w.rf(a))   # [lane, e]
        B = _unpack_bf16x8(w.rf(b))
        Am = np.zeros((16, 32), np.float32)
        Bm = np.zeros((32, 16), np.float32)
        for g in range(4):
            Am[:, 8 * g:8 * g + 8] = A[16 * g:16 * g + 16]
            Bm[8 * g:8 * g + 8, :] = B[16 * g:16 * g + 16].T
        with np.errstate(all="ignore"):
            Dm = Am.astype(np.float64) @ Bm.astype(np.float64)
        out = np.zeros((4, 64), np.float32)
        for g in range(4):
            out[:, 16 * g:16 * g + 16] = Dm[4 * g:4 * g + 4, :]
        if c != 0:
            with np.errstate(all="ignore"):
                out = (out.astype(np.float64) + w.f32(c).astype(np.float64)).astype(np.float32)
        w.f32(d)[:] = out
    return Inst(f"v_mfma_f32_16x16x32_bf16 {d}, {a}, {b}, {_imm(c)}", "mfma", reads=[a, b] + ([c] if c != 0 else []), writes=[d], emu=emu,
                acc_chain=(c != 0 and c.kind == d.kind and c.idx == d.idx))


# ---- LDS ------------------------------------------------------------------------------------------------------------------------
def _lds_addrs(w, vaddr, offset):
    return (w.rf(vaddr)[0].astype(np.int64) + offset)


def _ds_read(op, nbytes, dst, vaddr, offset, transpose=False):
    assert 0 <= offset < 65536 and dst.n * 4 == nbytes

    def emu(w):
        addrs = _lds_addrs(w, vaddr, offset)
        assert (addrs % (16 if nbytes == 16 else 8) == 0).all(), f"{op}: misaligned LDS address"
        assert addrs.min() >= 0 and addrs.max() + nbytes <= w.wg.lds.size, f"{op}: LDS address out of range"
        w.rf(dst)[:] = POISON

        def commit(addrs=addrs):
            raw = np.stack([w.wg.lds[ad:ad + nbytes] for ad in addrs])   # [64, nbytes]
            words = raw.view(np.uint32).reshape(64, nbytes // 4)
            if transpose:   # ds_read_b64_tr_b16: result lane p (of a 16-lane group), element j = element (p & 3) of the 8 bytes read by lane 4 j + (p >> 2)
                el = raw.view(np.uint16).reshape(64, 4)
                res = np.zeros((64, 4), np.uint16)
                for g in range(4):
                    for p in range(16):
                        for j in range(4):
                            res[16 * g + p, j] = el[16 * g + 4 * j + (p >> 2), p & 3]
                words = res.view(np.uint32).reshape(64, 2)
            w.rf(dst)[:] = words.T
        if w.wg.mode == "early":
            commit()
            w.lgq.append(lambda: None)
        else:
            w.lgq.append(commit)
    return Inst(f"{op} {dst}, {vaddr} offset:{offset}", "lds_rd", reads=[vaddr], writes=[dst], emu=emu)


def ds_read_b128(dst, vaddr, offset=0):
    return _ds_read("ds_read_b128", 16, dst, vaddr, offset)


def ds_read_b64_tr_b16(dst, vaddr, offset=0):
    return _ds_read("ds_read_b64_tr_b16", 8, dst, vaddr, offset, transpose=True)


def ds_write_b64(vaddr, data, offset=0):
    def emu(w):
        addrs = _lds_addrs(w, vaddr, offset)
        assert (addrs % 8 == 0).all()
        vals = w.rf(data).T.copy()   # [64, 2]

        def commit():
            for i, ad in enumerate(addrs):
                w.wg.lds[ad:ad + 8] = vals[i].view(np.uint8)
        commit()   # LDS writes of a wave are seen in order by its own later reads
        w.lgq.append(lambda: None)
    return Inst(f"ds_write_b64 {vaddr}, {data} offset:{offset}", "lds_wr", reads=[vaddr, data], emu=emu)


# ---- vector memory --------------------------------------------------------------------------------------------------------------
def global_load_lds_dwordx4(voff, sbase, offset=0):
    """LDS-DMA: lane i reads 16 bytes at sbase + voff[i] + offset and lands at LDS byte M0 + 16 i."""
    assert -4096 <= offset < 4096

    def emu(w):
        addrs = w.sget64(sbase) + w.rf(voff)[0].astype(np.int64) + offset
        dst = int(w.m0)
        assert dst % 16 == 0 and dst + 1024 <= w.wg.lds.size

        def commit(addrs=addrs, dst=dst):
            data = w.wg.gread(addrs, 16)
            w.wg.lds[dst:dst + 1024] = data.reshape(-1)
        if w.wg.mode == "early":
            commit()
            w.vmq.append(lambda: None)
        else:
            w.vmq.append(commit)
    return Inst(f"global_load_lds_dwordx4 {voff}, {sbase} offset:{offset}", "dma", reads=[voff, sbase, M0], emu=emu)


def global_load_lds_dword(voff, sbase, offset=0):
    """LDS-DMA, 4 bytes per lane: lane i reads a dword at sbase + voff[i] + offset and lands at LDS byte M0 + 4 i."""
    assert -4096 <= offset < 4096

    def emu(w):
        addrs = w.sget64(sbase) + w.rf(voff)[0].astype(np.int64) + offset
        dst = int(w.m0)
        assert dst % 4 == 0 and dst + 256 <= w.wg.lds.size

        def commit(addrs=addrs, dst=dst):
            data = w.wg.gread(addrs, 4)
            w.wg.lds[dst:dst + 256] = data.reshape(-1)
        if w.wg.mode == "early":
            commit()
            w.vmq.append(lambda: None)
        else:
            w.vmq.append(commit)
    return Inst(f"global_load_lds_dword {voff}, {sbase} offset:{offset}", "dma", reads=[voff, sbase, M0], emu=emu)


def global_load_dwordx4(dst, voff, sbase, offset=0):
    assert -4096 <= offset < 4096 and dst.n == 4

    def emu(w):
        addrs = w.sget64(sbase) + w.rf(voff)[0].astype(np.int64) + offset
        w.rf(dst)[:] = POISON

        def commit(addrs=addrs):
            data = w.wg.gread(addrs, 16)
            w.rf(dst)[:] = data.view(np.uint32).reshape(64, 4).T
        w.vmq.append(commit)
    return Inst(f"global_load_dwordx4 {dst}, {voff}, {sbase} offset:{offset}", "vmem_ld", reads=[voff, sbase], writes=[dst], emu=emu)


def global_load_dword(dst, voff, sbase, offset=0):
    assert -4096 <= offset < 4096 and dst.n == 1

    def emu(w):
        addrs = w.sget64(sbase) + w.rf(voff)[0].astype(np.int64) + offset
        w.rf(dst)[:] = POISON

        def commit(addrs=addrs):
            data = w.wg.gread(addrs, 4)
            w.rf(dst)[:] = data.view(np.uint32).reshape(64, 1).T
        w.vmq.append(commit)
    return Inst(f"global_load_dword {dst}, {voff}, {sbase} offset:{offset}", "vmem_ld", reads=[voff, sbase], writes=[dst], emu=emu)


def _gstore(op, n):
    def build(voff, data, sbase, offset=0):
        assert -4096 <= offset < 4096 and data.n == n

        def emu(w):
            addrs = w.sget64(sbase) + w.rf(voff)[0].astype(np.int64) + offset
            vals = np.ascontiguousarray(w.rf(data).T)   # [64, n] uint32, captured at issue
            w.wg.gwrite(addrs, vals.view(np.uint8).reshape(64, 4 * n))
            w.vmq.append(lambda: None)
        return Inst(f"{op} {voff}, {data}, {sbase} offset:{offset}", "vmem_st", reads=[voff, data, sbase], emu=emu)
    return build


global_store_dwordx4 = _gstore("global_store_dwordx4", 4)
global_store_dword = _gstore("global_store_dword", 1)


# ------------------------------------------------------------------------------------------------------------------------------------
# running a program on one workgroup
# ------------------------------------------------------------------------------------------------------------------------------------
def run_workgroup(prog, wg, waves, max_steps=10_000_000):
    labels = {ins.meta["name"]: i for i, ins in enumerate(prog) if ins.kind == "label"}
    steps = 0
    while not all(w.done for w in waves):
        at_barrier = []
        for w in waves:
            if w.done:
                continue
            while True:
                steps += 1
                assert steps < max_steps, "emulation ran away"
                ins = prog[w.pc]
                w.pc += 1
                if ins.kind == "barrier":
                    at_barrier.append(w)
                    break
                if ins.kind == "end":
                    w.done = True
                    while w.vmq:
                        w.vmq.pop(0)()
                    break
                if ins.kind == "branch":
                    cond = ins.meta["cond"]
                    if cond is None or w.scc == cond:
                        w.pc = labels[ins.meta["target"]]
                    continue
                if ins.emu is not None:
                    ins.emu(w)
        live = [w for w in waves if not w.done]
        assert len(at_barrier) == len(live) or not live, "waves disagree on the barrier count"
    return steps


# ------------------------------------------------------------------------------------------------------------------------------------
# lint: software-visible hazards of gfx950 that nothing inside an asm block pads for us
# ------------------------------------------------------------------------------------------------------------------------------------
# (writer kind, reader kind) -> instructions that must lie between them (each intervening instruction counts one wait state, s_nop N
# counts N + 1).  Conservative: MFMA results are given 16 states although the 8-pass form needs 12.
def _need(wk, rk, is_chain):
    if wk == "dot":
        return 0 if rk == "dot" else 4
    if wk == "mfma":
        if rk == "mfma":
            return 0 if is_chain else 16
        return 16
    if wk in ("valu", "trans", "permlane") and rk == "mfma":
        return 2
    if wk == "trans" and rk in ("valu", "trans", "permlane", "valu_sgpr", "lds_wr", "vmem_st", "lds_rd", "dma", "vmem_ld", "readlane"):
        return 1
    if wk in ("valu", "trans") and rk == "permlane":
        return 2
    if wk in ("valu", "trans", "permlane") and rk == "readlane":
        return 1
    if wk == "valu_sgpr" and rk in ("dma", "vmem_ld", "vmem_st"):
        return 5
    if wk == "readlane" and rk in ("dma", "vmem_ld", "vmem_st"):
        return 5
    return 0


def lint(prog, window=24, mfma_states=1):
    """Straight-line hazard check (branches are ignored: call it per basic block or on streams whose branch targets begin with an
    s_nop pad).  Returns a list of problem strings.  `mfma_states`: wait states an intervening MFMA is counted as (default 1 = like any instruction; an 8-pass
    MFMA behind another MFMA cannot issue before the matrix pipe has taken the first one, i.e. it really is worth 8: generators whose dependent VALU work
    follows a few MFMAs behind its producer pass 4)."""
    problems = []
    real = [ins for ins in prog if ins.kind not in ("comment", "label")]
    # SCC is one bit every SALU compare / add / shift overwrites: a reader (s_addc, s_cbranch_scc, s_cselect) must sit directly behind its writer
    for i, ins in enumerate(real):
        if ("scc", 0) in ins.reads and (i == 0 or ("scc", 0) not in real[i - 1].writes):
            problems.append(f"[{i}] {ins.text}: reads SCC but follows {real[i - 1].text if i else 'nothing'}")
    for i, ins in enumerate(real):
        if not ins.reads and not ins.writes:
            continue
        dist = 0
        for j in range(i - 1, max(-1, i - 1 - window * 2), -1):
            prev = real[j]
            if dist >= window:
                break
            wk = prev.kind
            for reg in prev.writes:
                if reg in ins.reads or (reg in ins.writes and wk == "mfma" and ins.kind != "mfma"):
                    chain = wk == "mfma" and ins.kind == "mfma" and ins.meta.get("acc_chain") and reg in ins.writes
                    need = _need(wk, ins.kind, chain)
                    if reg == ("m0", 0) and wk == "salu" and ins.kind == "dma":
                        need = 1
                    if dist < need:
                        problems.append(f"[{i}] {ins.text}  <- [{j}] {prev.text}: {reg[0]}{reg[1]} needs {need} wait states, has {dist}")
                    break
            dist += prev.meta.get("count", 1) if prev.kind == "nop" else (mfma_states if prev.kind == "mfma" else 1)
    return problems


# ------------------------------------------------------------------------------------------------------------------------------------
# gap scheduler
# ------------------------------------------------------------------------------------------------------------------------------------
def schedule_gaps(mfmas, streams, note, cap, cost_table):
    """Merge `streams` of filler instructions into the gaps behind `mfmas` (gap g = behind MFMA g; gap -1 = ahead of the first MFMA).  A stream =
    (instructions, first gap, last gap): its instructions go out in order, none before `first`, all of them by `last`.  Every gap is filled to the same
    issue budget (an in-order wave cannot bank the slack of a thin gap for a fat one): the most urgent stream - remaining cost per remaining gap -
    goes first; a stream past its last gap is flushed whatever the budget."""
    n = len(mfmas)
    out = [comment(note)]
    pos = [0] * len(streams)
    cost = lambda ins: cost_table.get(ins.kind, 1.0)
    rem = [sum(cost(i) for i in ins) for ins, _, _ in streams]
    counts = []
    for g in range(-1, n):
        if g >= 0:
            out += mfmas[g] if isinstance(mfmas[g], list) else [mfmas[g]]
        budget, c = cap, 0.0
        while True:
            best, bu = None, -1.0
            for si, (ins, g0, g1) in enumerate(streams):
                if pos[si] >= len(ins) or g < g0:
                    continue
                u = 1e9 if g >= g1 else rem[si] / (g1 - g + 1)
                if u > bu:
                    best, bu = si, u
            if best is None:
                break
            nxt = streams[best][0][pos[best]]
            if bu < 1e9 and budget - cost(nxt) < -0.25:
                break
            while True:    # an instruction that reads SCC (s_addc, s_cbranch_scc) stays glued to the one before it: another stream's SALU op in between
                out.append(nxt)   # would feed it a foreign carry / condition
                pos[best] += 1
                rem[best] -= cost(nxt)
                budget -= cost(nxt)
                c += cost(nxt)
                ins_b = streams[best][0]
                if pos[best] < len(ins_b) and (("scc", 0) in ins_b[pos[best]].reads or ins_b[pos[best]].kind == "label"):
                    nxt = ins_b[pos[best]]
                else:
                    break
        counts.append(round(c, 1))
    for si, (ins, _, _) in enumerate(streams):
        assert pos[si] == len(ins), (note, si, pos[si], len(ins))
    return out, counts


def to_asm(prog, indent="  "):
    return "\n".join((ins.text if ins.kind == "label" else indent + ins.text) for ins in prog)


def stats(prog):
    from collections import Counter
    return Counter(ins.kind for ins in prog if ins.kind not in ("comment", "label"))
