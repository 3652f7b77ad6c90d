"""Run the generated GEMM K loop (gemm_loop.py) on the CPU emulator for ONE workgroup = one (64 FM) x 256 output tile and compare the accumulators with numpy."""
import numpy as np
import isa
import gemm_loop as gl


def run(FM=5, K=256, mode="late", seed=0, ld_pad=64, mf16=False):
    rng = np.random.default_rng(seed)
    g = gl.Gemm16(FM) if mf16 else gl.Gemm(FM)
    prog = g.build() + [isa.s_endpgm()]
    BM = 64 * FM
    lda, ldb = K + ld_pad, K + 2 * ld_pad
    A = isa._bf16_round(rng.standard_normal((BM, lda)).astype(np.float32)).astype(np.uint16)
    B = isa._bf16_round(rng.standard_normal((256, ldb)).astype(np.float32)).astype(np.uint16)
    wg = isa.Workgroup(lds_bytes=g.LDS_BYTES, mode=mode)
    a_A, a_B = wg.add_buffer(A), wg.add_buffer(B)
    vals = dict(asrc=a_A, bsrc=a_B, lda=lda * 2, ldb=ldb * 2, lds=0, nk=K // 64)
    waves = []
    for wid in range(4):
        w = isa.Wave(wg, wid)
        for name, val in vals.items():
            r = g.S.names[name]
            w.s[r.idx] = np.uint32(val & 0xFFFFFFFF)
            if r.n == 2:
                w.s[r.idx + 1] = np.uint32(val >> 32)
        w.v[g.tmp[0].idx] = np.arange(64, dtype=np.uint32) + 64 * wid
        waves.append(w)
    steps = isa.run_workgroup(prog, wg, waves)
    Af = isa._bf16_to_f32(A[:, :K].astype(np.uint32)).astype(np.float64)
    Bf = isa._bf16_to_f32(B[:, :K].astype(np.uint32)).astype(np.float64)
    ref = Af @ Bf.T
    C = np.zeros((BM, 256))
    for wid, w in enumerate(waves):
        wm, wn = wid >> 1, wid & 1
        for i in range(FM):
            for j in range(4):
                acc = w.f32(g.acc(i, j))     # [16, 64]
                for r in range(16):
                    if mf16:   # four 16 x 16 sub-blocks of 4 registers: sub-block (ti, tj), register rr -> row ti 16 + 4 (lane >> 4) + rr, column tj 16 + (lane & 15)
                        ti, tj, rr = r >> 3, (r >> 2) & 1, r & 3
                        for gq in range(4):
                            row = wm * 32 * FM + i * 32 + ti * 16 + 4 * gq + rr
                            c0 = wn * 128 + j * 32 + tj * 16
                            C[row, c0:c0 + 16] = acc[r, 16 * gq:16 * gq + 16]
                        continue
                    for h in range(2):
                        row = wm * 32 * FM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h
                        C[row, wn * 128 + j * 32: wn * 128 + j * 32 + 32] = acc[r, 32 * h:32 * h + 32]
    err = np.abs(C - ref).max() / np.abs(ref).max()
    return dict(rel_err=float(err), steps=steps)


if __name__ == "__main__":
    for mf16 in (False, True):
        for FM in (5, 4):
            for mode in ("late", "early"):
                print("16x16x32" if mf16 else "32x32x16", FM, mode, run(FM=FM, mode=mode, mf16=mf16), run(FM=FM, K=512, mode=mode, seed=1, mf16=mf16))
