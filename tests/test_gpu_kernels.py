"""Per-kernel parity tests on a real MI355X (pytest -m gpu): every HIP kernel vs a plain-PyTorch fp32 reference
of the same op (tests/fake_kernels.py, evaluated on CPU copies), called through the C ABI."""
import math

import pytest
import torch

import fake_kernels as R
from golden_utils import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    from unidisc_amd import kernels

    return kernels


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (256, 384, 128), (200, 136, 72), (1024, 2048, 512), (136, 1001, 256), (8, 384, 32), (40, 64, 2056)])
@pytest.mark.parametrize("out_f32", [False, True])
def test_gemm_nt_plain(K, M, N, K_, out_f32):
    a, b = bf(rnd(M, K_, seed=1)), bf(rnd(N, K_, seed=2))
    ref = a.float() @ b.float().t()
    out = K.gemm_nt(a.to(DEV), b.to(DEV), out_dtype=torch.float32 if out_f32 else torch.bfloat16)
    torch.cuda.synchronize()
    tol = 1e-5 if out_f32 else 4e-3
    assert rel_err(out.float().cpu(), ref) < tol
    # transpose-detecting spot check on an asymmetric corner
    assert torch.allclose(out.float().cpu()[0, :4], ref[0, :4], atol=0.05 * ref.abs().max().item())
    assert torch.allclose(out.float().cpu()[:4, 0], ref[:4, 0], atol=0.05 * ref.abs().max().item())


@pytest.mark.parametrize("tile", [192, 256, 320, 0])
@pytest.mark.parametrize("M,N,K_", [(640, 512, 256), (700, 1001, 192), (1280, 2048, 2048), (333, 260, 128)])
def test_gemm_large_tile_kernels(K, tile, M, N, K_):
    """The LDS-DMA / staggered-phase kernels (forced per tile family) incl. ragged M/N edges and every epilogue."""
    a, b, bias = bf(rnd(M, K_, seed=80, scale=0.5)), bf(rnd(N, K_, seed=81, scale=0.3)), rnd(N, seed=82)
    acc = a.float() @ b.float().t()
    ga, gb, gbias = a.to(DEV), b.to(DEV), bias.to(DEV)
    K.gemm_set_tile(tile)
    try:
        ldc = (N + 7) // 8 * 8
        out = torch.zeros((M, ldc), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(ga, gb, out=out, N=N)
        assert rel_err(out.float().cpu()[:, :N], acc) < 4e-3
        assert torch.all(out.cpu()[:, N:] == 0)
        o32 = K.gemm_nt(ga, gb, out_dtype=torch.float32, epilogue=K.EPI_BIAS, bias=gbias)
        assert rel_err(o32.cpu(), acc + bias) < 1e-5
        aux = torch.zeros((M, ldc), dtype=torch.bfloat16, device=DEV)
        g = torch.zeros((M, ldc), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(ga, gb, out=g, N=N, epilogue=K.EPI_BIAS_GELU, bias=gbias, aux=aux)
        u = (acc + bias).bfloat16().float().requires_grad_()
        y = torch.nn.functional.gelu(u, approximate="tanh")
        (gp,) = torch.autograd.grad(y.sum(), u)
        assert rel_err(g.float().cpu()[:, :N], y.detach()) < 6e-3
        assert rel_err(aux.float().cpu()[:, :N], gp) < 6e-3     # aux = bf16(gelu'(u))
        dg = torch.zeros((M, ldc), dtype=torch.bfloat16, device=DEV)
        dbias = torch.zeros(N, dtype=torch.float32, device=DEV)
        K.gemm_nt(ga, gb, out=dg, N=N, epilogue=K.EPI_DGELU, aux=aux, bias=dbias)
        assert rel_err(dg.float().cpu()[:, :N], acc * gp) < 8e-3
        assert torch.allclose(dbias.cpu(), dg.float().cpu()[:, :N].sum(0), atol=2e-2, rtol=2e-3)  # fused bias gradient = column sums
        c0 = rnd(M, N, seed=83)
        c = c0.clone().to(DEV)
        K.gemm_nt(ga, gb, out=c, beta=1.0)
        assert rel_err(c.cpu(), acc + c0) < 1e-5
    finally:
        K.gemm_set_tile(-1)


@pytest.mark.parametrize("M,N,K_", [(5120, 8192, 128), (5120, 8192, 192), (10240, 8192, 320), (2560, 6144, 256), (5120, 2304 + 256, 64 * 5)])
def test_gemm_persistent_blocks_every_epilogue(K, M, N, K_):
    """More than 256 whole tiles: 256 persistent blocks walk them, each staging the next output tile's first K tile from inside the last K
    iteration (even and odd K-tile counts flip the stage parity between tiles), epilogue patches in the stage consumed last.  Every epilogue, and
    bit-identical to the one-block-per-tile launch (UDM_GEMM_PERSIST=0 path via gemm_set_persist)."""
    a, b, bias = bf(rnd(M, K_, seed=180, scale=0.5)), bf(rnd(N, K_, seed=181, scale=0.3)), rnd(N, seed=182)
    acc = a.float() @ b.float().t()
    ga, gb, gbias = a.to(DEV), b.to(DEV), bias.to(DEV)

    def run_all():
        out = K.gemm_nt(ga, gb, N=N)
        o32 = K.gemm_nt(ga, gb, out_dtype=torch.float32, epilogue=K.EPI_BIAS, bias=gbias)
        aux = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        g = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(ga, gb, out=g, N=N, epilogue=K.EPI_BIAS_GELU, bias=gbias, aux=aux)
        dg = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        dbias = torch.zeros(N, dtype=torch.float32, device=DEV)
        K.gemm_nt(ga, gb, out=dg, N=N, epilogue=K.EPI_DGELU, aux=aux, bias=dbias)
        c = torch.full((M, N), 0.25, dtype=torch.float32, device=DEV)
        K.gemm_nt(ga, gb, out=c, beta=1.0)                     # fp32 accumulate into C (the epilogue reads C back)
        return out, o32, aux, g, dg, c, dbias

    res_p = run_all()
    K.gemm_set_persist(0)
    try:
        res_1 = run_all()
    finally:
        K.gemm_set_persist(1)
    for x, y, name in zip(res_p[:6], res_1[:6], ("plain", "bias f32", "aux", "gelu", "dgelu", "beta accumulate")):
        assert torch.equal(x, y), name
    assert torch.allclose(res_p[6], res_1[6], rtol=1e-4, atol=1e-3)   # column sums are fp32 atomics: order differs
    out, o32, aux, g, dg, cacc, dbias = res_p
    assert rel_err(cacc.cpu(), acc + 0.25) < 1e-5
    assert rel_err(out.float().cpu(), acc) < 4e-3
    assert rel_err(o32.cpu(), acc + bias) < 1e-5
    u = (acc + bias).bfloat16().float().requires_grad_()
    y = torch.nn.functional.gelu(u, approximate="tanh")
    (gp,) = torch.autograd.grad(y.sum(), u)
    assert rel_err(g.float().cpu(), y.detach()) < 6e-3 and rel_err(aux.float().cpu(), gp) < 6e-3
    assert rel_err(dg.float().cpu(), acc * gp) < 8e-3
    assert torch.allclose(dbias.cpu(), dg.float().cpu().sum(0), atol=2e-2, rtol=2e-3)   # one coalesced atomic per wave: lane L carries column L of the wave's 64


@pytest.mark.parametrize("Kc,M,N", [(256, 192, 256), (1280, 2048, 2048), (128, 64, 200), (640, 1001, 328), (2560, 6144, 2048), (192, 320, 8192), (10240, 512, 512), (4096, 300, 260)])
def test_gemm_tn_kmajor(K, Kc, M, N):
    """wgrad form C = A^T B with both operands K-major (transposing LDS reads), incl. ragged M/N and padded strides."""
    lda, ldb = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    a, b = torch.zeros(Kc, lda, dtype=torch.bfloat16), torch.zeros(Kc, ldb, dtype=torch.bfloat16)
    a[:, :M], b[:, :N] = bf(rnd(Kc, M, seed=90, scale=0.5)), bf(rnd(Kc, N, seed=91, scale=0.5))
    ref = a[:, :M].float().t() @ b[:, :N].float()
    c0 = rnd(M, N, seed=92)
    out = c0.clone().to(DEV)
    K.gemm_tn(a.to(DEV), b.to(DEV), out, M=M, N=N, beta=0.0)
    assert rel_err(out.cpu(), ref) < 1e-5
    K.gemm_tn(a.to(DEV), b.to(DEV), out, M=M, N=N, beta=1.0)
    assert rel_err(out.cpu(), 2 * ref) < 2e-5  # beta=1 may split K and accumulate atomically
    cs = torch.zeros(lda, device=DEV)
    K.colsum(a.to(DEV), cs)
    assert torch.allclose(cs.cpu()[:M], a[:, :M].float().sum(0), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("Kc,M,N", [(10240, 2048, 2048), (2048, 512, 768), (4096, 256, 256), (256, 2048, 2048), (1024, 304, 264),
                                    (24576, 768, 768), (24576, 3072, 768), (24576, 768, 2304), (6464, 520, 264)])  # UniDisc-S wgrads: 28 / 7 / 9 uneven slices
def test_gemm_tn_splitk_workspace(K, Kc, M, N):
    """few tiles x long K: K split across CUs through a workspace + reduce pass (falls back to the plain kernel when it cannot split)."""
    a, b = bf(rnd(Kc, M, seed=93, scale=0.5)), bf(rnd(Kc, N, seed=94, scale=0.5))
    ref = a.float().t() @ b.float()
    c0 = rnd(M, N, seed=95)
    out = c0.clone().to(DEV)
    K.gemm_tn_splitk(a.to(DEV), b.to(DEV), out, beta=0.0)
    assert rel_err(out.cpu(), ref) < 1e-5
    K.gemm_tn_splitk(a.to(DEV), b.to(DEV), out, beta=1.0)
    assert rel_err(out.cpu(), 2 * ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,Kc,epi", [(640, 512, 256, "none"), (640, 256, 1024, "bias"), (512, 512, 512, "none"), (1024, 256, 256, "bias"), (640, 512, 320, "none")],
                         ids=["fm5_4tiles", "fm5_bias_16tiles", "fm4_8tiles", "fm4_bias_4tiles", "odd_tile_count_keeps_cpp_loop"])
def test_gemm_quad_asm_k_loop_is_bit_identical(K, M, N, Kc, epi):
    """The generated asm K loop of the one-wave-per-SIMD NT GEMM (csrc/asmgen/gemm_loop.py) against the C++ K loop it replaces: same LDS image, same k order per
    output -> the same bits; and both against the fp32 product.  Operands with padded leading dimensions."""
    a, b = bf(rnd(M, Kc + 64, seed=700, scale=0.5)), bf(rnd(N, Kc + 128, seed=701, scale=0.5))
    ga, gb = a.to(DEV)[:, :Kc], b.to(DEV)[:, :Kc]
    bias = rnd(N, seed=702).to(DEV)
    kw = dict(epilogue=K.EPI_BIAS, bias=bias) if epi == "bias" else {}
    outs = {}
    try:
        K.gemm_set_quad(2)
        for flag in (1, 0):
            K.debug_set("gemm_quad_asm", flag)
            outs[flag] = K.gemm_nt(ga, gb, **kw).cpu()
    finally:
        K.debug_set("gemm_quad_asm", -1)
        K.gemm_set_quad(1)
    ref = a[:, :Kc].float() @ b[:, :Kc].float().t() + (bias.cpu() if epi == "bias" else 0)
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[1].float(), ref) < 4e-3      # bf16 output rounding


@pytest.mark.parametrize("M0,M1,N,Kc", [(6144, 2048, 2048, 10240), (768, 256, 512, 1024), (256, 256, 256, 128), (2304, 768, 768, 24576)])
@pytest.mark.parametrize("beta", [0.0, 1.0])
def test_gemm_tn_pair_equals_two_launches(K, M0, M1, N, Kc, beta):
    """Two wgrads in one launch (udm_gemm_tn_pair_bf16: the qkv + out-proj weight gradients of a DiT block share a grid of 256 x 256 tiles): bit-identical to
    the 256-row-tile kernel run on each problem alone (same tiles, same k order), and equal to the fp32 products; different leading dimensions per problem."""
    a0, b0 = bf(rnd(Kc, M0 + 8, seed=490, scale=0.5)), bf(rnd(Kc, N, seed=491, scale=0.5))
    a1, b1 = bf(rnd(Kc, M1, seed=492, scale=0.5)), bf(rnd(Kc, N + 16, seed=493, scale=0.5))
    ga0, gb0, ga1, gb1 = a0.to(DEV)[:, :M0], b0.to(DEV), a1.to(DEV), b1.to(DEV)[:, :N]
    c0, c1 = rnd(M0, N, seed=494), rnd(M1, N, seed=495)
    o0, o1 = c0.clone().to(DEV), c1.clone().to(DEV)
    K.gemm_tn_pair(ga0, gb0, o0, ga1, gb1, o1, beta=beta)
    ref0 = a0[:, :M0].float().t() @ b0.float() + beta * c0
    ref1 = a1.float().t() @ b1[:, :N].float() + beta * c1
    assert rel_err(o0.cpu(), ref0) < 1e-5 and rel_err(o1.cpu(), ref1) < 1e-5
    try:
        K.gemm_set_quad(2)
        K.gemm_set_tile(-1)
        s0, s1 = c0.clone().to(DEV), c1.clone().to(DEV)
        K.gemm_tn(ga0, gb0, s0, M=M0, N=N, beta=beta)
        K.gemm_tn(ga1, gb1, s1, M=M1, N=N, beta=beta)
    finally:
        K.gemm_set_quad(1)
    split = ((M0 + M1) // 256) * (N // 256) <= 128 and Kc >= 1024      # few tiles over a long K: the pair is split in K through the workspace
    if split:    # slice sums in another order than the unsplit kernel: equal up to fp32 rounding
        assert rel_err(o0, s0) < 2e-6 and rel_err(o1, s1) < 2e-6
    else:        # (the single-problem launch may pick 192-row tiles where they fill the chip better: the same k order per output element either way)
        assert torch.equal(o0, s0) and torch.equal(o1, s1)
    with pytest.raises(ValueError):
        K.gemm_tn_pair(ga0[:, :200], gb0, o0[:200], ga1, gb1, o1)


@pytest.mark.parametrize("beta", [0.0, 1.0])
@pytest.mark.parametrize("shapes,Kc", [([(2304, 768), (768, 768), (3072, 768), (768, 3072)], 24576), ([(512, 256), (256, 768)], 2048), ([(256, 256)], 1024)])
def test_gemm_tn_multi_equals_separate_launches(K, shapes, Kc, beta):
    """Up to four wgrads over the same K in ONE split-K launch + ONE reduce (udm_gemm_tn_multi_bf16: UniDisc-S's qkv / out-proj / mlp.0 / mlp.2 weight gradients):
    every output equals the fp32 product and the single-problem launches (to fp32 summation order: the K split differs), different leading dimensions per operand,
    beta on every output; shapes that do not qualify are refused without touching anything."""
    probs, refs = [], []
    for i, (M, N) in enumerate(shapes):
        a, b = bf(rnd(Kc, M + 8 * i, seed=700 + i, scale=0.5)), bf(rnd(Kc, N + 16, seed=720 + i, scale=0.5))
        c = rnd(M, N, seed=740 + i)
        probs.append((a.to(DEV)[:, :M], b.to(DEV)[:, :N], c.clone().to(DEV)))
        refs.append((a[:, :M], b[:, :N], c))
    assert K.gemm_tn_multi(probs, beta=beta)
    for (ga, gb, out), (a, b, c) in zip(probs, refs):
        M, N = c.shape
        if Kc * M * N <= 2 ** 31:
            ref = a.float().t() @ b.float() + beta * c
            assert rel_err(out.cpu(), ref) < 1e-5
        single = c.clone().to(DEV)
        K.gemm_tn_splitk(ga, gb, single, M=M, N=N, beta=beta)
        assert rel_err(out, single) < 3e-6
    # not multiples of 256 / too many tiles / different K: refused, outputs untouched
    o = torch.full((192, 256), 3.0, device=DEV)
    assert not K.gemm_tn_multi([(probs[0][0][:, :192], probs[0][1][:, :256], o)]) and torch.all(o == 3.0)
    big = torch.zeros(4096, 4096, device=DEV)
    assert not K.gemm_tn_multi([(bf(rnd(1024, 4096, seed=1)).to(DEV), bf(rnd(1024, 4096, seed=2)).to(DEV), big)])


@pytest.mark.parametrize("M0,M1,N,Kc", [(768, 256, 512, 1024), (6144, 2048, 2048, 2560)])
def test_gemm_tn_pair_falls_back_when_quad_kernels_are_off(K, M0, M1, N, Kc):
    """udm_gemm_tn_pair_bf16 answers rc = 3 ("not applicable, nothing launched") when the one-wave-per-SIMD kernels are switched off (`gemm_set_quad(0)`, a documented
    A/B switch) or a pointer is off its 16-byte alignment: K.gemm_tn_pair then issues the two plain problems instead of raising (round-3 advisor finding)."""
    a0, b0 = bf(rnd(Kc, M0, seed=590, scale=0.5)), bf(rnd(Kc, N, seed=591, scale=0.5))
    a1, b1 = bf(rnd(Kc, M1, seed=592, scale=0.5)), bf(rnd(Kc, N, seed=593, scale=0.5))
    ref0, ref1 = a0.float().t() @ b0.float(), a1.float().t() @ b1.float()
    o0, o1 = torch.zeros(M0, N, device=DEV), torch.zeros(M1, N, device=DEV)
    try:
        K.gemm_set_quad(0)
        K.gemm_tn_pair(a0.to(DEV), b0.to(DEV), o0, a1.to(DEV), b1.to(DEV), o1)
    finally:
        K.gemm_set_quad(1)
    assert rel_err(o0.cpu(), ref0) < 1e-5 and rel_err(o1.cpu(), ref1) < 1e-5
    # a misaligned operand takes the same route and ends in the plain kernel's own loud alignment error (not a stale message)
    a0m = torch.empty(Kc * M0 + 4, dtype=torch.bfloat16, device=DEV)[4:].view(Kc, M0).copy_(a0)
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        K.gemm_tn_pair(a0m, b0.to(DEV), o0, a1.to(DEV), b1.to(DEV), o1)


@pytest.mark.parametrize("M,N,K_", [(512, 512, 128), (512, 256, 192), (768, 512, 448), (384, 256, 256), (1024, 768, 1024), (640, 512, 320), (2560, 2048, 2048)])
def test_gemm_nt_quad_one_wave_per_simd_every_epilogue(K, M, N, K_):
    """The one-wave-per-SIMD NT kernel (gemm_quad.hip) forced on every shape it fits (192-, 256- and 320-row tiles - the last with its fifth accumulator row in arch VGPRs -, K-tile counts 2 / 3 / 4 / 5 / 7 / 16 / 32):
    every epilogue against fp32, and bit-identical to the 8-wave kernel (same MFMA k order per output element)."""
    a, b, bias = bf(rnd(M, K_, seed=380, scale=0.5)), bf(rnd(N, K_, seed=381, scale=0.3)), rnd(N, seed=382)
    acc = a.float() @ b.float().t()
    ga, gb, gbias = a.to(DEV), b.to(DEV), bias.to(DEV)

    def run_all():
        out = K.gemm_nt(ga, gb, N=N)
        o32 = K.gemm_nt(ga, gb, out_dtype=torch.float32)
        ob = K.gemm_nt(ga, gb, epilogue=K.EPI_BIAS, bias=gbias)
        aux = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        g = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        K.gemm_nt(ga, gb, out=g, N=N, epilogue=K.EPI_BIAS_GELU, bias=gbias, aux=aux)
        dg = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
        dbias = torch.zeros(N, dtype=torch.float32, device=DEV)
        K.gemm_nt(ga, gb, out=dg, N=N, epilogue=K.EPI_DGELU, aux=aux, bias=dbias)
        return out, o32, ob, aux, g, dg, dbias

    try:
        K.gemm_set_quad(2)
        res_q = run_all()
        K.gemm_set_quad(0)
        res_8 = run_all()
    finally:
        K.gemm_set_quad(1)
    for x, y, name in zip(res_q[:6], res_8[:6], ("plain", "f32", "bias", "aux", "gelu", "dgelu")):
        assert torch.equal(x, y), name
    assert torch.allclose(res_q[6], res_8[6], rtol=1e-4, atol=1e-3)   # column sums are fp32 atomics: order differs
    out, o32, ob, aux, g, dg, dbias = res_q
    assert rel_err(out.float().cpu(), acc) < 4e-3 and rel_err(o32.cpu(), acc) < 1e-5
    assert rel_err(ob.float().cpu(), acc + bias) < 4e-3
    u = (acc + bias).bfloat16().float().requires_grad_()
    y = torch.nn.functional.gelu(u, approximate="tanh")
    (gp,) = torch.autograd.grad(y.sum(), u)
    assert rel_err(g.float().cpu(), y.detach()) < 6e-3 and rel_err(aux.float().cpu(), gp) < 6e-3
    assert rel_err(dg.float().cpu(), acc * gp) < 8e-3
    assert torch.allclose(dbias.cpu(), dg.float().cpu().sum(0), atol=2e-2, rtol=2e-3)


@pytest.mark.parametrize("Kc,M,N", [(128, 256, 256), (192, 512, 512), (256, 512, 256), (448, 256, 512), (1024, 768, 512), (320, 384, 256), (640, 192, 512),
                                    (10240, 2048, 2048), (2560, 8192, 2048)])
def test_gemm_tn_quad_one_wave_per_simd(K, Kc, M, N):
    """The one-wave-per-SIMD K-major kernel (gemm_quad.hip), forced on every shape it fits: K-tile counts 2 / 3 / 4 / 7 / 16 / 160 (prologue, the
    unrolled steady-state pair, the run-time tail), 192- and 256-row tiles, beta = 0 / 1, padded strides; against fp32 and against the 8-wave kernel."""
    lda, ldb = M + 8, N + 16
    a, b = torch.zeros(Kc, lda, dtype=torch.bfloat16), torch.zeros(Kc, ldb, dtype=torch.bfloat16)
    a[:, :M], b[:, :N] = bf(rnd(Kc, M, seed=190, scale=0.5)), bf(rnd(Kc, N, seed=191, scale=0.5))
    ref = a[:, :M].float().t() @ b[:, :N].float()
    c0 = rnd(M, N, seed=192)
    ad, bd = a.to(DEV), b.to(DEV)
    try:
        K.gemm_set_quad(2)
        out = c0.clone().to(DEV)
        K.gemm_tn(ad, bd, out, M=M, N=N, beta=0.0)
        assert rel_err(out.cpu(), ref) < 1e-5
        K.gemm_tn(ad, bd, out, M=M, N=N, beta=1.0)
        assert rel_err(out.cpu(), 2 * ref) < 2e-5
        quad = torch.empty(M, N, device=DEV)
        K.gemm_tn(ad, bd, quad, M=M, N=N, beta=0.0)
        K.gemm_set_quad(0)
        old = torch.empty(M, N, device=DEV)
        K.gemm_tn(ad, bd, old, M=M, N=N, beta=0.0)
        assert rel_err(quad.cpu(), old.cpu()) < 1e-6      # same products, fp32 accumulation in a different order
        if M % 256 == 0:                                   # workspace split-K through the quad tiles
            K.gemm_set_quad(2)
            sk = c0.clone().to(DEV)
            K.gemm_tn_splitk(ad[:, :M].contiguous(), bd[:, :N].contiguous(), sk, beta=1.0)
            assert rel_err(sk.cpu(), c0 + ref) < 2e-5
    finally:
        K.gemm_set_quad(1)


@pytest.mark.parametrize("M,N,K_", [(192, 256, 128), (256, 512, 192), (320, 256, 256), (640, 512, 448), (768, 256, 1024), (10240, 2048, 2048), (1280, 2048, 6144), (2560, 2048, 8192)])
def test_gemm_nn_dgrad_from_forward_shadow(K, M, N, K_):
    """NN form (gemm_quad.hip MODE 2): dX = dY W with W in its forward layout [out = K, in = N] - the NT kernel's A side next to the TN kernel's B
    side.  K-tile counts 2 / 3 / 4 / 7 / 16 / 32 / 96 / 128, every tile height, padded strides; against fp32, and bit-equal to the NT kernel fed
    the explicitly transposed weight (same products, same k order)."""
    assert K.gemm_nn_ok(M, N, K_)
    ragged_fills = (M + 8 > 320) and -(-(M + 8) // 320) * (N // 256) >= 128   # (a ragged last tile row is taken when 320-row tiles fill the chip)
    assert (K.gemm_nn_ok(M + 8, N, K_) == ragged_fills) and not K.gemm_nn_ok(M, N + 128, K_) and not K.gemm_nn_ok(M, N, K_ + 32)
    lda, ldb, ldc = K_ + 8, N + 16, N + 8
    a, w = torch.zeros(M, lda, dtype=torch.bfloat16), torch.zeros(K_, ldb, dtype=torch.bfloat16)
    a[:, :K_], w[:, :N] = bf(rnd(M, K_, seed=290, scale=0.5)), bf(rnd(K_, N, seed=291, scale=0.5))
    ref = a[:, :K_].float() @ w[:, :N].float()
    ad, wd = a.to(DEV), w.to(DEV)
    out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=DEV)
    K.gemm_nn(ad[:, :K_], wd[:, :N], out=out[:, :N])
    assert rel_err(out[:, :N].float().cpu(), ref) < 3e-3
    assert torch.all(out[:, N:] == 0)
    try:
        K.gemm_set_quad(2)
        wt = w[:, :N].t().contiguous().to(DEV)   # [N, K]: the transposed shadow the NT dgrad reads
        nt = K.gemm_nt(ad[:, :K_], wt, N=N)
        assert torch.equal(nt, out[:, :N])
    finally:
        K.gemm_set_quad(1)


@pytest.mark.parametrize("M,N,K_", [(9216, 2048, 2048), (9216, 2048, 6144), (5000, 2048, 512), (4100, 2560, 256), (10248, 2048, 128)])
def test_gemm_nn_ragged_last_tile_row(K, M, N, K_):
    """NN form with a row count no whole tile height divides into one round (config E: M = 9216): 320-row tiles, the last tile row hangs over M - its operand
    rows are clamped at the source, its output rows are not stored (the rows behind M keep their sentinel).  Against fp32 and bit-equal to the NT kernels."""
    assert K.gemm_nn_ok(M, N, K_)
    a, w = bf(rnd(M, K_, seed=292, scale=0.5)), bf(rnd(K_, N, seed=293, scale=0.5))
    ref = a.float() @ w.float()
    ad, wd = a.to(DEV), w.to(DEV)
    buf = torch.full((M + 320, N), 7.0, dtype=torch.bfloat16, device=DEV)
    K.gemm_nn(ad, wd, out=buf[:M])
    assert rel_err(buf[:M].float().cpu(), ref) < 3e-3
    assert torch.all(buf[M:] == 7.0)
    wt = w.t().contiguous().to(DEV)
    nt = K.gemm_nt(ad, wt, N=N)               # (one round of ragged 320-row tiles goes to the one-wave-per-SIMD kernel's ragged NT form as well)
    assert torch.equal(nt, buf[:M])
    bias = rnd(N, seed=294).to(DEV)
    ntb = torch.full((M + 320, N), 7.0, dtype=torch.bfloat16, device=DEV)
    K.gemm_nt(ad, wt, out=ntb[:M], N=N, epilogue=K.EPI_BIAS, bias=bias)
    assert torch.all(ntb[M:] == 7.0)
    try:
        K.gemm_set_quad(0)                    # the 8-wave kernels: same products, same k order
        assert torch.equal(K.gemm_nt(ad, wt, N=N), nt)
        assert torch.equal(K.gemm_nt(ad, wt, N=N, epilogue=K.EPI_BIAS, bias=bias), ntb[:M])
    finally:
        K.gemm_set_quad(1)


@pytest.mark.parametrize("cus", [248, 240, 224])
def test_gemm_single_round_shapes_split_by_rows_when_cus_are_held(K, cus):
    """gemm_set_cus(n) (data-parallel runs: a collective's kernels hold CUs): a GEMM of exactly one round of 256 one-workgroup tiles would run two rounds, so the host
    side runs the whole tile rows that fit n CUs and sends the leftover rows through a split-K launch (NN, TN) or small tiles (NT).  Same results: the rows of the
    main part and the NT remainder bit for bit, the split-K remainders to fp32 summation order."""
    M, d, dff = 10240, 2048, 8192
    dy, w = bf(rnd(M, dff, seed=400, scale=0.5)).to(DEV), bf(rnd(dff, d, seed=401, scale=0.3)).to(DEV)     # fc1 dgrad: dX[M, d] = dY[M, 4d] W[4d, d]
    x = bf(rnd(M, d, seed=402, scale=0.5)).to(DEV)
    wt, bias = bf(rnd(d, dff, seed=403, scale=0.3)).to(DEV), rnd(d, seed=404).to(DEV)                       # fc2 forward: [M, 4d] x [d, 4d]^T + bias
    ref_nn = K.gemm_nn(dy, w)
    ref_tn = K.gemm_tn(dy, x, torch.zeros(dff, d, dtype=torch.float32, device=DEV))                         # fc1 wgrad [4d, d] (32 x 8 tiles)
    ref_tn2 = K.gemm_tn(x, dy, torch.zeros(d, dff, dtype=torch.float32, device=DEV))                        # fc2-shaped wgrad [d, 4d] (8 x 32 tiles)
    ref_nt = K.gemm_nt(dy, wt, epilogue=K.EPI_BIAS, bias=bias)
    try:
        K.gemm_set_cus(cus)
        assert not K.gemm_tn_pair_ok(6144, 2048, 2048, M)                                                   # 192 + 64 tiles no longer fit one round
        main = (cus // 8) * 320
        nn = K.gemm_nn(dy, w)
        assert torch.equal(nn[:main], ref_nn[:main]) and rel_err(nn.float().cpu(), ref_nn.float().cpu()) < 2e-3
        tn = K.gemm_tn(dy, x, torch.zeros(dff, d, dtype=torch.float32, device=DEV))
        main_t = (cus // 8) * 256
        assert torch.equal(tn[:main_t], ref_tn[:main_t]) and rel_err(tn.cpu(), ref_tn.cpu()) < 1e-5
        c0 = torch.full((d, dff), 0.5, dtype=torch.float32, device=DEV)
        tn2 = K.gemm_tn(x, dy, c0, beta=1.0)                                                               # beta reaches the split-K remainder too
        assert rel_err(tn2.cpu(), ref_tn2.cpu() + 0.5) < 1e-5
        nt = K.gemm_nt(dy, wt, epilogue=K.EPI_BIAS, bias=bias)
        assert torch.equal(nt, ref_nt)
        # qkv-shaped wgrad [6144, 2048]: 256 tiles of 192 rows -> the leftover rows are a COLUMN SLICE of dY that ends the allocation (192-row tiles in the split-K launch)
        dq = bf(rnd(M, 6144, seed=405, scale=0.5)).to(DEV)
        K.gemm_set_cus(0)
        ref_q = K.gemm_tn(dq, x, torch.zeros(6144, d, dtype=torch.float32, device=DEV))
        K.gemm_set_cus(cus)
        tq = K.gemm_tn(dq, x, torch.zeros(6144, d, dtype=torch.float32, device=DEV))
        assert rel_err(tq.cpu(), ref_q.cpu()) < 1e-5
    finally:
        K.gemm_set_cus(0)
    assert K.gemm_tn_pair_ok(6144, 2048, 2048, M)


@pytest.mark.parametrize("M,N,K_", [(5120, 2048, 48512), (5056, 2048, 4096), (704, 512, 8192), (100, 300, 640), (5120, 2048, 192)])
def test_gemm_nt_splitk_bf16_output(K, M, N, K_):
    """NT split-K with a bf16 result (the head dgrad on the compacted rows): ragged row counts, shapes where it must fall back, long K."""
    a, b = bf(rnd(M, K_, seed=280, scale=0.5)), bf(rnd(N, K_, seed=281, scale=0.3))
    ref = a.float() @ b.float().t()
    out = K.gemm_nt_splitk(a.to(DEV), b.to(DEV))
    assert out.dtype == torch.bfloat16 and rel_err(out.float().cpu(), ref) < 4e-3
    buf = torch.zeros((M, N + 64), dtype=torch.bfloat16, device=DEV)          # row stride != N
    K.gemm_nt_splitk(a.to(DEV), b.to(DEV), out=buf[:, :N])
    assert rel_err(buf[:, :N].float().cpu(), ref) < 4e-3 and torch.all(buf[:, N:] == 0)


@pytest.mark.parametrize("M,N", [(64, 128), (320, 512)])
def test_gemm_gelu_epilogues_extreme_preactivations(K, M, N):
    """GELU / GELU' in the epilogues are written through the logistic function (exp2 + rcp): no inf * 0 for |x| up to 100."""
    K_ = 128
    a, b = bf(torch.full((M, K_), 0.125)), bf(torch.full((N, K_), 0.0625))  # accumulator = 1.0 everywhere
    vals = torch.tensor([-100.0, -40.0, -12.0, -9.0, -5.0, -1.0, -1e-3, 0.0, 1e-3, 1.0, 5.0, 9.0, 12.0, 40.0, 100.0, -0.75])
    bias = vals.repeat(N // vals.numel() + 1)[:N].clone()
    aux = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    g = K.gemm_nt(a.to(DEV), b.to(DEV), epilogue=K.EPI_BIAS_GELU, bias=(bias - 1.0).to(DEV), aux=aux).float().cpu()
    pre = bias.bfloat16().float()[None].expand(M, N)            # accumulator 1.0 + (bias - 1.0), rounded to bf16 like the reference's mlp.0 output
    ref = torch.nn.functional.gelu(pre.double(), approximate="tanh").float()
    assert torch.isfinite(g).all() and torch.allclose(g, ref, atol=2e-3, rtol=8e-3)
    u = pre.double().clone().requires_grad_()
    (gp,) = torch.autograd.grad(torch.nn.functional.gelu(u, approximate="tanh").sum(), u)
    saved = aux.float().cpu()                                     # aux holds bf16(gelu'(u))
    assert torch.isfinite(saved).all() and torch.allclose(saved, gp.float(), atol=2e-3, rtol=8e-3)
    dg = K.gemm_nt(a.to(DEV), b.to(DEV), epilogue=K.EPI_DGELU, aux=aux).float().cpu()
    assert torch.isfinite(dg).all() and torch.allclose(dg, gp.float(), atol=2e-3, rtol=1.2e-2)


def test_gemm_identity_asymmetric(K):
    # A = I (padded), B asymmetric: catches swapped row/col fragment maps
    n = 128
    a = bf(torch.eye(n))
    b = bf(torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 100)
    out = K.gemm_nt(a.to(DEV), b.to(DEV), out_dtype=torch.float32).cpu()
    assert torch.equal(out, b.float().t())


def test_gemm_epilogues(K):
    M, N, K_ = 264, 520, 192
    a, b, bias = bf(rnd(M, K_, seed=3, scale=0.5)), bf(rnd(N, K_, seed=4, scale=0.2)), rnd(N, seed=5)
    acc = a.float() @ b.float().t()
    out = K.gemm_nt(a.to(DEV), b.to(DEV), epilogue=K.EPI_BIAS, bias=bias.to(DEV)).float().cpu()
    assert rel_err(out, acc + bias) < 4e-3
    aux = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    g = K.gemm_nt(a.to(DEV), b.to(DEV), epilogue=K.EPI_BIAS_GELU, bias=bias.to(DEV), aux=aux).float().cpu()
    pre = (acc + bias).bfloat16()
    u = pre.float().requires_grad_()
    y = torch.nn.functional.gelu(u, approximate="tanh")
    (gp,) = torch.autograd.grad(y.sum(), u)
    assert rel_err(g, y.detach()) < 6e-3
    assert rel_err(aux.float().cpu(), gp) < 6e-3                  # the saved GELU derivative (bf16)
    dg = K.gemm_nt(a.to(DEV), b.to(DEV), epilogue=K.EPI_DGELU, aux=aux).float().cpu()
    assert rel_err(dg, acc * gp) < 8e-3
    c0 = rnd(M, N, seed=6)
    c = c0.clone().to(DEV)
    K.gemm_nt(a.to(DEV), b.to(DEV), out=c, beta=1.0)
    assert rel_err(c.cpu(), acc + c0) < 1e-5


def test_gemm_strided_views_and_partial_rows(K):
    # wgrad-style call: A = dY^T [N, M] using only the first `rows` rows, fp32 output into a view
    M, N, Kin, rows = 256, 200, 128, 193
    dyt, xt = bf(rnd(N, M, seed=7)), bf(rnd(Kin, M, seed=8))
    out = torch.zeros((rows, Kin), dtype=torch.float32, device=DEV)
    K.gemm_nt(dyt.to(DEV), xt.to(DEV), out=out, M=rows, N=Kin, K=M)
    assert rel_err(out.cpu(), dyt[:rows].float() @ xt.float().t()) < 1e-5


@pytest.mark.parametrize("R_,C", [(64, 64), (256, 136), (1280, 2048), (8, 192), (200, 72)])
def test_transpose_and_colsum(K, R_, C):
    x = bf(rnd(R_, C, seed=9))
    cs = torch.zeros(C, dtype=torch.float32, device=DEV)
    out = K.transpose(x.to(DEV), colsum=cs)
    assert torch.equal(out.cpu(), x.t().contiguous())
    assert torch.allclose(cs.cpu(), x.float().sum(0), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("R_,C", [(192, 64), (65, 64), (1001, 256), (256, 32)])
def test_cast_transpose(K, R_, C):
    w = rnd(R_, C, seed=10)
    Rp = (R_ + 127) // 128 * 128
    o = torch.zeros((Rp, C), dtype=torch.bfloat16, device=DEV)
    ot = torch.zeros((C, Rp), dtype=torch.bfloat16, device=DEV)
    K.cast_transpose(w.to(DEV), o, ot)
    assert torch.equal(o.cpu()[:R_], w.bfloat16()) and torch.all(o.cpu()[R_:] == 0)
    assert torch.equal(ot.cpu()[:, :R_], w.t().bfloat16()) and torch.all(ot.cpu()[:, R_:] == 0)


def test_cast_transpose_multi_one_launch(K):
    """Every weight of a forward cast in one launch (job table + bisection per block): bit-identical to the single-matrix kernel, incl. ragged
    shapes, a padded shadow (head: rows beyond R stay zero) and jobs with only one of the two outputs."""
    shapes = [(2048, 2048), (200, 136), (8192, 2048), (1001, 72), (64, 64), (130, 520)]
    items, refs = [], []
    for i, (R_, C) in enumerate(shapes):
        w = rnd(R_, C, seed=600 + i).to(DEV)
        Rp = (R_ + 127) // 128 * 128 if i == 3 else R_
        out = torch.zeros(Rp, C, dtype=torch.bfloat16, device=DEV) if i != 4 else None
        out_t = torch.zeros(C, Rp, dtype=torch.bfloat16, device=DEV) if i != 5 else None
        items.append((w, out, out_t))
        o1 = torch.zeros(Rp, C, dtype=torch.bfloat16, device=DEV)
        t1 = torch.zeros(C, Rp, dtype=torch.bfloat16, device=DEV)
        K.cast_transpose(w, o1, t1)
        refs.append((o1, t1))
    jobs = K.cast_transpose_jobs(items, torch.device(DEV))
    K.cast_transpose_multi(jobs)
    torch.cuda.synchronize()
    for (w, out, out_t), (o1, t1) in zip(items, refs):
        if out is not None:
            assert torch.equal(out, o1)
        if out_t is not None:
            assert torch.equal(out_t, t1)
        assert torch.equal(o1[: w.shape[0]], w.bfloat16())


def test_cast_roundtrip(K):
    x = rnd(4099, seed=11)
    y = torch.empty(4099, dtype=torch.bfloat16, device=DEV)
    K.cast_f32_bf16(x.to(DEV), y)
    assert torch.equal(y.cpu(), x.bfloat16())
    K.cast_f32_bf16(x.to(DEV), y, scale=0.125)
    assert torch.equal(y.cpu(), (x.bfloat16().float() * 0.125).bfloat16())
    z = torch.empty(4099, dtype=torch.float32, device=DEV)
    K.cast_bf16_f32(y, z, scale=2.0)
    assert torch.equal(z.cpu(), y.cpu().float() * 2.0)


# ------------------------------------------------------------------------------------------------ norms / residual
def _mod_inputs(B, d, n, seed):
    Bp = (B + 7) // 8 * 8
    mod = torch.zeros(Bp, n * d)
    mod[:B] = rnd(B, n * d, seed=seed, scale=0.3)
    return bf(mod)


@pytest.mark.parametrize("d", [64, 768, 2048])
@pytest.mark.parametrize("nt", [0, 1], ids=["rms", "layernorm"])
@pytest.mark.parametrize("img", [False, True], ids=["all_rows", "image_rows"])
def test_residual_with_fused_modulated_next_norm_equals_separate_kernels(K, d, img, nt):
    """residual add + the NEXT pre-norm in its adaLN-modulated form in one pass (udm_residual_norm_fwd_ada) against residual_fwd followed by the modulated norm_fwd:
    the same arithmetic on the same registers - bit-identical h, rstd; the adaLN tensor of the norm has its own row stride (the final layer's is 2 d wide)."""
    B, L = 3, 37
    M = B * L
    g = lambda t: t.to(DEV) if t is not None else None
    x_in, br = rnd(M, d, seed=730), bf(rnd(M, d, seed=731, scale=1.5))
    w_b, w_n = 1 + 0.1 * rnd(d, seed=732), 1 + 0.1 * rnd(d, seed=733)
    mod_g, mod_n = _mod_inputs(B, d, 6, 734), _mod_inputs(B, d, 2, 735)
    modality = (torch.arange(M) % L >= L // 2).long() if img else None
    any_img = torch.ones(1, dtype=torch.int32) if img else None
    kw = dict(w_b=g(w_b), norm_type=nt, mod=g(mod_g), gate_idx=5, modality=g(modality), p_drop=0.1, seed=5)
    xo_a, rstd_a, _ = K.residual_fwd(g(x_in), g(br), L, **kw)
    h_a, rn_a, _ = K.norm_fwd(xo_a, g(w_n), nt, L, mod=g(mod_n), mod_idx=(0, 1), modality=g(modality), any_img=g(any_img))
    xo_f, rstd_f, _, (h_f, rn_f, _) = K.residual_fwd(g(x_in), g(br), L, next_w=g(w_n), next_mod=g(mod_n), next_mod_idx=(0, 1), next_modality=g(modality),
                                                   next_any_img=g(any_img), **kw)
    assert torch.equal(xo_f, xo_a) and torch.equal(rstd_f, rstd_a)
    assert torch.equal(h_f, h_a) and torch.equal(rn_f, rn_a)


@pytest.mark.parametrize("variant", ["mod_sandwich", "mod_img_gate_sandwich_dropout", "mod_gate_plain", "gate_only"])
@pytest.mark.parametrize("B,L", [(3, 37), (2, 700)], ids=["b3_l37", "b2_l700_more_rows_than_blocks"])
def test_norm_residual_bwd_ada_equals_separate_kernels(K, variant, B, L):
    """The fused adaLN pass (modulated norm backward + gated residual-branch backward, udm_norm_residual_bwd_ada) against the two separate kernels it replaces
    (each tested against torch above): same dx, dw, d branch, dw_b, and the same shift / scale / gate gradients up to the order of the fp32 sums."""
    d, nt = 2048, 0
    M = B * L
    g = lambda t: t.to(DEV) if t is not None else None
    x, w, dy = rnd(M, d, seed=720, scale=2.0) + 0.3, 1 + 0.1 * rnd(d, seed=721), bf(rnd(M, d, seed=722))
    br, dx0 = bf(rnd(M, d, seed=723, scale=1.5)), rnd(M, d, seed=724)
    mod_n, mod_r = _mod_inputs(B, d, 6, 725), _mod_inputs(B, d, 6, 726)
    use_mod = variant != "gate_only"
    use_gate = "gate" in variant
    sandwich = "sandwich" in variant
    img = "img" in variant
    p_drop = 0.1 if "dropout" in variant else 0.0
    modality = (torch.arange(M) % L >= L // 2).long() if img else None
    any_img = torch.ones(1, dtype=torch.int32) if img else None
    w_b = 1 + 0.1 * rnd(d, seed=727) if sandwich else None
    _, rstd, _ = K.norm_fwd(g(x), g(w), nt, L)
    _, rstd_b, _ = K.residual_fwd(g(rnd(M, d, seed=728)), g(br), L, w_b=g(w_b), norm_type=nt) if sandwich else (None, None, None)
    kw_r = dict(w_b=g(w_b), norm_type=nt, p_drop=p_drop, seed=11)
    # separate kernels
    dx_a, dw_a, dwb_a = g(dx0.clone()), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV) if sandwich else None
    dmn_a, dmr_a = torch.zeros(mod_n.shape, device=DEV), torch.zeros(mod_r.shape, device=DEV)
    K.norm_bwd(g(dy), g(x), rstd, None, g(w), nt, L, dx_a, dw_a, accumulate=True, mod=g(mod_n) if use_mod else None, dmod=dmn_a if use_mod else None, mod_idx=(3, 4),
               modality=g(modality), any_img=g(any_img))
    db_a = K.residual_bwd(dx_a, g(br), L, rstd=rstd_b, mean=None, mod=g(mod_r) if use_gate else None, dmod=dmr_a if use_gate else None, gate_idx=5 if use_gate else None,
                          modality=g(modality), dw_b=dwb_a, **kw_r)
    # fused
    dx_f, dw_f, dwb_f = g(dx0.clone()), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV) if sandwich else None
    dmn_f, dmr_f = torch.zeros(mod_n.shape, device=DEV), torch.zeros(mod_r.shape, device=DEV)
    db_f = K.norm_residual_bwd_ada(g(dy), g(x), rstd, None, g(w), nt, L, dx_f, dw_f, g(br), accumulate=True, w_b=g(w_b), rstd_b=rstd_b, dw_b=dwb_f, p_drop=p_drop, seed=11,
                                   mod_n=g(mod_n) if use_mod else None, dmod_n=dmn_f if use_mod else None, mod_idx=(3, 4), modality=g(modality), any_img=g(any_img),
                                   mod_r=g(mod_r) if use_gate else None, dmod_r=dmr_f if use_gate else None, gate_idx=5 if use_gate else None, modality_r=g(modality))
    assert rel_err(dx_f, dx_a) < 1e-5
    assert rel_err(db_f.float(), db_a.float()) < 1e-4      # (bf16 outputs: a block sum and a wave sum round a few elements differently)
    assert rel_err(dw_f, dw_a) < 1e-4
    if sandwich:
        assert rel_err(dwb_f, dwb_a) < 1e-4
    if use_mod:
        assert rel_err(dmn_f, dmn_a) < 1e-4
    if use_gate:
        assert rel_err(dmr_f, dmr_a) < 1e-4


@pytest.mark.parametrize("B,out,inp", [(8, 12288, 128), (16, 4096, 128), (5, 100, 32), (64, 4608, 128)], ids=["adaln_1p4b", "final_layer", "ragged_tiny", "batch64"])
def test_small_batch_linear_bwd(K, B, out, inp):
    """adaLN_modulation's backward in one launch (udm_small_batch_linear_bwd) against the fp32 statement on the bf16-rounded operands; dX and db accumulate,
    dW is overwritten; padded weight rows and leading dimensions."""
    dy = rnd(B, out + 8, seed=710, scale=0.5)[:, :out]
    x, w = bf(rnd(B, inp + 8, seed=711))[:, :inp], bf(rnd(out + 8, inp, seed=712, scale=0.3))
    dx0, db0 = rnd(B, inp, seed=713), rnd(out, seed=714)
    g = lambda t: t.to(DEV)
    dw, db, dx = torch.full((out, inp), 7.0, device=DEV), g(db0.clone()), g(dx0.clone())
    K.small_batch_linear_bwd(g(dy), g(x), g(w), dw, db, dx)
    d16 = dy.bfloat16().float()
    assert rel_err(dw.cpu(), d16.t() @ x.float()) < 1e-5
    assert rel_err(db.cpu(), db0 + d16.sum(0)) < 1e-5
    assert rel_err(dx.cpu(), dx0 + d16 @ w[:out].float()) < 1e-5
    parts = torch.full((K.small_batch_linear_bwd_tiles(out), B, inp), float("nan"), device=DEV)
    K.small_batch_linear_bwd(g(dy), g(x), g(w), dw, None, dx_parts=parts)      # no bias; the input gradient as partial tiles (every element written)
    assert rel_err(parts.sum(0).cpu(), d16 @ w[:out].float()) < 1e-5


@pytest.mark.parametrize("d", [64, 768, 2048])
@pytest.mark.parametrize("nt", [0, 1])
@pytest.mark.parametrize("mode", ["plain", "mod_all", "mod_img"])
@pytest.mark.parametrize("B,L", [(3, 40), (5, 37)], ids=["b3_l40", "b5_l37_ragged_row_chunks"])   # (modulated backward: a block owns a run of rows of ONE batch element)
def test_norm_fwd_bwd(K, d, nt, mode, B, L):
    M = B * L
    x, w, dy = rnd(M, d, seed=12, scale=2.0) + 0.3, 1 + 0.1 * rnd(d, seed=13), bf(rnd(M, d, seed=14))
    mod = _mod_inputs(B, d, 6, 15) if mode != "plain" else None
    modality = (torch.arange(M) % L >= L // 2).long() if mode == "mod_img" else None
    any_img = torch.ones(1, dtype=torch.int32) if mode == "mod_img" else None
    idx = (3, 4)
    yr, rstd_r, mean_r = R.norm_fwd(x, w, nt, L, mod=mod, mod_idx=idx, modality=modality, any_img=any_img)
    g = lambda t: t.to(DEV) if t is not None else None
    y, rstd, mean = K.norm_fwd(g(x), g(w), nt, L, mod=g(mod), mod_idx=idx, modality=g(modality), any_img=g(any_img))
    assert rel_err(y.float().cpu(), yr.float()) < 4e-3
    assert torch.allclose(rstd.cpu(), rstd_r, rtol=1e-5)
    dx0 = rnd(M, d, seed=16)
    dx_r, dw_r = dx0.clone(), torch.zeros(d)
    dmod_r = torch.zeros(mod.shape) if mod is not None else None
    R.norm_bwd(dy, x, rstd_r, mean_r, w, nt, L, dx_r, dw_r, accumulate=True, mod=mod, dmod=dmod_r, mod_idx=idx, modality=modality, any_img=any_img)
    dx, dw = g(dx0.clone()), torch.zeros(d, device=DEV)
    dmod = torch.zeros(mod.shape, device=DEV) if mod is not None else None
    K.norm_bwd(g(dy), g(x), rstd, mean, g(w), nt, L, dx, dw, accumulate=True, mod=g(mod), dmod=dmod, mod_idx=idx, modality=g(modality), any_img=g(any_img))
    assert rel_err(dx.cpu(), dx_r) < 1e-4
    assert rel_err(dw.cpu(), dw_r) < 1e-3
    if mod is not None:
        assert rel_err(dmod.cpu(), dmod_r) < 1e-3


@pytest.mark.parametrize("d", [64, 768, 2048])
@pytest.mark.parametrize("variant", ["plain", "sandwich_rms", "sandwich_ln", "gate_all", "gate_img", "gate_sandwich"])
@pytest.mark.parametrize("B,L", [(2, 24), (5, 37)], ids=["b2_l24", "b5_l37_ragged_row_chunks"])
def test_residual_fwd_bwd(K, d, variant, B, L):
    M = B * L
    x_in, br, dx = rnd(M, d, seed=17), bf(rnd(M, d, seed=18, scale=1.5)), rnd(M, d, seed=19)
    w_b = 1 + 0.1 * rnd(d, seed=20) if "sandwich" in variant else None
    nt = 1 if variant == "sandwich_ln" else 0
    mod = _mod_inputs(B, d, 6, 21) if variant.startswith("gate") else None
    gi = 5 if mod is not None else None
    modality = (torch.arange(M) % L >= L // 3).long() if variant == "gate_img" else None
    g = lambda t: t.to(DEV) if t is not None else None
    xr, rstd_r, mean_r = R.residual_fwd(x_in, br, L, w_b=w_b, norm_type=nt, mod=mod, gate_idx=gi, modality=modality)
    xo, rstd, mean = K.residual_fwd(g(x_in), g(br), L, w_b=g(w_b), norm_type=nt, mod=g(mod), gate_idx=gi, modality=g(modality))
    assert rel_err(xo.cpu(), xr) < 2e-3  # bf16 rounding of the normalised branch can flip an ulp
    dw_r = torch.zeros(d) if w_b is not None else None
    dmod_r = torch.zeros(mod.shape) if mod is not None else None
    db_r = R.residual_bwd(dx, br, L, w_b=w_b, rstd=rstd_r, mean=mean_r, norm_type=nt, mod=mod, dmod=dmod_r, gate_idx=gi, modality=modality, dw_b=dw_r)
    dw = torch.zeros(d, device=DEV) if w_b is not None else None
    dmod = torch.zeros(mod.shape, device=DEV) if mod is not None else None
    db = K.residual_bwd(g(dx), g(br), L, w_b=g(w_b), rstd=rstd, mean=mean, norm_type=nt, mod=g(mod), dmod=dmod, gate_idx=gi, modality=g(modality), dw_b=dw)
    assert rel_err(db.float().cpu(), db_r.float()) < 6e-3
    if w_b is not None:
        assert rel_err(dw.cpu(), dw_r) < 5e-3
    if mod is not None:
        assert rel_err(dmod.cpu(), dmod_r) < 5e-3


@pytest.mark.parametrize("d", [64, 768, 2048])
@pytest.mark.parametrize("variant", ["plain", "sandwich_rms", "sandwich_ln", "dropout"])
def test_residual_with_fused_next_norm_equals_separate_kernels(K, d, variant):
    """udm_residual_norm_fwd must be bit-identical to udm_residual_fwd followed by udm_norm_fwd (same arithmetic, row kept in registers)."""
    B, L = 2, 24
    M = B * L
    x_in, br = rnd(M, d, seed=71).to(DEV), bf(rnd(M, d, seed=72, scale=1.5)).to(DEV)
    w_b = (1 + 0.1 * rnd(d, seed=73)).to(DEV) if variant.startswith("sandwich") else None
    nt = 1 if variant == "sandwich_ln" else 0
    w_n = (1 + 0.1 * rnd(d, seed=74)).to(DEV)
    kw = dict(w_b=w_b, norm_type=nt, p_drop=0.2 if variant == "dropout" else 0.0, seed=99)
    x1, r1, m1 = K.residual_fwd(x_in, br, L, **kw)
    h1, rn1, mn1 = K.norm_fwd(x1, w_n, nt, L)
    x2, r2, m2, (h2, rn2, mn2) = K.residual_fwd(x_in, br, L, next_w=w_n, **kw)
    assert torch.equal(x1, x2) and torch.equal(h1, h2) and torch.equal(rn1, rn2)
    if w_b is not None:
        assert torch.equal(r1, r2)
    if nt == 1:
        assert torch.equal(mn1, mn2) and torch.equal(m1, m2)


@pytest.mark.parametrize("d", [2048, 4096, 768, 256, 1032])
@pytest.mark.parametrize("nt,sandwich,p,acc", [(0, True, 0.0, True), (0, True, 0.1, True), (1, True, 0.0, True), (0, False, 0.1, True), (0, True, 0.0, False), (1, False, 0.0, False)])
def test_norm_residual_bwd_fused_equals_the_two_kernels(K, d, nt, sandwich, p, acc):
    """The fused norm-backward + residual-branch-backward pass (one kernel per row pair of the block backward: block per row at d = 2048 / 4096,
    wave per row below) against the
    two kernels it replaces run back to back, and - through them - against the fp32 references those are tested with: dx, d branch, both
    weight gradients; rms / LayerNorm, with and without the sandwich norm, dropout mask regenerated from the same (seed, index)."""
    M, L = 1000, 250     # not a multiple of the grid: the row loop's tail
    x, dx0 = rnd(M, d, seed=400), rnd(M, d, seed=401)
    w, wb = 1 + 0.1 * rnd(d, seed=402), 1 + 0.1 * rnd(d, seed=403)
    dy, br = bf(rnd(M, d, seed=404)), bf(rnd(M, d, seed=405, scale=0.7))
    xg, wg, wbg, dyg, brg = x.to(DEV), w.to(DEV), wb.to(DEV), dy.to(DEV), br.to(DEV)
    _, rstd, mean = K.norm_fwd(xg, wg, nt, L)
    rb = mb = None
    if sandwich:
        _, rb, mb = K.residual_fwd(xg, brg, L, w_b=wbg, norm_type=nt)
    dxa, dwa, dwba = dx0.clone().to(DEV), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    K.norm_bwd(dyg, xg, rstd, mean, wg, nt, L, dxa, dwa, accumulate=acc)
    da = K.residual_bwd(dxa, brg, L, w_b=wbg if sandwich else None, rstd=rb, mean=mb, norm_type=nt, dw_b=dwba if sandwich else None, p_drop=p, seed=77)
    dxb, dwb_, dwbb = dx0.clone().to(DEV), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dbias = torch.zeros(d, device=DEV)
    db = K.norm_residual_bwd(dyg, xg, rstd, mean, wg, nt, L, dxb, dwb_, brg, accumulate=acc, w_b=wbg if sandwich else None, rstd_b=rb, mean_b=mb,
                             dw_b=dwbb if sandwich else None, p_drop=p, seed=77, dbias=dbias)
    torch.cuda.synchronize()
    assert torch.allclose(dbias.cpu(), db.float().cpu().sum(0), rtol=1e-4, atol=1e-3 * float(db.float().abs().max()) * M ** 0.5)   # fused bias gradient
    assert rel_err(dxb.cpu(), dxa.cpu()) < 1e-6
    assert rel_err(db.float().cpu(), da.float().cpu()) < 1e-3          # bf16 outputs: a few last-bit flips from the different summation order
    assert torch.equal((db == 0).cpu(), (da == 0).cpu())                # the same dropout mask
    assert rel_err(dwb_.cpu(), dwa.cpu()) < 1e-5
    if sandwich:
        assert rel_err(dwbb.cpu(), dwba.cpu()) < 1e-5


def test_residual_dropout_mask_consistent(K):
    M, d, L, p = 64, 256, 32, 0.25
    x_in, br, dx = torch.zeros(M, d), bf(torch.ones(M, d)), torch.ones(M, d)
    xo, _, _ = K.residual_fwd(x_in.to(DEV), br.to(DEV), L, p_drop=p, seed=1234)
    xo = xo.cpu()
    keep = xo != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    assert torch.allclose(xo[keep], torch.full_like(xo[keep], 1 / (1 - p)), rtol=1e-6)
    db = K.residual_bwd(dx.to(DEV), br.to(DEV), L, p_drop=p, seed=1234).float().cpu()
    assert torch.equal(db != 0, keep)
    xo2, _, _ = K.residual_fwd(x_in.to(DEV), br.to(DEV), L, p_drop=p, seed=1235)
    assert not torch.equal(xo2.cpu() != 0, keep)


def test_residual_dropout_mask_statistics(K):
    """16 random bits per decision: the keep rate is p rounded to 2^-16, neighbours (same Philox call, adjacent 16-bit halves) are independent."""
    M, d, L, p = 1024, 2048, 128, 0.1
    xo, _, _ = K.residual_fwd(torch.zeros(M, d, device=DEV), bf(torch.ones(M, d)).to(DEV), L, p_drop=p, seed=7)
    keep = (xo != 0).float().cpu()
    assert abs(keep.mean().item() - (1 - p)) < 1.5e-3          # 2.1 M draws: 3 sigma = 6e-4
    a, b = keep[:, 0::2].reshape(-1), keep[:, 1::2].reshape(-1)
    corr = ((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std())
    assert abs(corr.item()) < 5e-3
    col = keep.mean(0)
    assert (col - (1 - p)).abs().max().item() < 0.05           # no column is systematically kept / dropped (1024 draws each)


# ------------------------------------------------------------------------------------------------ qk-norm + rope
@pytest.mark.parametrize("d,D", [(64, 32), (768, 64), (2048, 128), (256, 64)])
@pytest.mark.parametrize("qk_norm", [True, False])
@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("q_scale", [1.0, "attention"])
def test_qknorm_rope(K, d, D, qk_norm, per_sample, q_scale):
    """q_scale = "attention": the engine's form - the stored q carries log2(e) / sqrt(D) (one rounding), the backward takes the gradient wrt that stored q"""
    q_scale = K.attention_q_scale(D) if q_scale == "attention" else 1.0
    B, L = 2, 20
    M = B * L
    qkv, dqkr = bf(rnd(M, 3 * d, seed=22)), bf(rnd(M, 2 * d, seed=23))
    if per_sample:
        ang = rnd(B, L, D // 2, seed=24)
    else:
        ang = rnd(L, D // 2, seed=24)
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    gq, bq, gk, bk = (1 + 0.1 * rnd(d, seed=25), 0.1 * rnd(d, seed=26), 1 + 0.1 * rnd(d, seed=27), 0.1 * rnd(d, seed=28)) if qk_norm else (None,) * 4
    g = lambda t: t.to(DEV) if t is not None else None
    ref, _ = R.qknorm_rope_fwd(qkv, cos, sin, L, D, gq=gq, bq=bq, gk=gk, bk=bk, q_scale=q_scale)
    out, stats = K.qknorm_rope_fwd(g(qkv), g(cos), g(sin), L, D, gq=g(gq), bq=g(bq), gk=g(gk), bk=g(bk), q_scale=q_scale)
    assert rel_err(out.float().cpu()[:, :d], ref.float()[:, :d]) < 6e-3 and rel_err(out.float().cpu()[:, d:], ref.float()[:, d:]) < 6e-3
    dqkv_r = torch.zeros(M, 3 * d, dtype=torch.bfloat16)
    grads_r = [torch.zeros(d) for _ in range(4)] if qk_norm else [None] * 4
    R.qknorm_rope_bwd(dqkr, qkv, dqkv_r, cos, sin, L, D, gq=gq, gk=gk, dgq=grads_r[0], dbq=grads_r[1], dgk=grads_r[2], dbk=grads_r[3], q_scale=q_scale)
    dqkv = torch.zeros(M, 3 * d, dtype=torch.bfloat16, device=DEV)
    grads = [torch.zeros(d, device=DEV) for _ in range(4)] if qk_norm else [None] * 4
    K.qknorm_rope_bwd(g(dqkr), g(qkv), dqkv, g(cos), g(sin), L, D, gq=g(gq), gk=g(gk), stats=stats, dgq=grads[0], dbq=grads[1], dgk=grads[2], dbk=grads[3], q_scale=q_scale)
    assert rel_err(dqkv.float().cpu()[:, :d], dqkv_r.float()[:, :d]) < 8e-3 and rel_err(dqkv.float().cpu()[:, d:2 * d], dqkv_r.float()[:, d:2 * d]) < 8e-3
    assert torch.all(dqkv.cpu()[:, 2 * d:] == 0)
    if qk_norm:
        for a, b in zip(grads, grads_r):
            assert rel_err(a.cpu(), b) < 5e-3


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, B, L, H, D, sid, do):
    q, k, v = (t.float().clone().requires_grad_() for t in (q, k, v))
    with torch.enable_grad():
        o = R._attn(q, k, v, B, L, H, D, sid)
        o.backward(do.float())
    return o.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("tr", [True, False])
@pytest.mark.parametrize("D,H", [(32, 2), (64, 3), (128, 2)])
@pytest.mark.parametrize("L", [32, 100, 384])
@pytest.mark.parametrize("use_sid", [False, True])
def test_attention_fwd_bwd(K, tr, D, H, L, use_sid):
    B = 2
    d = H * D
    M = B * L
    q, k, v, do = (bf(rnd(M, d, seed=s)) for s in (30, 31, 32, 33))
    sid = None
    if use_sid:
        sid = torch.zeros(B, L, dtype=torch.int64)
        sid[:, L // 3:] = 1
        sid[:, (2 * L) // 3:] = 2
        sid[1, -5:] = -1  # padding tail
    K.set_tr_read(tr)
    try:
        o_r, dq_r, dk_r, dv_r = _attn_ref(q, k, v, B, L, H, D, sid, do)
        g = lambda t: t.to(DEV) if t is not None else None
        o, lse = K.attention_fwd_generic(g(q), g(k), g(v), B, L, H, D, g(sid))
        assert rel_err(o.float().cpu(), o_r) < 1e-2
        dq, dk, dv = K.attention_bwd_generic(g(q), g(k), g(v), o, g(do), lse, B, L, H, D, g(sid))
        assert rel_err(dv.float().cpu(), dv_r) < 1.5e-2
        assert rel_err(dq.float().cpu(), dq_r) < 1.5e-2
        assert rel_err(dk.float().cpu(), dk_r) < 1.5e-2
        # LSE: natural-log LSE of scaled scores = lse2 * ln2
        if not use_sid:
            s = (q.float().reshape(B, L, H, D).transpose(1, 2) @ k.float().reshape(B, L, H, D).transpose(1, 2).transpose(-1, -2)) / math.sqrt(D)
            assert torch.allclose(lse.cpu() * math.log(2.0), torch.logsumexp(s, -1), atol=2e-2, rtol=1e-3)
    finally:
        K.set_tr_read(True)


@pytest.mark.parametrize("D,H,L,use_sid", [(32, 2, 100, False), (64, 3, 384, True), (128, 2, 384, False), (128, 8, 512, False), (128, 3, 200, True)])
def test_attention_fwd_bwd_with_prescaled_q(K, D, H, L, use_sid):
    """UDM_ATTN_Q_PRESCALED (the engine's form): q holds bf16(q log2(e) / sqrt(D)); O and LSE as for the plain call, dq is the gradient wrt the STORED q
    (= dq / q_scale), dk and dv unchanged.  (128, 8, 512) takes the persistent 64-queries-per-wave forward, the others the 8-wave kernels."""
    B = 2
    d, M = H * D, B * L
    qs = K.attention_q_scale(D)
    qf, k, v, do = (rnd(M, d, seed=s_) for s_ in (230, 231, 232, 233))
    q_st = bf(qf * qs)
    k, v, do = bf(k), bf(v), bf(do)
    sid = None
    if use_sid:
        sid = torch.zeros(B, L, dtype=torch.int64)
        sid[:, L // 2:] = 1
        sid[1, -7:] = -1
    o_r, dq_r, dk_r, dv_r = _attn_ref(q_st.float() / qs, k, v, B, L, H, D, sid, do)   # the function of the UNSCALED q the stored one stands for
    g = lambda t: t.to(DEV) if t is not None else None
    o, lse = K.attention_fwd_generic(g(q_st), g(k), g(v), B, L, H, D, g(sid), q_prescaled=True)
    assert rel_err(o.float().cpu(), o_r) < 1e-2
    dq, dk, dv = K.attention_bwd_generic(g(q_st), g(k), g(v), o, g(do), lse, B, L, H, D, g(sid), q_prescaled=True)
    assert rel_err(dv.float().cpu(), dv_r) < 1.5e-2
    assert rel_err(dk.float().cpu(), dk_r) < 1.5e-2
    assert rel_err(dq.float().cpu() * qs, dq_r) < 1.5e-2
    if not use_sid:
        s_ = ((q_st.float() / qs).reshape(B, L, H, D).transpose(1, 2) @ k.float().reshape(B, L, H, D).transpose(1, 2).transpose(-1, -2)) / math.sqrt(D)
        assert torch.allclose(lse.cpu() * math.log(2.0), torch.logsumexp(s_, -1), atol=2e-2, rtol=1e-3)


@pytest.mark.parametrize("L", [2, 31, 33, 64, 65, 127, 129, 191, 193, 257, 640, 1000])
@pytest.mark.parametrize("B,H", [(1, 1), (3, 5)])
def test_attention_d128_wave_specialised_backward_ragged_lengths(K, L, B, H):
    """D = 128 without a document mask takes the wave-specialised dK/dV kernel (32-query steps through a 6-stage ring, head / tail
    iterations, ragged last step, key blocks past L): every boundary length class, and B*H not a multiple of 8."""
    D = 128
    M, d = B * L, H * D
    q, k, v, do = (bf(rnd(M, d, seed=s)) for s in (130, 131, 132, 133))
    o_r, dq_r, dk_r, dv_r = _attn_ref(q, k, v, B, L, H, D, None, do)
    g = lambda t: t.to(DEV)
    o, lse = K.attention_fwd_generic(g(q), g(k), g(v), B, L, H, D, None)
    assert rel_err(o.float().cpu(), o_r) < 1e-2
    dq, dk, dv = K.attention_bwd_generic(g(q), g(k), g(v), o, g(do), lse, B, L, H, D, None)
    assert rel_err(dv.float().cpu(), dv_r) < 1.5e-2
    assert rel_err(dq.float().cpu(), dq_r) < 1.5e-2
    assert rel_err(dk.float().cpu(), dk_r) < 1.5e-2


def _doc_layouts(B, L):
    """Packed-sample id layouts: contiguous documents with a padding tail, one document, non-contiguous ids, a row of padding only, padding inside."""
    g = torch.Generator().manual_seed(L)
    out = {}
    sid = torch.zeros(B, L, dtype=torch.int64)
    cuts = sorted(torch.randint(1, L, (3,), generator=g).tolist())
    for i, c in enumerate(cuts):
        sid[:, c:] = i + 1
    sid[1 % B, -(L // 5 + 1):] = -1
    out["contiguous"] = sid
    out["single"] = torch.zeros(B, L, dtype=torch.int64)
    nc = (torch.arange(L) // 37 % 3)[None].repeat(B, 1)          # ids 0 1 2 0 1 2 ... : spans overlap, nothing may be skipped wrongly
    out["interleaved_ids"] = nc
    pad = sid.clone()
    pad[0] = -1                                                     # a whole row of padding
    pad[B - 1, L // 2:L // 2 + 70] = -1                             # padding in the middle of a row (whole 64-tile of padding when L is large)
    out["padding"] = pad
    if L >= 512:                                                    # documents that start and end on 128-row block boundaries, then a padding tail
        al = torch.full((B, L), -1, dtype=torch.int64)
        n = (L // 128) * 128
        al[:, :n] = (torch.arange(n) // 256)[None]
        al[B - 1, n - 128:n] = -1
        out["aligned"] = al
        a64 = torch.zeros(B, L, dtype=torch.int64)                 # documents that start / end on 64-row tiles but not on 128-row blocks, and one that ends off-tile
        for i, c in enumerate((192, 448, L - 100)):
            a64[:, c:] = i + 1
        out["aligned64"] = a64
    return out


@pytest.mark.parametrize("L", [100, 640, 1500])
def test_attention_doc_ranges_kernel_matches_definition(K, L):
    import fake_kernels

    for name, sid in _doc_layouts(3, L).items():
        r = K.attention_doc_ranges(sid.to(DEV)).cpu()
        assert torch.equal(r, fake_kernels.attention_doc_ranges(sid)), name


@pytest.mark.parametrize("D,H", [(64, 3), (128, 2)])
@pytest.mark.parametrize("L", [100, 640, 1500])
def test_attention_tile_skipping_is_exact(K, D, H, L):
    """With `doc_ranges` the kernels walk only the tiles that can hold a matching sample id; results must be BIT-identical to the unskipped run
    (the per-element id test stays) and match the fp32 reference."""
    B = 3
    M, d = B * L, H * D
    q, k, v, do = (bf(rnd(M, d, seed=s)) for s in (230, 231, 232, 233))
    g = lambda t: t.to(DEV)
    for name, sid in _doc_layouts(B, L).items():
        sd = sid.to(DEV)
        r = K.attention_doc_ranges(sd)
        o0, lse0 = K.attention_fwd_generic(g(q), g(k), g(v), B, L, H, D, sd)
        o1, lse1 = K.attention_fwd_generic(g(q), g(k), g(v), B, L, H, D, sd, r)
        assert torch.equal(o0, o1) and torch.equal(lse0, lse1), name
        g0 = K.attention_bwd_generic(g(q), g(k), g(v), o0, g(do), lse0, B, L, H, D, sd)
        g1 = K.attention_bwd_generic(g(q), g(k), g(v), o1, g(do), lse1, B, L, H, D, sd, r)
        for a, b_, nm in zip(g0, g1, "qkv"):
            if D == 128 and nm != "q":   # document-pure key blocks take the wave-specialised dK/dV kernel: same math, another summation order
                assert rel_err(a.float(), b_.float()) < 4e-3, (name, nm)
            else:
                assert torch.equal(a, b_), (name, nm)
        o_r, dq_r, dk_r, dv_r = _attn_ref(q, k, v, B, L, H, D, sid, do)
        assert rel_err(o1.float().cpu(), o_r) < 1e-2, name
        for a, b_, nm in zip(g1, (dq_r, dk_r, dv_r), "qkv"):
            assert rel_err(a.float().cpu(), b_) < 1.5e-2, (name, nm)
        pad_rows = (sid.reshape(-1) < 0)
        assert (o1.float().cpu()[pad_rows] == 0).all() and all((t.float().cpu()[pad_rows] == 0).all() for t in g1), name   # padding rows: zeros


# ------------------------------------------------------------------------------------------------ embedding / CE / small ops
def test_embedding_fwd_bwd(K):
    V, d, M, hot = 97, 192, 512, 40
    E, Em = rnd(V, d, seed=50), rnd(2, d, seed=51)
    ids = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(1))
    ids[::2] = hot
    mod = (torch.arange(M) % 7 > 2).long()
    x = K.embedding_fwd(ids.to(DEV), E.to(DEV), mod.to(DEV), Em.to(DEV)).cpu()
    assert torch.equal(x, R.embedding_fwd(ids, E, mod, Em))
    dx = rnd(M, d, seed=52)
    dE, dEm = torch.zeros(V, d, device=DEV), torch.zeros(2, d, device=DEV)
    K.embedding_bwd(ids.to(DEV), dx.to(DEV), dE, hot, modality=mod.to(DEV), dEm=dEm)
    dE_r, dEm_r = torch.zeros(V, d), torch.zeros(2, d)
    R.embedding_bwd(ids, dx, dE_r, hot, modality=mod, dEm=dEm_r)
    assert torch.allclose(dE.cpu(), dE_r, atol=1e-4, rtol=1e-5) and torch.allclose(dEm.cpu(), dEm_r, atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("V,Vt", [(65, 41), (1001, 1001), (40193, 32001)])
@pytest.mark.parametrize("restrict", [True, False])
def test_subs_ce(K, V, Vt, restrict):
    if restrict and V == Vt:
        pytest.skip("text-only vocabulary has nothing to restrict")
    M = 48
    mask_id = Vt - 1
    Vp = (V + 127) // 128 * 128
    logits = torch.zeros(M, Vp, dtype=torch.bfloat16)
    logits[:, :V] = bf(rnd(M, V, seed=60, scale=2.0))
    modality = (torch.arange(M) % 4 >= 2).long() if V > Vt else torch.zeros(M, dtype=torch.long)
    gen = torch.Generator().manual_seed(3)
    x0 = torch.where(modality == 1, torch.randint(Vt, max(V, Vt + 1), (M,), generator=gen), torch.randint(0, Vt - 1, (M,), generator=gen))
    xt = x0.clone()
    xt[::2] = mask_id
    g_up = rnd(M, seed=61)
    lp_r, lse_r = R.subs_ce_fwd(logits, x0, xt, modality, V, Vt, mask_id, restrict)
    gl = logits.clone().to(DEV)
    lp, lse = K.subs_ce_fwd(gl, x0.to(DEV), xt.to(DEV), modality.to(DEV), V, Vt, mask_id, restrict)
    assert torch.allclose(lp.cpu(), lp_r, atol=2e-4, rtol=1e-5)
    assert torch.all(lp.cpu()[1::2] == 0)
    full = K.subs_logprobs(gl, xt.to(DEV), modality.to(DEV), V, Vt, mask_id, restrict, out_dtype=torch.float32).cpu()
    full_r = R.subs_logprobs(logits, xt, modality, V, Vt, mask_id, restrict, out_dtype=torch.float32)
    assert torch.allclose(full, full_r, atol=2e-4, rtol=1e-5)
    assert torch.allclose(full.gather(1, x0[:, None])[:, 0], lp.cpu(), atol=2e-4)
    ref = logits.clone()
    R.subs_ce_bwd(ref, x0, xt, modality, lse_r, g_up, V, Vt, mask_id, restrict)
    K.subs_ce_bwd(gl, x0.to(DEV), xt.to(DEV), modality.to(DEV), lse, g_up.to(DEV), V, Vt, mask_id, restrict)
    assert rel_err(gl.float().cpu(), ref.float()) < 6e-3
    assert torch.all(gl.cpu()[:, V:] == 0) and torch.all(gl.cpu()[1::2] == 0)
    if restrict:   # narrow form (head per modality): rows [0, n) are the text GROUP, the rest the image group - only the group's columns are written
        n = 20
        sentinel = torch.full_like(logits, 7.0)
        sentinel[:, :V] = logits[:, :V]
        gn = sentinel.clone().to(DEV)
        K.subs_ce_bwd(gn, x0.to(DEV), xt.to(DEV), modality.to(DEV), lse, g_up.to(DEV), V, Vt, mask_id, restrict, narrow_txt_rows=n)
        gn, full_b = gn.cpu(), gl.cpu()
        hi_t, lo_i = min((Vt + 63) // 64 * 64, Vp), Vt // 8 * 8
        assert torch.equal(gn[:n, :hi_t], full_b[:n, :hi_t]) and torch.equal(gn[n:, lo_i:], full_b[n:, lo_i:])       # what it writes equals the whole-row form
        assert torch.equal(gn[:n, hi_t:], sentinel[:n, hi_t:]) and torch.equal(gn[n:, :lo_i], sentinel[n:, :lo_i])   # ... and nothing else is touched


def test_small_ops(K):
    B, Bp = 5, 8
    sigma = torch.rand(B, generator=torch.Generator().manual_seed(5)) * 3
    te = torch.zeros(Bp, 256, dtype=torch.bfloat16, device=DEV)
    K.timestep_embedding(sigma.to(DEV), te, B, 256)
    ref = torch.zeros(Bp, 256, dtype=torch.bfloat16)
    R.timestep_embedding(sigma, ref, B, 256)
    assert torch.allclose(te.float().cpu(), ref.float(), atol=1e-2)
    x, dy = bf(rnd(Bp, 128, seed=70)), bf(rnd(Bp, 128, seed=71))
    assert rel_err(K.silu_fwd(x.to(DEV)).float().cpu(), R.silu_fwd(x).float()) < 4e-3
    assert rel_err(K.silu_bwd(x.to(DEV), dy.to(DEV)).float().cpu(), R.silu_bwd(x, dy).float()) < 6e-3


def test_error_reporting(K):
    a = torch.zeros(8, 12, dtype=torch.bfloat16, device=DEV)  # K = 12 is not a multiple of 8
    with pytest.raises(RuntimeError, match="multiples of 8"):
        K.gemm_nt(a, a)
    with pytest.raises(RuntimeError, match="GPU tensors"):
        K.gemm_nt(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))


@pytest.mark.parametrize("case", ["weighted", "weighted_cap", "weighted_no_image", "mean", "mean_full_mask", "mean_no_modality", "all_padding"])
def test_diffusion_loss_kernel_matches_the_tensor_statements(K, case):
    """udm_diffusion_loss (the loss arithmetic of compute_loss in one launch) against the reference's own sequence of tensor statements (fake_kernels.diffusion_loss,
    gradient through autograd): value, text / image terms, fractions, per-token NLLs and d loss / d log_p, including the NaN -> 0 conventions of an empty modality."""
    import fake_kernels

    B, L = 6, 200
    g = torch.Generator().manual_seed(11)
    log_p = -torch.rand(B, L, generator=g) * 9
    sigma = torch.rand(B, generator=g) * 3 + 0.05
    dsigma = 1 + torch.rand(B, generator=g)
    w_std, w_loss = dsigma / torch.expm1(sigma), dsigma / (torch.expm1(sigma) + 0.2)
    att = torch.rand(B, L, generator=g) < 0.9
    is_img = torch.zeros(B, L, dtype=torch.bool)
    is_img[:, 72:] = True
    kw = dict(weighted=case.startswith("weighted"), text_w=0.7, img_w=1.3)
    if case == "weighted_cap":
        kw["ratio"] = 0.25
    if case == "weighted_no_image":
        is_img[:] = False
    if case == "mean_full_mask":
        kw["full_mask"] = True
    if case == "all_padding":
        att[:] = False
        kw["weighted"] = False
    mm = None if case == "mean_no_modality" else torch.stack([~is_img, is_img], -1)
    n_r, c_r, s_r = fake_kernels.diffusion_loss(log_p, w_loss, w_std, att, mm, **kw)
    dev = lambda t: t.to(DEV) if t is not None else None
    n, c, s = K.diffusion_loss(dev(log_p), dev(w_loss), dev(w_std), dev(att), dev(mm), **kw)
    n, c, s = n.cpu(), c.cpu(), s.cpu()
    assert torch.allclose(n, n_r, rtol=1e-6, atol=0)
    used = [0, 5] + ([1, 2] if kw["weighted"] else []) + ([3, 4, 6, 7] if mm is not None and case != "all_padding" else [])
    assert torch.allclose(s[used], s_r[used], rtol=2e-6, atol=1e-7), (s, s_r)
    assert torch.isfinite(s[0]) and torch.isfinite(c).all()
    if case == "weighted_no_image":
        # the empty modality contributes 0 (nan_to_num) and the other one is untouched; the tensor statements propagate 0 x inf = NaN into EVERY token's gradient
        # here (the backward of a sum divided by a zero count) - the kernel deliberately returns the finite gradient of the non-empty term instead
        assert float(s[2]) == 0.0 and float(s[1]) > 0.0 and torch.isnan(c_r).all()
        n_txt = float((att & ~is_img).sum())
        assert torch.allclose(c, -(w_loss[:, None] * att) * (0.7 / n_txt), rtol=2e-6, atol=0)
    elif case == "all_padding":
        assert float(s[0]) == 0.0 and float(c.abs().max()) == 0.0 and torch.isnan(c_r).all()   # (same remark: 0 / 0 -> loss 0, gradient 0 instead of NaN)
    else:
        assert torch.allclose(c, c_r, rtol=2e-6, atol=1e-10)


@pytest.mark.parametrize("draws", ["none", "txt_only", "both"])
def test_qxt_absorbing_kernel_matches_the_tensor_statements(K, draws):
    """udm_qxt_absorbing (q_xt after its random draws, one launch) against the reference's statements: bit-identical xt, move_indices and per-row flags,
    including rows drawn for both modalities (which mask neither) and draws that sit exactly on the fp32-rounded threshold."""
    import fake_kernels

    B, L, mask_id = 64, 333, 4242
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 4000, (B, L), generator=g)
    r_move = torch.rand(B, L, generator=g)
    mc = torch.rand(B, 1, generator=g)
    r_move[3, :7] = mc[3]                      # equality: `<` must stay strict
    mm = torch.zeros(B, L, 2, dtype=torch.bool)
    mm[:, :100, 0] = True
    mm[:, 100:, 1] = True
    p = 0.3
    r_txt = r_img = None
    kw = {}
    if draws != "none":
        r_txt = torch.rand(B, 1, generator=g)
        r_txt[0] = torch.tensor(p / 2 if draws == "both" else p, dtype=torch.float32)   # exactly the threshold as the comparison sees it: not masked
        kw = dict(r_txt=r_txt, p_txt=p if draws == "txt_only" else p / 2, modality_mask=mm)
        if draws == "both":
            r_img = torch.rand(B, 1, generator=g)
            r_txt[1], r_img[1] = 0.0, 0.0                                                  # drawn for both: neither
            kw.update(r_img=r_img, p_img=p / 2)
    ref = fake_kernels.qxt_absorbing(x, r_move, mc, mask_id, **kw)
    dev = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    out = K.qxt_absorbing(x.to(DEV), r_move.to(DEV), mc.to(DEV), mask_id, **dev)
    for a, b in zip(out, ref):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.cpu(), b)
    if draws == "both":
        assert not bool(out[2][1]) and not bool(out[3][1]) and not bool(out[4][1])


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 12, 64, 100, 1000, 4099])
@pytest.mark.parametrize("antithetic", [True, False])
def test_sample_t_noise_kernel_is_bit_identical_to_the_tensor_statements(K, n, antithetic):
    """udm_sample_t_noise against the reference's statements RUN ON THE DEVICE (the product's previous path): t, sigma, dsigma and the move chance bit for bit -
    t and the move chance feed exact mask comparisons.  Batch sizes that are not powers of two exercise the divide-as-multiply-by-reciprocal rounding."""
    import fake_kernels

    g = torch.Generator().manual_seed(n)
    for rep in range(20):
        u = torch.rand(n, generator=g).to(DEV)
        if rep == 0:
            u[0] = 0.0
        ref = fake_kernels.sample_t_noise(u, antithetic=antithetic, sampling_eps=1e-3, noise_eps=1e-3)
        out = K.sample_t_noise(u, antithetic=antithetic, sampling_eps=1e-3, noise_eps=1e-3)
        for name, a, b in zip(("t", "sigma", "dsigma", "move_chance"), out, ref):
            assert torch.equal(a, b), (name, n, rep, (a != b).nonzero().flatten()[:4], a[a != b][:4], b[a != b][:4])
