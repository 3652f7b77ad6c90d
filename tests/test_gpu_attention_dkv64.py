"""The generated one-wave-per-SIMD attention backward - the dQ pass (csrc/attention_dq64.hip, asmgen/attn_dq64.py: 64 queries per wave, + delta and the planes) and
the dK / dV pass (csrc/attention_dkv64.hip, asmgen/attn_dkv64.py: 64 keys per wave) - against a torch fp32 attention backward and against the 8-wave kernels of
attention.hip / attention_dkv_ws.hip on the same inputs: separate buffers and the engine's strided layout, one to five blocks per persistent
workgroup, the balanced walk (half blocks), more blocks than CUs x 2, shapes the kernel does not take (B*H = 6, one head, L = 256: fall back).
Replaces the dK / dV half of the backward of flash_attn_qkvpacked_func, models/dit.py:843."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def K():
    from unidisc_amd import kernels as K
    return K


def _ref(q, k, v, do, B, L, H, D):
    """q is PRE-SCALED by log2(e) / sqrt(D): the scores are base-2 exponents.  Returns dq (wrt the stored q), dk, dv in fp32 (autograd on the fp32 attention)."""
    q, k, v = (t.float().clone().requires_grad_() for t in (q, k, v))
    qh, kh, vh = (t.reshape(B, L, H, D).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * math.log(2.0)
    o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * L, H * D)
    o.backward(do.float())
    return q.grad, k.grad, v.grad


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("B,H,L", [(1, 8, 512), (2, 4, 768), (1, 8, 1024), (3, 8, 1280), (8, 16, 1280), (1, 16, 2048), (2, 24, 512), (2, 3, 512), (8, 1, 512), (1, 8, 256),
                                   (1, 24, 4096), (4, 8, 512)])     # 384 blocks of 16 tiles: whole rounds + halves with another tile count; 64 blocks: fewer than CUs
def test_dkv64_matches_reference_and_8wave_kernel(K, B, H, L):
    D, dev = 128, "cuda"
    g = torch.Generator(device=dev).manual_seed(B * 1000 + L)
    q, k, v, do = ((1.2 * torch.randn(B * L, H * D, device=dev, generator=g)) for _ in range(4))
    q, k, v, do = (q * K.attention_q_scale(D)).to(BF16), k.to(BF16), v.to(BF16), do.to(BF16)
    o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D, q_prescaled=True)
    K.set_attention_dkv64(True)
    K.set_attention_dq64(True)
    dq, dk, dv = K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D, q_prescaled=True)
    K.set_attention_dkv64(False)
    K.set_attention_dq64(False)
    try:
        dq8, dk8, dv8 = K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D, q_prescaled=True)
        K.set_attention_dq64(True)         # the generated dQ pass feeding the 8-wave dK / dV kernel through the planes it leaves behind
        dqm, dkm, dvm = K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D, q_prescaled=True)
    finally:
        K.set_attention_dkv64(True)
        K.set_attention_dq64(True)
    dq_r, dk_r, dv_r = _ref(q, k, v, do, B, L, H, D)
    assert torch.isfinite(dk.float()).all() and torch.isfinite(dv.float()).all() and torch.isfinite(dq.float()).all()
    eq, ek, ev = _rel(dq.float(), dq_r), _rel(dk.float(), dk_r), _rel(dv.float(), dv_r)
    eq8, ek8, ev8 = _rel(dq8.float(), dq_r), _rel(dk8.float(), dk_r), _rel(dv8.float(), dv_r)
    # bf16 output rounding is ~2.3e-3, P / dS are rounded to bf16 in every kernel; no worse than the kernels they replace
    assert eq < 8e-3 and ek < 8e-3 and ev < 8e-3 and eq < 1.15 * eq8 + 1e-4 and ek < 1.15 * ek8 + 1e-4 and ev < 1.15 * ev8 + 1e-4, (eq, eq8, ek, ek8, ev, ev8)
    assert _rel(dq.float(), dq8.float()) < 5e-3 and _rel(dk.float(), dk8.float()) < 5e-3 and _rel(dv.float(), dv8.float()) < 5e-3
    assert torch.equal(dqm, dq) and _rel(dkm.float(), dk8.float()) < 5e-3 and _rel(dvm.float(), dv8.float()) < 5e-3


def test_dkv64_engine_layout_and_untouched_neighbours(K):
    """q | k in one [M, 2d] buffer, v at column 2d of [M, 3d]; dK into columns d..2d of [M, 2d], dV into columns 2d..3d of [M, 3d] (how the DiT block calls it):
    the neighbouring columns (dq's, and the q / k gradient slots of the [M, 3d] buffer) keep what they held."""
    B, H, L, D, dev = 2, 4, 1280, 128, "cuda"
    d, M = H * D, B * L
    g = torch.Generator(device=dev).manual_seed(7)
    qkr = torch.randn(M, 2 * d, device=dev, generator=g)
    qkr[:, :d] *= K.attention_q_scale(D)
    qkr = qkr.to(BF16)
    qkv = torch.randn(M, 3 * d, device=dev, generator=g).to(BF16)
    do = torch.randn(M, d, device=dev, generator=g).to(BF16)
    o, lse = K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True)
    outs = []
    for flag in (True, False):
        dqkr = torch.full_like(qkr, 7.0)
        dqkv = torch.full_like(qkv, 7.0)
        K.set_attention_dkv64(flag)
        K.set_attention_dq64(flag)
        try:
            K.attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, q_prescaled=True)
        finally:
            K.set_attention_dkv64(True)
            K.set_attention_dq64(True)
        outs.append((dqkr, dqkv))
    (a, av), (b, bv) = outs
    assert torch.all(av[:, :2 * d] == 7.0) and torch.all(bv[:, :2 * d] == 7.0)
    dq_r, dk_r, dv_r = _ref(qkr[:, :d], qkr[:, d:], qkv[:, 2 * d:], do, B, L, H, D)
    assert _rel(a[:, d:].float(), dk_r) < 8e-3 and _rel(av[:, 2 * d:].float(), dv_r) < 8e-3
    assert _rel(a[:, d:].float(), b[:, d:].float()) < 5e-3 and _rel(av[:, 2 * d:].float(), bv[:, 2 * d:].float()) < 5e-3
    assert _rel(a[:, :d].float(), dq_r) < 8e-3 and _rel(a[:, :d].float(), b[:, :d].float()) < 5e-3


def test_generated_attention_programs_respect_the_cu_plan(K):
    """While a collective holds CUs (`udm_gemm_set_cus`, the data-parallel schedule `overlap_planned`) the persistent grids of the forward and of both backward passes shrink
    to the CUs that are left (224 here: 640 blocks no longer split into whole rounds + halves) - same blocks, same arithmetic."""
    B, H, L, D, dev = 8, 16, 1280, 128, "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    q, k, v, do = ((1.2 * torch.randn(B * L, H * D, device=dev, generator=g)) for _ in range(4))
    q, k, v, do = (q * K.attention_q_scale(D)).to(BF16), k.to(BF16), v.to(BF16), do.to(BF16)
    o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D, q_prescaled=True)
    outs = []
    for cus in (0, 224):
        K.gemm_set_cus(cus)
        try:
            o_c, lse_c = K.attention_fwd_generic(q, k, v, B, L, H, D, q_prescaled=True)
            outs.append((o_c, lse_c) + K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D, q_prescaled=True))
        finally:
            K.gemm_set_cus(0)
    (o0, l0, *g0), (o1, l1, *g1) = outs
    # the forward's half blocks (grid 256) and whole blocks (grid 224) take their lazy-rescale decisions per wave of 32 / 64 queries: equal to bf16 rounding, not bit for bit
    assert _rel(o1.float(), o0.float()) < 2.5e-3 and (l1 - l0).abs().max() < 1e-4
    for a, b in zip(g0, g1):      # the backward passes have no data-dependent decisions: the same o / lse in, the same bits out
        assert torch.equal(a, b)
