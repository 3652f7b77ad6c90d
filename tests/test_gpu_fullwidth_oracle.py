"""HIP path vs the CPU oracle at BASELINE.json's REAL widths (pytest -m gpu), same seeded batch, same generator stream.

The goldens (tests/golden/*.npz) pin the oracle to the imported reference at d = 64; this file composes the kernels that only exist at full
width — block-per-row norm kernels (d >= 2048), wave-specialised dK/dV (D = 128), 320-row / persistent GEMM tiles (M = 10 240), split-K head
dgrad on compacted rows, the 48 512-padded vocabulary head — against `oracle.compute_loss` (reference: model.py:797-1173, models/dit.py:948-1033)
on one or two blocks, which the oracle finishes in well under a minute.

Every comparison is recorded in the parity ledger (tests/ledger.py -> profiles/r05_parity_ledger.json) and asserted at <= 3x the error
achieved there.  Masks (xt, move_indices, token_mask) and t are bit-exact.  Two comparators are reported for floating point:
  * `fp32`: the oracle in fp32 (truth);
  * `bf16`: the oracle with the reference's autocast rounding points emulated — the reference's own bf16 numerics, which is what north_star's
    "within 1e-3 rel on bf16 logits / loss" refers to.
"""
import os

import pytest
import torch

from ledger import check, record
from oracle import unidisc_oracle as O
from oracle.cases import lumina_rope_2d
from product_utils import product_config

pytestmark = pytest.mark.gpu
DEV = "cuda"

_LARGE = dict(hidden_size=2048, n_heads=16, cond_dim=128, txt_length=256, img_length=1024, text_vocab_size=32001, vocab_size=32001 + 16384,
              norm_type="rms", qk_norm=True, sandwich_normalization=True, modality_embed=True, rope_2d=True, linear_factor=2.0, time_conditioning=False,
              multimodal_batches=True, force_argmax_valid_indices=True, mask_entire_modality=0.1, softmin_snr=5, text_loss_weight=1.0, img_loss_weight=0.5,
              force_full_attention_mask=True)
_SMALL = dict(hidden_size=768, n_heads=12, cond_dim=128, txt_length=128, img_length=256, text_vocab_size=32001, vocab_size=32001 + 8192,
              norm_type="rms", qk_norm=True, sandwich_normalization=True, modality_embed=True, rope_2d=False, time_conditioning=False,
              multimodal_batches=True, force_argmax_valid_indices=True, mask_entire_modality=0.1, softmin_snr=5, text_loss_weight=1.0, img_loss_weight=None,
              force_full_attention_mask_loss_only=True)
_PLUMB = dict(hidden_size=256, n_heads=4, cond_dim=128, txt_length=128, img_length=0, text_vocab_size=1001, vocab_size=1001, norm_type="layernorm",
              qk_norm=False, sandwich_normalization=False, modality_embed=False, rope_2d=False, time_conditioning=True, multimodal_batches=False,
              force_argmax_valid_indices=False)

# name -> (case, batch size, asserted bounds).  Bounds: loss relative error vs the fp32 oracle; per-token NLL rel-RMS; worst / median
# per-parameter gradient rel-RMS vs the fp32 oracle.  Numbers are <= 3x the errors recorded in profiles/r05_parity_ledger.json, except the loss:
# achieved 5e-7 .. 4e-6 (a mean over thousands of tokens), asserted at 5e-5 - twenty times inside north_star's 1e-3.  The worst gradient is
# always a 768- / 2048-element qk-norm vector deep in the stack (bf16 noise of every layer above it); the reference's own bf16 run (oracle with
# its rounding points emulated) is recorded next to it as the floor.
FULLWIDTH = {
    # BASELINE configs[2] at its exact per-GPU shape M = 8 x 1280 = 10 240 rows (the bench's GEMM tiles), one block
    "config_c_1block_b8": (dict(_LARGE, n_blocks=1), 8, dict(loss=5e-5, nll=1e-3, grad_max=3.2e-2, grad_med=2e-2)),
    # the same width, two blocks composed (block -> block fused residual+norm), B = 2
    "config_c_2blocks_b2": (dict(_LARGE, n_blocks=2), 2, dict(loss=5e-5, nll=1e-3, grad_max=1.1e-1, grad_med=2.2e-2)),
    # north_star's adaLN-Zero variant AT THE WIDTH ITS FUSED KERNELS EXIST FOR (round 6): `time_conditioning=True` on the same 1.4 B block
    # (rms + qk-norm + sandwich + modality embed + multimodal_batches: image-only modulate / gate, gate_msa unused - models/dit.py:922-925, 966-984, 1015-1022,
    # 1078-1091).  d = 2048 is where `udm_norm_residual_bwd_ada`, the deferred adaLN_modulation backward through `_ada_buf` and the "next block's adaLN one
    # block early" forward dispatch (unidisc_amd/dit.py `tc_fused`); adaLN weights are randomised below (the reference zero-initialises them), and every
    # `sigma_map.*` / `adaLN_modulation.*` gradient is asserted by name
    "config_c_adaln_1block_b8": (dict(_LARGE, n_blocks=1, time_conditioning=True), 8, dict(loss=5e-5, nll=1e-3, grad_max=3.2e-2, grad_med=2e-2)),
    "config_c_adaln_2blocks_b2": (dict(_LARGE, n_blocks=2, time_conditioning=True), 2, dict(loss=5e-5, nll=1e-3, grad_max=1.1e-1, grad_med=2.2e-2)),
    # BASELINE configs[2] at its FULL DEPTH: all 24 blocks, d = 2048, L = 1280, two sequences (the fp32 oracle's fwd+bwd takes ~23 s per sequence on the GPU
    # box's host).  Achieved (profiles/r05_parity_ledger.json): loss 3.5e-6 / 5.5e-6, NLL 3.7e-4 / 3.9e-4, median gradient 8.9e-3 / 9.2e-3; the worst parameter (a
    # qk-norm vector, 1.06e-1 / 1.55e-1) is additionally held to 1.5x the reference's own floor with the flash-attention rounding points (below)
    # grad_max (round 4): <= 2x the recorded worst parameter (1.55e-1 / 1.05e-1 in profiles/r05_parity_ledger.json), on top of the 1.5x-of-floor assertion
    # grad_max = None (round 5): at this depth the worst parameter's absolute error (0.10 - 0.16) says nothing - an absolute bound of 0.25 would
    # hide a regression of 60 % - the assertions that bind are the ratios to the reference's own flash-rounding floor below (worst <= 1.5x, every parameter <= 2x)
    # (`config_c_24blocks_b1` ran here until round 5: the two-sequence case below subsumes it, and the suite has a time budget - VERDICT r5 item 8)
    "config_c_24blocks_b2": (dict(_LARGE, n_blocks=24), 2, dict(loss=5e-5, nll=1.2e-3, grad_max=None, grad_med=2.7e-2)),
    # BASELINE configs[1]: UniDisc-S, all 12 blocks, L = 128 + 256
    "unidisc_s_12blocks_b4": (dict(_SMALL, n_blocks=12), 4, dict(loss=5e-5, nll=1.2e-3, grad_max=None, grad_med=3e-2)),
    # BASELINE configs[0]: 2-layer d = 256 text-only adaLN DiT, L = 128, vocabulary 1k (+ [MASK])
    "config_a_plumbing_b8": (dict(_PLUMB, n_blocks=2), 8, dict(loss=5e-5, nll=1e-3, grad_max=2.5e-2, grad_med=2e-2)),
    # the same LayerNorm DiT WITHOUT time conditioning and without sandwich norms: the residual adds carry the fused next pre-norm in its LayerNorm form (round 5: the
    # attention branch's add passed no norm type and normalised as RMS - no golden case has this combination)
    "layernorm_no_adaln_b4": (dict(_PLUMB, n_blocks=2, time_conditioning=False), 4, dict(loss=5e-5, nll=1e-3, grad_max=2.5e-2, grad_med=2e-2)),
}


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _make_batch(case, B, gen):
    Lt, Li, Vt = case["txt_length"], case["img_length"], case["text_vocab_size"]
    if Li == 0:
        am = torch.ones(B, Lt, dtype=torch.bool)
        am[1, Lt - 17:] = False   # ragged text: one padded row
        return dict(input_ids=torch.randint(0, Vt - 1, (B, Lt), generator=gen), attention_mask=am)
    return dict(txt_input_ids=torch.randint(0, Vt - 1, (B, Lt), generator=gen, dtype=torch.int32),
                img_input_ids=torch.randint(0, case["vocab_size"] - Vt, (B, Li), generator=gen, dtype=torch.int32).to(torch.int16),
                txt_attention_mask=torch.ones(B, Lt, dtype=torch.bool))


@pytest.mark.parametrize("name", sorted(FULLWIDTH))
def test_training_step_matches_oracle_at_full_width(name):
    case, B, bound = FULLWIDTH[name]
    from unidisc_amd import Diffusion

    cfg = product_config(case)
    torch.manual_seed(0)
    diff = Diffusion(cfg, None, DEV)
    diff.backbone.train()
    diff.rng_device = "cpu"
    wg = torch.Generator().manual_seed(5)
    with torch.no_grad():   # non-trivial head / adaLN weights (the reference zero-initialises them)
        for n, p in sorted(diff.backbone.named_parameters()):
            if n.endswith("linear.weight") or "adaLN_modulation" in n:
                p.copy_((torch.randn(p.shape, generator=wg) * (0.5 / p.shape[-1] ** 0.5)).to(DEV))
    assert diff.vocab_size == case["vocab_size"] and diff.mask_index == case["text_vocab_size"] - 1
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in diff.backbone.named_parameters()}
    batch = _make_batch(case, B, torch.Generator().manual_seed(77))

    ocfg = O.OracleConfig.from_case(case)
    bufs = O.make_buffers(ocfg, lumina_rope_2d)
    ob = O.update_batch(ocfg, {k: v.clone() for k, v in batch.items()})
    o32 = O.compute_loss(ocfg, P, bufs, ob, torch.Generator().manual_seed(123))
    o32.loss.backward()
    # the reference's own bf16 numerics = the noise floor: ONE emulated pass (round 6; two before) with the autocast rounding points AND the rounding points INSIDE
    # the reference's flash-attention backward (P, dS in bf16; delta from the stored bf16 O) - tensor-boundary rounding alone runs that backward in fp32 and
    # understates the noise of dq / dk, i.e. of exactly the qk-norm vectors that are the worst parameters here (round 2 finding)
    P16f = {k: v.detach().clone().requires_grad_() for k, v in P.items()}
    o16 = O.compute_loss(ocfg, P16f, bufs, ob, torch.Generator().manual_seed(123), bf16=True, flash_rounding=True)
    o16.loss.backward()

    torch.manual_seed(123)
    out = diff.training_step({k: v.clone() for k, v in batch.items()}, 1)
    # integer / boolean quantities: bit-exact
    assert torch.equal(diff._last["xt"].cpu(), o32.aux["xt"])
    assert torch.equal(diff._last["move_indices"].cpu(), o32.aux["move_indices"])
    assert torch.equal(out.token_mask.cpu(), o32.token_mask)
    assert torch.equal(diff._last["t"].cpu(), o32.aux["t"])
    assert 0 < int(o32.aux["move_indices"].sum()) < o32.aux["move_indices"].numel()

    l, l32, l16 = float(out.loss.detach()), float(o32.loss.detach()), float(o16.loss.detach())
    floor = abs(l16 - l32) / abs(l32)
    record(name, "ref_bf16_vs_fp32_loss_rel", floor, note="the reference's own bf16-vs-fp32 noise floor (oracle bf16 emulation: bf16 log-softmax)")
    check(name, "loss_rel_vs_fp32_oracle", abs(l - l32) / abs(l32), bound["loss"])
    # the HIP path keeps the log-sum-exp in fp32, so it sits next to the fp32 truth; its distance to the bf16 emulation is that emulation's own error
    check(name, "loss_rel_vs_bf16_oracle", abs(l - l16) / abs(l16), floor + bound["loss"])
    record(name, "ref_bf16_vs_fp32_nll_relrms", _rel(o16.nlls.detach(), o32.nlls))
    check(name, "nll_relrms_vs_fp32_oracle", _rel(out.nlls.cpu(), o32.nlls), bound["nll"])
    assert torch.all(out.nlls.cpu()[~o32.aux["move_indices"]] == 0)   # unmasked tokens: nll exactly 0
    for k in ("txt_loss", "img_loss"):
        v = getattr(o32, k)
        if isinstance(v, torch.Tensor) and float(v) != 0:
            check(name, f"{k}_rel_vs_fp32_oracle", abs(float(getattr(out, k)) - float(v)) / abs(float(v)), 3 * bound["loss"])

    out.loss.backward()
    torch.cuda.synchronize()
    errs = []
    for k, p in diff.backbone.named_parameters():
        if P[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        errs.append((_rel(p.grad.cpu(), P[k].grad), k))
    errs.sort(reverse=True)
    floor_f = {k: _rel(P16f[k].grad, P[k].grad) for k in P if P[k].grad is not None}
    ff = sorted(((v, k) for k, v in floor_f.items()), reverse=True)
    record(name, "ref_bf16_flash_rounding_vs_fp32_grad_relrms_worst_param", ff[0][0], note=ff[0][1])
    record(name, "ref_bf16_flash_rounding_vs_fp32_grad_relrms_median_param", ff[len(ff) // 2][0])
    record(name, "ref_bf16_flash_rounding_vs_fp32_grad_relrms_same_param_as_ours", floor_f[errs[0][1]], note=errs[0][1])
    ratios = sorted(((e / max(floor_f[k], 1e-12), k) for e, k in errs), reverse=True)
    # VERDICT r02 (weak 1): the worst parameter - always a qk-norm vector, a column sum of dq / dk with heavy cancellation - must sit within 1.5x the reference's
    # own bf16 noise for the worst parameter once that floor includes the flash-attention backward's rounding points (measured 0.85-1.3x per parameter at
    # 1 / 2 / 12 / 24 blocks; against the boundary-rounding-only floor the same gradients look 3-6x off, which was the round-2 finding)
    check(name, "grad_relrms_worst_param_over_flash_rounding_floor_worst", errs[0][0] / ff[0][0], 1.5, note=f"{errs[0][1]} vs {ff[0][1]}")
    check(name, "grad_err_over_flash_rounding_floor_worst_ratio", ratios[0][0], 2.0, note=ratios[0][1])
    record(name, "grad_err_over_flash_rounding_floor_median_ratio", ratios[len(ratios) // 2][0])
    if os.environ.get("UDM_DUMP_GRAD_ERRS"):   # per-parameter table (ours, boundary-rounding floor, flash-rounding floor) for diagnosis
        import json
        with open(os.environ["UDM_DUMP_GRAD_ERRS"] + f".{name}.json", "w") as f:
            json.dump({k: dict(ours=e, floor_flash=floor_f[k], numel=P[k].numel(), gnorm=float(P[k].grad.norm())) for e, k in errs}, f, indent=0)
    if case.get("time_conditioning"):
        # the conditioning path's gradients flow through the per-block column sums / the deferred adaLN_modulation backward: each one by name, not through a median
        ada = [(e, k) for e, k in errs if k.startswith("sigma_map.") or "adaLN_modulation" in k]
        n_ada = 4 + 2 * case["n_blocks"] + 2
        assert len(ada) == n_ada, (len(ada), n_ada, [k for _, k in ada])
        for e, k in ada:
            assert float(P[k].grad.abs().max()) > 0, k
            check(name, f"grad_relrms[{k}]", e, 2.4e-2)      # achieved 4.0e-3 .. 7.8e-3 at d = 2048 (profiles/r06_parity_ledger.json)
    if bound["grad_max"] is not None:
        check(name, "grad_relrms_worst_param", errs[0][0], bound["grad_max"], note=errs[0][1])
    else:
        record(name, "grad_relrms_worst_param", errs[0][0], note=errs[0][1] + " (recorded: asserted through the floor ratios)")
    check(name, "grad_relrms_median_param", errs[len(errs) // 2][0], bound["grad_med"])
    allg = torch.cat([p.grad.reshape(-1).cpu() for k, p in diff.backbone.named_parameters() if P[k].grad is not None])
    allo = torch.cat([P[k].grad.reshape(-1) for k, p in diff.backbone.named_parameters() if P[k].grad is not None])
    check(name, "grad_relrms_all_params", _rel(allg, allo), bound["grad_med"])


# ------------------------------------------------------------------------------------------------ sampler-facing outputs at full width
def test_forward_logprobs_logits_and_guided_sampling_at_full_width():
    """`Diffusion.forward` (model.py:674-795) at BASELINE configs[2]'s real width and vocabulary (d = 2048, V = 48 385, L = 1280, one block, B = 2) against the
    oracle: raw logits (`return_logits=True`) and SUBS log-probs - the same set of finite entries, rel-RMS of the finite ones within 1.25 x the oracle's own
    bf16 emulation + 5e-4 - and one guided reverse-diffusion update (`udm_ddpm_sample_rows_cfg`, model_eval.py:1761-1834, 2073-2106) on THOSE bf16 logits
    against the oracle's update on the same logits and uniforms: token-exact."""
    from unidisc_amd import Diffusion
    from unidisc_amd import kernels as K

    case, B = dict(_LARGE, n_blocks=1), 2
    name = "config_c_1block_b2_forward"
    cfg = product_config(case)
    torch.manual_seed(0)
    diff = Diffusion(cfg, None, DEV)
    diff.backbone.eval()
    wg = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in sorted(diff.backbone.named_parameters()):
            if n.endswith("linear.weight"):
                p.copy_((torch.randn(p.shape, generator=wg) * (0.5 / p.shape[-1] ** 0.5)).to(DEV))
    P = {k: v.detach().cpu().clone() for k, v in diff.backbone.named_parameters()}
    ocfg = O.OracleConfig.from_case(case)
    bufs = O.make_buffers(ocfg, lumina_rope_2d)
    gen = torch.Generator().manual_seed(91)
    Lt, Li, Vt, V = case["txt_length"], case["img_length"], case["text_vocab_size"], case["vocab_size"]
    L, mask = Lt + Li, Vt - 1
    x0 = torch.cat([torch.randint(0, Vt - 1, (B, Lt), generator=gen), torch.randint(Vt, V, (B, Li), generator=gen)], 1)
    modality = torch.cat([torch.zeros(B, Lt, dtype=torch.int64), torch.ones(B, Li, dtype=torch.int64)], 1)
    xt = torch.where(torch.rand(B, L, generator=gen) < 0.6, torch.full_like(x0, mask), x0)
    t = torch.tensor([0.7, 0.35])
    sigma = O.loglinear_noise(t)[0]
    with torch.no_grad():
        lg32 = O.dit_forward(ocfg, P, bufs, xt, sigma, modality, None, False)
        lg16 = O.dit_forward(ocfg, P, bufs, xt, sigma, modality, None, True)
        lp32 = O.subs_parameterization(ocfg, lg32, xt, modality, None, False).float()
        lp16 = O.subs_parameterization(ocfg, lg16, xt, modality, None, True).float()
        logits = diff.forward(xt.to(DEV), sigma.to(DEV), batch=None, modality=modality.to(DEV), return_logits=True)
        lp = diff.forward(xt.to(DEV), sigma.to(DEV), batch=None, modality=modality.to(DEV))
    assert logits.dtype == torch.bfloat16 and tuple(logits.shape) == (B, L, V) == tuple(lp.shape)
    floor = _rel(lg16.float(), lg32)
    record(name, "ref_bf16_vs_fp32_logits_relrms", floor, note="oracle with the reference's bf16 rounding points against the fp32 oracle")
    check(name, "logits_relrms_vs_fp32_oracle", _rel(logits.float().cpu(), lg32), 1.25 * floor + 5e-4)
    finite = lp32 > -1e5
    assert torch.equal(lp.float().cpu() > -1e5, finite)                       # carry-over rows, the [MASK] column and the other modality's ids: the same -inf pattern
    lp_floor = _rel(lp16[finite], lp32[finite])
    record(name, "ref_bf16_vs_fp32_logprobs_relrms", lp_floor)
    check(name, "logprobs_relrms_vs_fp32_oracle", _rel(lp.float().cpu()[finite], lp32[finite]), 1.25 * lp_floor + 5e-4)
    # one guided update on the product's own bf16 logits: conditional = logits, unconditional = a second forward with the text masked out
    x_un = xt.clone()
    x_un[:, :Lt] = mask
    with torch.no_grad():
        logits_u = diff.forward(x_un.to(DEV), sigma.to(DEV), batch=None, modality=modality.to(DEV), return_logits=True)
    w = torch.tensor([1.5, 0.5])
    dt = 0.05
    rows = (xt.reshape(-1) == mask).nonzero().reshape(-1)
    u = torch.rand(rows.numel(), V, generator=gen)
    lc, lu = logits.float().cpu().reshape(B * L, V)[rows], logits_u.float().cpu().reshape(B * L, V)[rows]
    b_of = rows // L
    mixed = (1 + w[b_of, None]) * lc - w[b_of, None] * lu
    lpm = O.subs_parameterization(ocfg, mixed[None], None, modality.reshape(1, -1)[:, rows], None, False).float()[0]
    q = lpm.exp() * dt
    q[:, mask] = (t - dt)[b_of]
    want = O.sample_categorical(q[None], u[None])[0]
    Vp = logits.shape[-1] if logits.stride(1) == logits.shape[-1] else V
    pad = lambda z: torch.cat([z.reshape(B * L, V).index_select(0, rows.to(z.device)), torch.zeros(rows.numel(), (V + 7) // 8 * 8 - V, dtype=z.dtype, device=z.device)], 1).contiguous()
    tok = K.ddpm_sample_rows(pad(logits), V, Vt, mask, t=t[b_of].to(DEV), s=(t - dt)[b_of].to(DEV), modality=modality.reshape(-1)[rows].to(DEV), restrict=True,
                             u=u.to(DEV), logits_u=pad(logits_u), w=w[b_of].contiguous().to(DEV)).cpu()
    agree = float((tok == want).float().mean())
    record(name, "guided_update_token_agreement", agree, note=f"{rows.numel()} [MASK] rows, V = {V}")
    assert torch.equal(tok, want)


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: packed rows, L = 4608
_PACKED = dict(_LARGE, txt_length=512, img_length=4096, interleaved=True, img_loss_weight=0.2, mask_entire_modality=0.2)


def _packed_batch(case, B, gen, samples=4, txt=128, img=1024):
    Vt, V = case["text_vocab_size"], case["vocab_size"]
    ids, mod, sid = [], [], []
    for s_ in range(samples):
        ids += [torch.randint(0, Vt - 1, (B, txt), generator=gen), torch.randint(Vt, V, (B, img), generator=gen)]
        mod += [torch.zeros(B, txt, dtype=torch.int64), torch.ones(B, img, dtype=torch.int64)]
        sid += [torch.full((B, txt + img), s_, dtype=torch.int64)]
    ids = torch.cat(ids, 1)
    return dict(input_ids=ids, modality=torch.cat(mod, 1), sample_ids=torch.cat(sid, 1), attention_mask=torch.ones_like(ids, dtype=torch.bool))


def test_config_e_packed_l4608_matches_oracle():
    """BASELINE configs[4] at its real width and length (d = 2048, D = 128, rows of 4 packed samples = 4608 tokens, document mask from the sample ids),
    one block: the same seeded step through the fp32 oracle and the product.  Masks bit-exact; ledger rows `config_e_1block_b2_*`.  (The fp8 attention
    forward that BASELINE configs[4] names was built in round 2-4, never beat the bf16 kernel inside the step and was 12.8 x noisier on the qk-norm
    gradients: removed in round 5)"""
    from unidisc_amd import Diffusion

    case, B = dict(_PACKED, n_blocks=1), 2   # (B = 1 trips the reference's own `.squeeze(-1)` on the interleaved ignore mask, model.py - the oracle restates it)
    name = "config_e_1block_b2"
    cfg = product_config(case)
    batch = _packed_batch(case, B, torch.Generator().manual_seed(78))
    ocfg = O.OracleConfig.from_case(case)
    bufs = O.make_buffers(ocfg, lumina_rope_2d)
    res = {}
    P = None
    for mode in ("bf16",):
        torch.manual_seed(0)
        diff = Diffusion(cfg, None, DEV)
        diff.backbone.train()
        diff.rng_device = "cpu"
        wg = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for n, p in sorted(diff.backbone.named_parameters()):
                if n.endswith("linear.weight"):
                    p.copy_((torch.randn(p.shape, generator=wg) * (0.5 / p.shape[-1] ** 0.5)).to(DEV))
        if P is None:
            P = {k: v.detach().cpu().clone().requires_grad_() for k, v in diff.backbone.named_parameters()}
            ob = O.update_batch(ocfg, {k: v.clone() for k, v in batch.items()})
            o32 = O.compute_loss(ocfg, P, bufs, ob, torch.Generator().manual_seed(123))
            o32.loss.backward()
        torch.manual_seed(123)
        out = diff.training_step({k: v.clone() for k, v in batch.items()}, 1)
        assert torch.equal(diff._last["xt"].cpu(), o32.aux["xt"]) and torch.equal(diff._last["move_indices"].cpu(), o32.aux["move_indices"])
        assert torch.equal(out.token_mask.cpu(), o32.token_mask)
        out.loss.backward()
        torch.cuda.synchronize()
        res[mode] = (float(out.loss.detach()), out.nlls.detach().cpu(), {k: p.grad.cpu() for k, p in diff.backbone.named_parameters() if p.grad is not None})
        del diff
        torch.cuda.empty_cache()
    l32 = float(o32.loss.detach())
    # stated = asserted tolerances against the fp32 oracle: as the other full-width rows
    tol = dict(bf16=dict(loss=5e-5, nll=1.5e-3, grad_max=6e-2, grad_med=2e-2))
    for mode in ("bf16",):
        l, nll, g = res[mode]
        T = f"{name}_{mode}_attention"
        check(T, "loss_rel_vs_fp32_oracle", abs(l - l32) / abs(l32), tol[mode]["loss"])
        check(T, "nll_relrms_vs_fp32_oracle", _rel(nll, o32.nlls), tol[mode]["nll"])
        errs = sorted(((_rel(g[k], P[k].grad), k) for k in g if P[k].grad is not None), reverse=True)
        check(T, "grad_relrms_worst_param", errs[0][0], tol[mode]["grad_max"], note=errs[0][1])
        check(T, "grad_relrms_median_param", errs[len(errs) // 2][0], tol[mode]["grad_med"])
