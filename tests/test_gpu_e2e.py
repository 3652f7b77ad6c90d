"""End-to-end parity on a real MI355X (pytest -m gpu): the HIP path vs the golden vectors of the imported
reference and vs the CPU oracle on the same seeded inputs.

Tolerances: every comparison is recorded in the parity ledger (tests/ledger.py -> profiles/r05_parity_ledger.json) together with the
reference's own bf16-vs-fp32 deviation (~1e-3 on the loss, ~3e-3 rel-RMS on logits), and asserted against a STATED bound that is
<= 3x the error achieved there: loss 1e-3 relative (north_star's figure), logits / NLL / gradients as rel-RMS against the reference's fp32 run.
Mask indices (xt, move_indices, token_mask) must be bit-exact.
"""
import pytest
import torch

import unidisc_amd.diffusion as diff_mod
from golden_utils import CASE_NAMES, Golden, rel_err
from ledger import check, record
from oracle import unidisc_oracle as O
from oracle.cases import lumina_rope_2d
from product_utils import build_product, product_config

pytestmark = pytest.mark.gpu
DEV = "cuda"
# Stated bounds (<= 3x the errors recorded in profiles/r05_parity_ledger.json; d = 64 goldens, so per-parameter gradients of 64-element vectors are
# the noisiest quantity): relative error of the loss, rel-RMS of logits / per-token NLL / per-parameter gradients against the reference's fp32 run.
LOSS_BOUND, NLL_BOUND, GRAD_BOUND = 1e-3, 4.5e-3, 6e-2


def logits_bound(ref_floor):
    """bf16 logits against the reference's fp32 run: north_star's literal 1e-3 cannot be met by ANY bf16 evaluation - the reference's own bf16 run sits
    3.0e-3 .. 6.3e-3 rel-RMS from its fp32 run.  The bound is therefore tied to that floor, case by case: 1.25 x the reference's own bf16
    deviation + 5e-4 (a flat 1e-2 was 1.6 - 3.3 x what is achieved)."""
    return 1.25 * ref_floor + 5e-4


@pytest.mark.parametrize("name", CASE_NAMES)
def test_logits_match_golden(name):
    g = Golden(name)
    diff = build_product(g, DEV)
    xt = g.t("fp32/xt").to(DEV)
    sigma = O.loglinear_noise(g.t("fp32/t"))[0].to(DEV)
    modality = g.t("fp32/modality").to(DEV) if g.has("fp32/modality") and g.case["multimodal_batches"] else None
    with torch.no_grad():
        logits = diff.backbone(xt, sigma, modality=modality)
    truth, ref16 = g.t("fp32/logits"), g.t("bf16/logits")
    assert logits.dtype == torch.bfloat16 and logits.shape == truth.shape
    e, budget = rel_err(logits.float().cpu(), truth), rel_err(ref16, truth)
    record(f"golden_logits[{name}]", "ref_bf16_vs_fp32_logits_relrms", budget, note="the reference's own bf16 run against its fp32 run")
    check(f"golden_logits[{name}]", "logits_relrms_vs_fp32_reference", e, logits_bound(budget))
    # ... and directly against the reference's OWN bf16 run (CPU bf16 autocast): two bf16 evaluations of the same network differ by about sqrt(2) x the distance of
    # either from fp32 (independent roundings), so the claim "these are the reference's bf16 numerics" is this row staying at that level - asserted at
    # 2 x the reference's own bf16-vs-fp32 deviation + 2e-3 (the SURVEY F9 budget)
    check(f"golden_logits[{name}]", "logits_relrms_vs_bf16_reference", rel_err(logits.float().cpu(), ref16), 2 * budget + 2e-3)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_training_step_matches_golden(name):
    g = Golden(name)
    diff = build_product(g, DEV)
    diff.rng_device = "cpu"  # replay the reference's CPU generator stream
    torch.manual_seed(g.case["step_seed"])
    out = diff.training_step(g.batch(), 1)
    # integer / boolean quantities: bit-exact
    assert torch.equal(diff._last["xt"].cpu(), g.t("fp32/xt"))
    assert torch.equal(diff._last["move_indices"].cpu(), g.t("fp32/move_indices"))
    assert torch.equal(out.token_mask.cpu(), g.t("fp32/token_mask"))
    assert torch.equal(diff._last["t"].cpu(), g.t("fp32/t"))
    l32, l16 = float(g.t("fp32/loss")), float(g.t("bf16/loss"))
    T = f"golden_step[{name}]"
    record(T, "ref_bf16_vs_fp32_loss_rel", abs(l16 - l32) / abs(l32), note="the reference's own bf16 run against its fp32 run")
    check(T, "loss_rel_vs_fp32_reference", abs(float(out.loss.detach()) - l32) / abs(l32), LOSS_BOUND)
    nll_truth, nll_ref = g.t("fp32/nlls"), g.t("bf16/nlls")
    e, budget = rel_err(out.nlls.cpu(), nll_truth), rel_err(nll_ref, nll_truth)
    record(T, "ref_bf16_vs_fp32_nll_relrms", budget)
    check(T, "nll_relrms_vs_fp32_reference", e, NLL_BOUND)
    assert torch.all(out.nlls.cpu()[~g.t("fp32/move_indices")] == 0)  # unmasked tokens: nll exactly 0
    for k in ("txt_loss", "img_loss"):
        if g.has("fp32/" + k):
            v32 = float(g.t("fp32/" + k))
            check(T, f"{k}_rel_vs_fp32_reference", abs(float(getattr(out, k)) - v32) / max(abs(v32), 1e-6), 3 * LOSS_BOUND)
    out.loss.backward()
    torch.cuda.synchronize()
    gref, gb16 = g.grads("fp32"), g.grads("bf16")
    named = dict(diff.backbone.named_parameters())
    assert set(gref) == {k for k, p in named.items() if p.grad is not None}
    errs = sorted(((rel_err(named[k].grad.cpu(), gr), k) for k, gr in gref.items()), reverse=True)
    floors = sorted((rel_err(gb16[k], gr) for k, gr in gref.items()), reverse=True)
    record(T, "ref_bf16_vs_fp32_grad_relrms_worst_param", floors[0])
    check(T, "grad_relrms_worst_param", errs[0][0], GRAD_BOUND, note=errs[0][1])
    check(T, "grad_relrms_median_param", errs[len(errs) // 2][0], GRAD_BOUND / 2)


def test_matches_oracle_at_unidisc_s_width():
    """Same seeded inputs through the oracle (CPU fp32) and the HIP path at UniDisc-S width (d=768, H=12, D=64,
    joint 32001+8192 vocabulary) with 2 blocks and a short joint sequence — a size the oracle finishes in seconds."""
    case = dict(hidden_size=768, n_heads=12, cond_dim=128, n_blocks=2, batch_size=2, txt_length=64, img_length=64, text_vocab_size=32001,
                vocab_size=40193, norm_type="rms", qk_norm=True, sandwich_normalization=True, modality_embed=True, rope_2d=False,
                time_conditioning=False, multimodal_batches=True, force_argmax_valid_indices=True, mask_entire_modality=0.1, softmin_snr=5,
                text_loss_weight=1.0, img_loss_weight=None, force_full_attention_mask_loss_only=True)
    from unidisc_amd import Diffusion

    cfg = product_config(case)
    torch.manual_seed(0)
    diff = Diffusion(cfg, None, DEV)
    diff.backbone.train()
    diff.rng_device = "cpu"
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in sorted(diff.backbone.named_parameters()):
            if n.endswith("linear.weight"):
                p.copy_((torch.randn(p.shape, generator=gen) / p.shape[-1] ** 0.5).to(DEV))
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in diff.backbone.named_parameters()}
    B, Lt, Li = 2, 64, 64
    batch = dict(txt_input_ids=torch.randint(0, 32000, (B, Lt), generator=gen, dtype=torch.int32),
                 img_input_ids=torch.randint(0, 8192, (B, Li), generator=gen, dtype=torch.int32).to(torch.int16),
                 txt_attention_mask=torch.ones(B, Lt, dtype=torch.bool))
    ocfg = O.OracleConfig.from_case(case)
    ob = O.update_batch(ocfg, {k: v.clone() for k, v in batch.items()})
    oout = O.compute_loss(ocfg, P, O.make_buffers(ocfg, lumina_rope_2d), ob, torch.Generator().manual_seed(123))
    oout.loss.backward()
    torch.manual_seed(123)
    out = diff.training_step(batch, 1)
    assert torch.equal(diff._last["xt"].cpu(), oout.aux["xt"])
    T = "oracle_step[unidisc_s_width_2blocks]"
    check(T, "loss_rel_vs_fp32_oracle", abs(float(out.loss.detach()) - float(oout.loss.detach())) / abs(float(oout.loss.detach())), LOSS_BOUND)
    check(T, "nll_relrms_vs_fp32_oracle", rel_err(out.nlls.cpu(), oout.nlls), NLL_BOUND)
    out.loss.backward()
    torch.cuda.synchronize()
    errs = sorted(((rel_err(p.grad.cpu(), P[k].grad), k) for k, p in diff.backbone.named_parameters()), reverse=True)
    check(T, "grad_relrms_worst_param", errs[0][0], GRAD_BOUND, note=errs[0][1])
    check(T, "grad_relrms_median_param", errs[len(errs) // 2][0], GRAD_BOUND / 2)


def test_dropout_training_runs_and_is_finite():
    g = Golden("c_large")
    cfg = product_config(g.case)
    cfg.model.dropout = 0.1
    from unidisc_amd import Diffusion

    diff = Diffusion(cfg, None, DEV)
    diff.backbone.load_state_dict(g.params())
    diff.backbone.to(DEV).train()
    out = diff.training_step(g.batch(), 1)
    out.loss.backward()
    assert torch.isfinite(out.loss)
    assert all(torch.isfinite(p.grad).all() for p in diff.backbone.parameters())
    diff.backbone.eval()
    with torch.no_grad():
        a = diff.backbone(g.t("fp32/xt").to(DEV), None, modality=g.t("fp32/modality").to(DEV))
        b = diff.backbone(g.t("fp32/xt").to(DEV), None, modality=g.t("fp32/modality").to(DEV))
    assert torch.equal(a, b)


def test_subs_logprobs_forward_contract():
    g = Golden("b_small")
    diff = build_product(g, DEV)
    xt, mod = g.t("fp32/xt").to(DEV), g.t("fp32/modality").to(DEV)
    with torch.no_grad():
        lp = diff.forward(xt, None, batch=None, modality=mod)
        logits = diff.forward(xt, None, batch=None, modality=mod, return_logits=True)
    assert lp.shape == logits.shape == tuple(g.t("fp32/log_probs").shape)
    ref = g.t("fp32/log_probs")
    finite = ref > -1e5
    assert torch.equal(lp.float().cpu() > -1e5, finite)
    assert torch.allclose(lp.float().cpu()[finite], ref[finite], atol=0.12, rtol=0.02)


@pytest.mark.parametrize("frac", [0.0, 1.0])
def test_no_masked_rows_and_all_masked_rows(frac):
    """Extremes of the corruption: a batch without a single [MASK] (the compacted vocabulary head has zero real rows: log p = 0 everywhere, loss 0,
    all gradients exactly 0) and a fully masked batch (every row goes through the head); both must agree with the oracle."""
    g = Golden("c_large")
    diff = build_product(g, DEV)
    b = O.update_batch(g.cfg, g.batch())
    x0, mod = b["input_ids"], b["modality"]
    xt = torch.where(torch.full_like(x0, frac, dtype=torch.float32) > 0.5, torch.full_like(x0, diff.mask_index), x0)
    lp = diff.backbone.forward_logp(xt.to(DEV), x0.to(DEV), None, modality=mod.to(DEV), restrict_modality=True)
    P, buf = g.params(), g.buffers()
    logits = O.dit_forward(g.cfg, P, buf, xt, None, mod, None, False)
    ref = O.subs_parameterization(g.cfg, logits, xt, mod).gather(-1, x0[..., None]).squeeze(-1)
    if frac == 0.0:
        assert torch.all(lp == 0) and torch.all(ref == 0)
    else:
        assert rel_err(lp.float().cpu(), ref) < 1e-2
    lp.sum().backward()
    torch.cuda.synchronize()
    grads = [p.grad for p in diff.backbone.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(x).all() for x in grads)
    if frac == 0.0:
        assert all(float(x.abs().max()) == 0.0 for x in grads)


@pytest.mark.gpu
def test_chunked_head_and_cross_entropy_on_gpu():
    """model.head_chunk_rows on the device: same loss (bit-equal log-probabilities per row: the chunks run the same kernels on the same rows) and
    the same gradients up to the summation order of the head's weight gradient across chunks."""
    g = Golden("c_large")
    res = []
    for chunk in (0, 64):
        diff = build_product(g, device=DEV)
        diff.rng_device = "cpu"
        diff.backbone.head_chunk_rows = chunk
        torch.manual_seed(g.case["step_seed"])
        batch = {k: torch.cat([v] * 4).to(DEV) for k, v in g.batch().items()}
        out = diff.training_step(batch, 1)
        out.loss.backward()
        res.append((float(out.loss), out.nlls.detach().cpu(), {k: p.grad.cpu() for k, p in diff.backbone.named_parameters() if p.grad is not None}))
    (l0, n0, g0), (l1, n1, g1) = res
    assert l0 == l1 and torch.equal(n0, n1)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 2e-3, (k, rel_err(g1[k], g0[k]))


def test_modality_range_check_is_deferred_but_still_raises():
    """model.py:311's assert (both modalities present, nothing else) no longer synchronises the host at the top of the step, but a bad batch still raises
    the same AssertionError inside the same training step, before a loss comes back."""
    g = Golden("c_large")
    diff = build_product(g, DEV)
    torch.manual_seed(0)
    batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in g.batch().items()}   # (a host batch is checked on the host, immediately)
    out = diff.training_step(batch, 1)   # a good batch passes and leaves nothing pending
    assert torch.isfinite(out.loss) and not diff._checks
    bad = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in diff.update_batch(batch).items()}
    diff._flush_checks()
    bad["modality"] = torch.zeros_like(bad["modality"])   # text only: max() == 0
    with pytest.raises(AssertionError):
        diff.training_step({k: v for k, v in bad.items() if k in ("input_ids", "attention_mask", "modality")}, 2)
    assert not diff._checks


def test_update_batch_device_token_batches_take_the_fused_assembly():
    """A token batch that is already on the device in the dataset's dtypes is assembled by one launch (tokens.hip) instead of the reference's tensor statements:
    same joint ids, attention mask, modality map and derived fields, bit for bit, as the host batch that goes through the statements."""
    g = Golden("c_large")
    diff = build_product(g, DEV)
    host = g.batch()
    assert host["txt_input_ids"].dtype == torch.int32 and host["img_input_ids"].dtype == torch.int16
    a = diff.update_batch(host)
    calls = []
    orig = diff_mod.K.assemble_joint_tokens
    diff_mod.K.assemble_joint_tokens = lambda *args, **kw: (calls.append(1), orig(*args, **kw))[1]
    try:
        b = diff.update_batch({k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in host.items()})
    finally:
        diff_mod.K.assemble_joint_tokens = orig
    diff._flush_checks()
    assert calls == [1]
    assert set(a) == set(b)
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert a[k].dtype == b[k].dtype and torch.equal(a[k].cpu(), b[k].cpu()), k


def test_key_padding_mask_use_attention_mask_gpu():
    """`model.use_attention_mask` on the HIP path (padding mask -> key mask codes in the attention kernels, forward and backward): logits against the oracle with
    the same dense allow-mask, and the training step (config switch -> get_cond_dict -> backbone) differs from the unmasked one and has finite gradients."""
    g = Golden("c_large")
    diff = build_product(g, DEV)
    diff.backbone.eval()
    xt, mod = g.t("fp32/xt"), g.t("fp32/modality")
    B, L = xt.shape
    km = torch.ones(B, L, dtype=torch.bool)
    km[0, 5:9] = False
    km[1, L - 7:] = False
    with torch.no_grad():
        got = diff.backbone(xt.to(DEV), None, modality=mod.to(DEV), attention_mask=km.to(DEV)).float().cpu()
        base = diff.backbone(xt.to(DEV), None, modality=mod.to(DEV)).float().cpu()
        ref = O.dit_forward(g.cfg, g.params(), g.buffers(), xt, None, mod, None, False, allow_mask=km[:, None, :].expand(B, L, L))
    floor = rel_err(g.t("bf16/logits"), g.t("fp32/logits"))
    assert rel_err(got, ref) <= 3 * floor + 5e-3 and rel_err(base, ref) > 10 * rel_err(got, ref)
    # through the config switch, with a backward (a case whose update_batch keeps the padding mask: c_large forces the full mask)
    g = Golden("d_adaln_mm")
    diff = build_product(g, DEV)
    diff.backbone.train()
    diff.config.model.use_attention_mask = True
    diff.rng_device = "cpu"
    batch = g.batch()
    batch["txt_attention_mask"] = batch["txt_attention_mask"].clone()
    batch["txt_attention_mask"][0, -3:] = False
    torch.manual_seed(g.case["step_seed"])
    out = diff.training_step(batch, 1)
    out.loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out.loss) and all(torch.isfinite(p.grad).all() for p in diff.backbone.parameters() if p.grad is not None)
    diff.config.model.use_attention_mask = False
    diff.backbone.zero_grad(set_to_none=True)
    torch.manual_seed(g.case["step_seed"])
    out0 = diff.training_step(batch, 1)
    assert float(out0.loss.detach()) != float(out.loss.detach())


@pytest.mark.parametrize("name", ["c_large", "f_interleaved"])
def test_gradient_checkpointing_on_gpu(name):
    """trainer.use_gradient_checkpointing on the HIP path: each block is re-run from its saved input right before its backward (same kernels, same dropout
    seeds).  Same loss bit for bit; gradients equal up to the run-to-run noise of the backward's fp32 atomics; peak memory of the step is lower."""
    g = Golden(name)
    res = []
    for ck in (False, True):
        diff = build_product(g, DEV)
        diff.rng_device = "cpu"
        diff.backbone.use_gradient_checkpointing = ck
        diff.backbone.dropout = 0.1 if hasattr(diff.backbone, "dropout") else 0.0
        torch.manual_seed(g.case["step_seed"])
        torch.cuda.reset_peak_memory_stats()
        out = diff.training_step(g.batch(), 1)
        out.loss.backward()
        torch.cuda.synchronize()
        res.append((float(out.loss.detach()), {k: p.grad.float().cpu() for k, p in diff.backbone.named_parameters() if p.grad is not None}, torch.cuda.max_memory_allocated()))
    assert res[0][0] == res[1][0]
    assert set(res[0][1]) == set(res[1][1])
    for k in res[0][1]:
        assert rel_err(res[1][1][k], res[0][1][k]) < 2e-3, k
