"""Modality attention dropout (`model.flex_attention_{txt,img}_masking_prob`, reference model.py:863-878 + model_utils.py:721-737; shipped in
configs/experiments/small_scale_train_caching.yaml:34-35): per sample, text queries may be restricted to text keys and / or image queries to image keys.

Fixture: tests/golden/g_attn_dropout.npz from the imported reference (oracle/make_golden_attn_dropout.py; FlexAttention replaced by its definition, dense-mask
SDPA).  CPU: the oracle reproduces the reference's draws, masks, loss and gradients; the product's host logic (draw order, the `& ~should_mask_*` rule, the
mask codes handed to the attention kernels) with kernel doubles.  GPU: the attention kernels with mask codes against dense-mask SDPA, and the training step
against the golden."""
import pytest
import torch

import fake_kernels
from golden_utils import Golden, rel_err
from oracle import unidisc_oracle as O
from product_utils import build_product

NAME = "g_attn_dropout"


def test_oracle_reproduces_reference_with_attention_dropout():
    g = Golden(NAME)
    P = g.params(requires_grad=True)
    ob = O.update_batch(g.cfg, g.batch())
    out = O.compute_loss(g.cfg, P, g.buffers(), ob, g.generator())
    out.loss.backward()
    assert torch.equal(out.aux["xt"], g.t("fp32/xt")) and torch.equal(out.aux["move_indices"], g.t("fp32/move_indices"))
    want = O.modality_dropout_mask(g.t("fp32/txt_attn_dropout"), g.t("fp32/img_attn_dropout"), g.cfg.txt_length, g.cfg.length)
    assert torch.equal(out.aux["allow_mask"], want)                      # the same two per-sample draws, after the `& ~should_mask_*` rule
    assert bool(g.t("fp32/txt_attn_dropout").any()) and bool(g.t("fp32/img_attn_dropout").any()) and not bool(want.all())
    assert torch.equal(out.token_mask, g.t("fp32/token_mask"))
    assert abs(float(out.loss) - float(g.t("fp32/loss"))) <= 1e-5 * abs(float(g.t("fp32/loss")))
    assert rel_err(out.aux["logits"].detach(), g.t("fp32/logits")) < 1e-5
    worst = max(rel_err(P[k].grad, v) for k, v in g.grads("fp32").items())
    assert worst < 2e-4, worst


def test_mask_codes_describe_the_reference_mask():
    """`kernels.modality_mask_codes` + the pair predicate of csrc/attention_common.h (restated by the kernel double) == `_attn_mask` of the reference."""
    from unidisc_amd import kernels as K

    B, L, Lt = 4, 24, 10
    td = torch.tensor([False, True, False, True])
    idr = torch.tensor([False, False, True, True])
    code = K.modality_mask_codes(td, idr, Lt, L)
    ids = ((code & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000
    kb, qm = (code >> 32) & 0xFF, (code >> 40) & 0xFF
    allow = (ids[:, :, None] == ids[:, None, :]) & (ids[:, :, None] >= 0) & ((qm[:, :, None] & kb[:, None, :]) != 0)
    assert torch.equal(allow, O.modality_dropout_mask(td, idr, Lt, L))


def test_product_host_logic_with_attention_dropout(monkeypatch):
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g = Golden(NAME)
    diff = build_product(g, device="cpu")
    diff.rng_device = "cpu"
    torch.manual_seed(g.case["step_seed"])
    out = diff.training_step(g.batch(), 1)
    assert torch.equal(diff._last["xt"], g.t("fp32/xt")) and torch.equal(out.token_mask, g.t("fp32/token_mask"))
    l32 = float(g.t("fp32/loss"))
    assert abs(float(out.loss) - l32) <= 5e-3 * abs(l32), (float(out.loss), l32)
    out.loss.backward()
    named = dict(diff.backbone.named_parameters())
    for k, gr in g.grads("fp32").items():
        assert rel_err(named[k].grad, gr) < 0.06, k
    # without the dropout the same step gives another loss: the mask is really applied
    diff.config.model.flex_attention_txt_masking_prob = diff.config.model.flex_attention_img_masking_prob = None
    torch.manual_seed(g.case["step_seed"])
    assert abs(float(diff.training_step(g.batch(), 1).loss) - l32) > 1e-3 * abs(l32)


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,L,Lt", [(128, 2, 320, 64), (64, 3, 200, 72), (32, 2, 130, 40)])
def test_gpu_attention_kernels_with_mask_codes(D, H, L, Lt):
    """Forward, dQ and dK / dV with the asymmetric modality mask (codes in the sample-id slot) against dense-mask SDPA in fp32."""
    from unidisc_amd import kernels as K

    B, d = 4, H * D
    gen = torch.Generator().manual_seed(3)
    q, k, v, do = ((torch.randn(B * L, d, generator=gen) * 0.7).bfloat16() for _ in range(4))
    td = torch.tensor([False, True, False, True])
    idr = torch.tensor([False, False, True, True])
    allow = O.modality_dropout_mask(td, idr, Lt, L)
    qq, kk, vv = (t.float().reshape(B, L, H, D).transpose(1, 2).clone().requires_grad_() for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv, attn_mask=allow[:, None])
    ref.backward(do.float().reshape(B, L, H, D).transpose(1, 2))
    back = lambda t: t.transpose(1, 2).reshape(B * L, d)
    code = K.modality_mask_codes(td, idr, Lt, L).cuda()
    qd, kd, vd, dod = (t.cuda() for t in (q, k, v, do))
    o, lse = K.attention_fwd_generic(qd, kd, vd, B, L, H, D, sample_ids=code)
    assert rel_err(o.float().cpu(), back(ref.detach())) < 6e-3
    dq, dk, dv = K.attention_bwd_generic(qd, kd, vd, o, dod, lse, B, L, H, D, sample_ids=code)
    assert rel_err(dq.float().cpu(), back(qq.grad)) < 1.5e-2
    assert rel_err(dk.float().cpu(), back(kk.grad)) < 1.5e-2
    assert rel_err(dv.float().cpu(), back(vv.grad)) < 1.5e-2
    # and it is not the unmasked result
    o_full, _ = K.attention_fwd_generic(qd, kd, vd, B, L, H, D)
    assert rel_err(o_full.float().cpu(), back(ref.detach())) > 5e-2


@pytest.mark.gpu
def test_gpu_training_step_with_attention_dropout_matches_golden():
    from ledger import check

    g = Golden(NAME)
    diff = build_product(g, "cuda")
    diff.rng_device = "cpu"
    torch.manual_seed(g.case["step_seed"])
    out = diff.training_step(g.batch(), 1)
    assert torch.equal(diff._last["xt"].cpu(), g.t("fp32/xt")) and torch.equal(out.token_mask.cpu(), g.t("fp32/token_mask"))
    l32 = float(g.t("fp32/loss"))
    T = f"golden_step[{NAME}]"
    check(T, "loss_rel_vs_fp32_reference", abs(float(out.loss.detach()) - l32) / abs(l32), 1e-3)
    check(T, "nll_relrms_vs_fp32_reference", rel_err(out.nlls.cpu(), g.t("fp32/nlls")), 4.5e-3)
    out.loss.backward()
    torch.cuda.synchronize()
    named = dict(diff.backbone.named_parameters())
    errs = sorted(((rel_err(named[k].grad.cpu(), gr), k) for k, gr in g.grads("fp32").items()), reverse=True)
    check(T, "grad_relrms_worst_param", errs[0][0], 6e-2, note=errs[0][1])
    check(T, "grad_relrms_median_param", errs[len(errs) // 2][0], 3e-2)
