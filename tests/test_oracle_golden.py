"""Pin the CPU oracle against golden vectors produced by the imported reference (CPU only)."""
import pytest
import torch

from oracle import unidisc_oracle as O
from oracle.cases import lumina_rope_2d
from golden_utils import CASE_NAMES, Golden, rel_err


@pytest.fixture(scope="module", params=CASE_NAMES)
def golden(request):
    return Golden(request.param)


def test_rotary_buffers_match_reference(golden):
    bufs = O.make_buffers(golden.cfg, lumina_rope_2d)
    ref = golden.buffers()
    for k, v in ref.items():
        assert k in bufs, k
        assert torch.allclose(bufs[k][: v.shape[0]], v, atol=1e-6), k


def test_update_batch_and_corruption_bit_exact(golden):
    cfg = golden.cfg
    b = O.update_batch(cfg, golden.batch())
    assert torch.equal(b["input_ids"], golden.t("fp32/input_ids"))
    assert torch.equal(b["attention_mask"], golden.t("fp32/attention_mask"))
    if golden.has("fp32/modality"):
        assert torch.equal(b["modality"], golden.t("fp32/modality"))
    g = golden.generator()
    t = O.sample_t(cfg, b["input_ids"].shape[0], g)
    assert torch.equal(t, golden.t("fp32/t"))
    sigma, _ = O.loglinear_noise(t)
    mc = 1 - torch.exp(-sigma[:, None])
    assert torch.equal(mc, golden.t("fp32/move_chance"))
    xt, ign, smt, smi, move = O.q_xt(cfg, b["input_ids"], mc, b, True, g)
    assert torch.equal(xt, golden.t("fp32/xt"))
    assert torch.equal(move, golden.t("fp32/move_indices"))
    if golden.has("fp32/ignore_batch_mask"):
        assert torch.equal(ign, golden.t("fp32/ignore_batch_mask"))
        assert torch.equal(smt, golden.t("fp32/should_mask_txt"))
        assert torch.equal(smi, golden.t("fp32/should_mask_img"))


def test_fp32_forward_loss_and_grads(golden):
    cfg = golden.cfg
    P = golden.params(requires_grad=True)
    b = O.update_batch(cfg, golden.batch())
    out = O.compute_loss(cfg, P, golden.buffers(), b, golden.generator(), bf16=False)
    aux = out.aux
    assert torch.equal(aux["xt"], golden.t("fp32/xt"))
    assert rel_err(aux["logits"].detach(), golden.t("fp32/logits")) < 1e-5
    lp_ref = golden.t("fp32/log_probs")
    finite = lp_ref > -1e5
    assert torch.equal(aux["log_probs"].detach() > -1e5, finite)
    assert torch.allclose(aux["log_probs"].detach()[finite], lp_ref[finite], atol=2e-5, rtol=1e-5)
    assert torch.allclose(out.nlls, golden.t("fp32/nlls"), atol=1e-5, rtol=1e-5)
    assert torch.equal(out.token_mask, golden.t("fp32/token_mask"))
    assert abs(float(out.loss) - float(golden.t("fp32/loss"))) <= 1e-5 * abs(float(golden.t("fp32/loss")))
    for k in ("txt_loss", "img_loss"):
        if golden.has("fp32/" + k):
            assert torch.allclose(getattr(out, k), golden.t("fp32/" + k), atol=1e-6, rtol=1e-5), k
    for k in ("txt_nlls", "img_nlls"):
        if golden.has("fp32/" + k):
            assert torch.allclose(getattr(out, k), golden.t("fp32/" + k), atol=1e-5, rtol=1e-5), k
    for k, v in out.extra_losses.items():
        assert torch.allclose(torch.as_tensor(v).float(), golden.t("fp32/extra/" + k), atol=1e-6), k
    out.loss.backward()
    gref = golden.grads("fp32")
    assert set(gref) == {k for k, p in P.items() if p.grad is not None}
    for k, g in gref.items():
        assert rel_err(P[k].grad, g) < 2e-4, (k, rel_err(P[k].grad, g))


def test_bf16_emulation_within_reference_noise_floor(golden):
    """The bf16-emulating oracle must sit as close to fp32 truth as the reference's own bf16 run (F9)."""
    cfg = golden.cfg
    P = golden.params()
    b = O.update_batch(cfg, golden.batch())
    with torch.no_grad():
        out = O.compute_loss(cfg, P, golden.buffers(), b, golden.generator(), bf16=True)
    truth, ref16 = golden.t("fp32/logits"), golden.t("bf16/logits")
    budget = rel_err(ref16, truth)
    ours = rel_err(out.aux["logits"], truth)
    assert ours <= 2.0 * budget + 1e-3, (ours, budget)
    l32, l16 = float(golden.t("fp32/loss")), float(golden.t("bf16/loss"))
    assert abs(float(out.loss) - l32) <= 2.0 * abs(l16 - l32) + 2e-3 * abs(l32)


def test_analytic_known_answers():
    """SURVEY §8c KATs: unmasked rows give nll 0; zero logits give log(#valid ids); w(t)=1/t up to eps."""
    cfg = O.OracleConfig(hidden_size=8, n_heads=1, cond_dim=4, n_blocks=0, txt_length=4, img_length=4, vocab_size=30,
                         text_vocab_size=11)
    B, L = 2, 8
    modality = torch.tensor([[0] * 4 + [1] * 4] * B)
    x0 = torch.where(modality == 0, torch.randint(0, 10, (B, L)), torch.randint(11, 30, (B, L)))
    xt = x0.clone()
    xt[:, ::2] = cfg.mask_index
    lp = O.subs_parameterization(cfg, torch.zeros(B, L, 30), xt, modality)
    log_p = torch.gather(lp, -1, x0[..., None]).squeeze(-1)
    assert torch.all(log_p[:, 1::2] == 0)
    assert torch.allclose(log_p[:, 0:4:2], torch.full((B, 2), -torch.log(torch.tensor(10.0))), atol=1e-6)
    assert torch.allclose(log_p[:, 4::2], torch.full((B, 2), -torch.log(torch.tensor(19.0))), atol=1e-6)
    t = torch.tensor([0.25, 0.75])
    s, ds = O.loglinear_noise(t)
    assert torch.allclose(ds / torch.expm1(s), 1 / t, rtol=1e-5)
    tt = O.sample_t(cfg, 8, torch.Generator().manual_seed(0))
    u = (tt - cfg.sampling_eps) / (1 - cfg.sampling_eps)
    assert torch.all((u >= torch.arange(8) / 8 - 1e-6) & (u < (torch.arange(8) + 1) / 8 + 1e-6))
