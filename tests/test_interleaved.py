"""Interleaved / packed batches: the oracle's restatement and the product against the golden fixture recorded from the
imported reference (oracle/make_golden_interleaved.py): two rows, three packed samples, two images inside one sample, tail padding."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_kernels  # noqa: E402
from golden_utils import Golden, rel_err  # noqa: E402
from oracle import unidisc_oracle as O  # noqa: E402

NAME = "f_interleaved"


def test_oracle_interleaved_matches_reference():
    g = Golden(NAME)
    cfg = g.cfg
    assert cfg.interleaved
    P, buf = g.params(True), g.buffers()
    batch = O.update_batch(cfg, g.batch())
    assert torch.equal(batch["sample_ids"], g.t("batch/sample_ids")) and torch.equal(batch["attention_mask"], g.t("fp32/attention_mask"))
    out = O.compute_loss(cfg, P, buf, batch, generator=g.generator())
    a = out.aux
    assert torch.equal(a["move_indices"], g.t("fp32/move_indices")) and torch.equal(a["xt"], g.t("fp32/xt"))     # integer / mask work: bit-exact
    assert torch.equal(a["ignore_batch_mask"].reshape(-1), g.t("fp32/ignore_batch_mask").reshape(-1).bool())
    assert rel_err(a["logits"], g.t("fp32/logits")) < 5e-6 and rel_err(a["log_probs"], g.t("fp32/log_probs")) < 1e-6
    assert abs(float(out.loss) - float(g.t("fp32/loss"))) < 1e-6 * abs(float(g.t("fp32/loss")))
    out.loss.backward()
    for k, v in g.grads("fp32").items():
        assert rel_err(P[k].grad, v) < 2e-5, k
    # the image-count embedding is used for image 0 and image 1 of a sample, never beyond
    gc = P["img_count_embedding"].grad
    assert (gc[:2].abs().sum(-1) > 0).all() and (gc[2:] == 0).all()


def test_oracle_interleaved_rotary_layout():
    g = Golden(NAME)
    cfg, P, buf = g.cfg, g.params(), g.buffers()
    b = O.update_batch(cfg, g.batch())
    x = torch.zeros(*b["input_ids"].shape, cfg.hidden_size)
    x2, cos, sin = O.interleaved_rotary(cfg, P, buf, x, b["modality"], b["sample_ids"])
    lay = g.case["layout"]
    # row 0: text of sample 1 restarts at position 0 of the 1-D table; its image gets the 256-token 2-D table and count embedding 0
    s1 = 12 + 256 + 8
    assert torch.equal(cos[0, s1:s1 + 10], buf["rotary_cos_emb_txt"][:10])
    assert torch.equal(cos[0, s1 + 10:s1 + 266], buf["rotary_cos_emb_img_256"]) and torch.equal(x2[0, s1 + 10], P["img_count_embedding"][0])
    # row 1: the second image of the same sample gets count embedding 1; text after it continues the sample's 1-D positions
    i2 = 16 + 256 + 8
    assert torch.equal(x2[1, i2], P["img_count_embedding"][1]) and torch.equal(x2[1, 16], P["img_count_embedding"][0])
    assert torch.equal(cos[1, i2 + 256:i2 + 268], buf["rotary_cos_emb_txt"][i2 + 256:i2 + 268])
    # padding: zero tables (q, k rotate to zero), no embedding
    assert (cos[:, -12:] == 0).all() and (sin[:, -12:] == 0).all() and (x2[:, -12:] == 0).all()
    assert sum(n for row in lay for (_, _, n) in row) == 2 * cfg.length


# ------------------------------------------------------------------------------------------------ product (host logic here, HIP kernels under -m gpu)
def _product_step(g, device):
    from product_utils import build_product

    diff = build_product(g, device)
    diff.rng_device = "cpu"   # replay the reference's CPU generator stream
    torch.manual_seed(g.case["step_seed"])
    out = diff.training_step(g.batch(), 1)
    assert torch.equal(diff._last["xt"].cpu(), g.t("fp32/xt")) and torch.equal(diff._last["move_indices"].cpu(), g.t("fp32/move_indices"))   # bit-exact
    assert torch.equal(out.token_mask.cpu(), g.t("fp32/token_mask"))
    return diff, out


def _check_step(g, diff, out, loss_tol, nll_tol, grad_tol):
    l32 = float(g.t("fp32/loss"))
    assert abs(float(out.loss) - l32) <= loss_tol * abs(l32), (float(out.loss), l32)
    assert rel_err(out.nlls.cpu(), g.t("fp32/nlls")) < nll_tol
    assert torch.all(out.nlls.cpu()[~g.t("fp32/move_indices")] == 0)
    out.loss.backward()
    named = dict(diff.backbone.named_parameters())
    gref = g.grads("fp32")
    assert set(gref) == {k for k, p in named.items() if p.grad is not None}
    bad = [(k, rel_err(named[k].grad.cpu(), v)) for k, v in gref.items() if rel_err(named[k].grad.cpu(), v) > grad_tol]
    assert not bad, bad[:6]


def test_product_rotary_layout_matches_oracle():
    """`DIT._rotary_interleaved` (tensor form) against the oracle's block loop: tables, and the image-count embedding row of every position."""
    from product_utils import build_product

    g = Golden(NAME)
    diff = build_product(g, "cpu")
    b = O.update_batch(g.cfg, g.batch())
    P, buf = g.params(), g.buffers()
    x = torch.zeros(*b["input_ids"].shape, g.cfg.hidden_size)
    x2, cos, sin = O.interleaved_rotary(g.cfg, P, buf, x, b["modality"], b["sample_ids"])
    c, s, cnt = diff.backbone._rotary_interleaved(b["modality"], b["sample_ids"])
    assert torch.equal(c, cos) and torch.equal(s, sin)
    want = torch.where(cnt[..., None] >= 0, P["img_count_embedding"][cnt.clamp(min=0)], torch.zeros(()))
    assert torch.equal(want, x2)


def test_product_interleaved_host_logic_with_kernel_doubles(monkeypatch):
    import unidisc_amd.dit as dit_mod
    import unidisc_amd.diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g = Golden(NAME)
    diff, out = _product_step(g, "cpu")
    _check_step(g, diff, out, loss_tol=1e-2, nll_tol=3e-2, grad_tol=8e-2)   # the doubles round to bf16 where the kernels do


@pytest.mark.gpu
def test_product_interleaved_training_step_gpu():
    """HIP path on packed samples: document-masked attention from sample ids, per-block rotary tables, image-count embedding and its gradient."""
    g = Golden(NAME)
    diff, out = _product_step(g, "cuda")
    _check_step(g, diff, out, loss_tol=4e-3, nll_tol=1e-2, grad_tol=6e-2)
    gc = dict(diff.backbone.named_parameters())["img_count_embedding"].grad.cpu()
    assert (gc[2:] == 0).all()


@pytest.mark.gpu
def test_packed_samples_equal_separate_rows_gpu():
    """Size-independent property of the document mask + per-sample rotary positions: a sample gets the same logits packed behind another sample
    as alone in its own row (up to bf16 summation order: its key tiles start at a different offset)."""
    from product_utils import build_product

    g = Golden(NAME)
    diff = build_product(g, "cuda")
    diff.backbone.eval()
    b = O.update_batch(g.cfg, g.batch())
    ids, mod, sid = b["input_ids"][0], b["modality"][0], b["sample_ids"][0]
    L = ids.numel()
    n0, n1 = int((sid == 0).sum()), int((sid == 1).sum())
    assert n0 > 0 and n1 > 0 and bool((sid[:n0] == 0).all()) and bool((sid[n0:n0 + n1] == 1).all())

    def alone(lo, n):
        i, m, s_ = torch.zeros(L, dtype=torch.int64), torch.zeros(L, dtype=torch.int64), torch.full((L,), -1, dtype=torch.int64)
        i[:n], m[:n], s_[:n] = ids[lo:lo + n], mod[lo:lo + n], 0
        return i, m, s_

    rows = [alone(0, n0), alone(n0, n1)]
    with torch.no_grad():
        packed = diff.backbone(ids[None].cuda(), None, modality=mod[None].cuda(), sample_ids=sid[None].cuda()).float().cpu()[0]
        sep = diff.backbone(torch.stack([r[0] for r in rows]).cuda(), None, modality=torch.stack([r[1] for r in rows]).cuda(),
                            sample_ids=torch.stack([r[2] for r in rows]).cuda()).float().cpu()
    assert rel_err(sep[0, :n0], packed[:n0]) < 1e-2
    assert rel_err(sep[1, :n1], packed[n0:n0 + n1]) < 1e-2
