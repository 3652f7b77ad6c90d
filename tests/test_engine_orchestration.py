"""CPU check of the hand-scheduled engine's orchestration with kernel test doubles (tests/fake_kernels.py).

The real numerics tests are the `-m gpu` ones; this one catches plumbing mistakes (buffer shapes, adaLN chunk
indices, gradient bookkeeping, callback order) without a GPU by swapping `unidisc_amd.dit.K` for torch stand-ins.
"""
import pytest
import torch

import fake_kernels
from golden_utils import CASE_NAMES, Golden, rel_err
from product_utils import build_product


@pytest.fixture()
def fake_k(monkeypatch):
    import unidisc_amd.dit as dit_mod
    import unidisc_amd.diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    return fake_kernels


@pytest.mark.parametrize("name", CASE_NAMES)
def test_compute_loss_and_grads_match_golden(name, fake_k):
    g = Golden(name)
    diff = build_product(g, device="cpu")
    diff.rng_device = "cpu"
    ready = []
    diff.backbone.grad_ready_callback = lambda flat, lo, hi: ready.append((lo, hi))
    torch.manual_seed(g.case["step_seed"])
    out = diff.training_step(g.batch(), 1)
    assert torch.equal(diff._last["xt"], g.t("fp32/xt"))
    assert torch.equal(diff._last["move_indices"], g.t("fp32/move_indices"))
    assert torch.equal(out.token_mask, g.t("fp32/token_mask"))
    l32, l16 = float(g.t("fp32/loss")), float(g.t("bf16/loss"))
    assert abs(float(out.loss) - l32) <= 3 * abs(l16 - l32) + 5e-3 * abs(l32), (float(out.loss), l32, l16)
    assert torch.allclose(out.nlls, g.t("fp32/nlls"), atol=0.15, rtol=0.05)
    out.loss.backward()
    assert ready[0][0] == 0 and all(a[1] == b[0] for a, b in zip(ready, ready[1:]))  # contiguous, in completion order
    assert ready[-1][1] >= sum(p.numel() for p in diff.backbone.parameters())
    gref = g.grads("fp32")
    gb16 = g.grads("bf16")
    named = dict(diff.backbone.named_parameters())
    assert set(gref) == {k for k, p in named.items() if p.grad is not None}
    for k, gr in gref.items():
        budget = rel_err(gb16[k], gr)
        e = rel_err(named[k].grad, gr)
        assert e <= 3 * budget + 0.03, (k, e, budget)


def test_logits_path_and_state_dict(fake_k):
    g = Golden("c_large")
    diff = build_product(g, device="cpu")
    xt, sigma = g.t("fp32/xt"), None
    with torch.no_grad():
        logits = diff.backbone(xt, sigma, modality=g.t("fp32/modality"))
    assert logits.shape == g.t("fp32/logits").shape and logits.dtype == torch.bfloat16
    truth = g.t("fp32/logits")
    assert rel_err(logits.float(), truth) <= 3 * rel_err(g.t("bf16/logits"), truth) + 5e-3
    with torch.no_grad():
        lp = diff.forward(xt, sigma, batch=None, modality=g.t("fp32/modality"))
    ref = g.t("fp32/log_probs")
    finite = ref > -1e5
    assert torch.equal(lp.float() > -1e5, finite)
    assert torch.allclose(lp.float()[finite], ref[finite], atol=0.1, rtol=0.05)


@pytest.mark.parametrize("name", ["b_small", "c_large", "d_adaln_mm"])
def test_masked_row_head_compaction_is_exact(name, fake_k):
    """The vocabulary head restricted to the [MASK] rows - and, without adaLN, the last block's out-proj / MLP / final norm restricted to them too -
    must give the same loss and the same gradients as the full-row run."""
    g = Golden(name)
    res = []
    splits = []
    for compact, last, split in ((False, False, True), (True, False, True), (True, True, True), (True, True, False)):
        diff = build_product(g, device="cpu")
        diff.rng_device = "cpu"
        diff.backbone.compact_head = compact
        diff.backbone.compact_last_block = last
        diff.backbone.split_head = split     # (text rows x text ids) + (image rows x image ids) instead of rows x all ids: exact under force_argmax_valid_indices
        seen = []
        orig = type(diff.backbone)._engine_backward
        diff.backbone._engine_backward = lambda S, grad, mode, orig=orig, bb=diff.backbone: (seen.append((S["hf"].shape[0], bool(S.get("stream_compact")))), splits.append((compact, split, S.get("head_groups"))), orig(bb, S, grad, mode))[2]
        torch.manual_seed(g.case["step_seed"])
        batch = {k: torch.cat([v] * 4) for k, v in g.batch().items()}   # 512 rows: the padded [MASK] row list is shorter than the batch
        out = diff.training_step(batch, 1)
        out.loss.backward()
        M = 4 * g.case["batch_size"] * (g.case["txt_length"] + g.case["img_length"])
        assert seen and (seen[0][0] < M) == compact and seen[0][1] == (last and not g.case["time_conditioning"]), (seen, M)
        res.append((out.loss.detach().clone(), out.nlls.detach().clone(), {k: p.grad.clone() for k, p in diff.backbone.named_parameters() if p.grad is not None}))
    (l0, n0, g0) = res[0]
    for (l1, n1, g1) in res[1:]:
        assert torch.allclose(l0, l1, rtol=1e-6, atol=1e-7) and torch.allclose(n0, n1, rtol=1e-6, atol=1e-7)
        assert set(g0) == set(g1)
        for k in g0:
            assert torch.allclose(g0[k], g1[k], rtol=1e-5, atol=1e-7), k
    restrict = bool(g.case["force_argmax_valid_indices"]) and g.case["img_length"] > 0
    for compact, split, groups in splits:   # the split really ran where it applies (two non-empty groups, multiples of 64), and only there
        if compact and split and restrict:
            assert groups is not None and all(n % 64 == 0 for n in groups) and sum(groups) > 0, (compact, split, groups)
        else:
            assert groups is None, (compact, split, groups)


@pytest.mark.parametrize("name", ["b_small", "c_large", "d_adaln_mm"])
@pytest.mark.parametrize("compact", [False, True])
def test_chunked_head_and_cross_entropy_is_exact(name, compact, fake_k):
    """model.head_chunk_rows: head + SUBS cross-entropy over row chunks (logits recomputed per chunk in the backward, wgrad accumulated) gives
    the loss and the gradients of the one-piece head."""
    g = Golden(name)
    res = []
    for chunk in (0, 64):
        diff = build_product(g, device="cpu")
        diff.rng_device = "cpu"
        diff.backbone.compact_head = compact
        diff.backbone.head_chunk_rows = chunk
        torch.manual_seed(g.case["step_seed"])
        batch = {k: torch.cat([v] * 4) for k, v in g.batch().items()}   # 512 rows: several chunks also when the head is compacted
        out = diff.training_step(batch, 1)
        out.loss.backward()
        res.append((out.loss.detach().clone(), out.nlls.detach().clone(), {k: p.grad.clone() for k, p in diff.backbone.named_parameters() if p.grad is not None}))
    (l0, n0, g0), (l1, n1, g1) = res
    assert torch.allclose(l0, l1, rtol=1e-6, atol=1e-7) and torch.allclose(n0, n1, rtol=1e-6, atol=1e-7)
    for k in g0:
        assert torch.allclose(g0[k], g1[k], rtol=1e-5, atol=1e-7), k


def test_val_and_test_prefixes_update_attached_metrics(fake_k):
    """model.py:1163-1171: prefix 'val' / 'test' feed (nlls, token_mask) - and the per-modality pairs - to the metric collections on the trainer."""
    class Mean:
        def __init__(self):
            self.s, self.n = 0.0, 0.0

        def update(self, value, weight):
            self.s += float((value * weight).sum())
            self.n += float(weight.sum())

    g = Golden("c_large")
    diff = build_product(g, device="cpu")
    diff.rng_device = "cpu"
    with pytest.raises(RuntimeError):
        diff.compute_loss(diff.update_batch(g.batch()), prefix="val")
    diff.valid_metrics, diff.valid_txt_metrics, diff.valid_img_metrics, diff.test_metrics = Mean(), Mean(), Mean(), Mean()
    torch.manual_seed(g.case["step_seed"])
    ref = diff.training_step(g.batch(), 1)
    torch.manual_seed(g.case["step_seed"])
    with torch.no_grad():
        assert diff.compute_loss(diff.update_batch(g.batch()), prefix="val") is None
    assert diff.valid_metrics.n == float(ref.token_mask.sum()) and abs(diff.valid_metrics.s - float((ref.nlls * ref.token_mask).sum())) < 1e-3
    assert abs(diff.valid_txt_metrics.s + diff.valid_img_metrics.s - diff.valid_metrics.s) < 1e-3
    torch.manual_seed(g.case["step_seed"])
    with torch.no_grad():
        diff.compute_loss(diff.update_batch(g.batch()), prefix="test")
    assert diff.test_metrics.n == diff.valid_metrics.n
    with pytest.raises(ValueError):
        diff.compute_loss(diff.update_batch(g.batch()), prefix="bogus")


def test_zero_modality_loss_weight_is_kept(fake_k):
    """trainer.text_loss_weight = 0.0 is an image-only objective (model.py:1041-1044 multiplies by the configured value); an `or 1.0` default once turned it into 1.0."""
    g = Golden("c_large")
    losses = {}
    for tw, iw in ((0.0, 0.5), (1.0, 0.5), (1.0, 0.0)):
        case = dict(g.case, text_loss_weight=tw, img_loss_weight=iw)
        from product_utils import product_config
        from unidisc_amd import Diffusion
        diff = Diffusion(product_config(case), None, "cpu")
        diff.backbone.load_state_dict(g.params(), strict=True)
        diff.backbone.train()
        diff.rng_device = "cpu"
        torch.manual_seed(g.case["step_seed"])
        out = diff.training_step(g.batch(), 1)
        losses[(tw, iw)] = (float(out.loss.detach()), float(out.txt_loss), float(out.img_loss))
    full, img_only, txt_only = losses[(1.0, 0.5)], losses[(0.0, 0.5)], losses[(1.0, 0.0)]
    assert img_only[1] == 0.0 and abs(img_only[0] - full[2]) <= 1e-6 * abs(full[2]) and img_only[0] != full[0]
    assert txt_only[2] == 0.0 and abs(txt_only[0] - full[1]) <= 1e-6 * abs(full[1])


def test_key_padding_mask_use_attention_mask(fake_k):
    """`model.use_attention_mask` (model.py:405-406 -> sdpa(attn_mask=attention_mask), models/dit.py:829): the batch's padding mask hides padded KEYS from every
    query of the sample.  Logits against the oracle with the same dense allow-mask; padded keys really are invisible (changing their tokens changes nothing at
    the valid positions); combined with modality attention dropout both masks apply."""
    from oracle import unidisc_oracle as O
    from unidisc_amd import ModalityMask

    g = Golden("c_large")
    diff = build_product(g, device="cpu")
    diff.backbone.eval()
    xt, mod = g.t("fp32/xt"), g.t("fp32/modality")
    B, L = xt.shape
    km = torch.ones(B, L, dtype=torch.bool)
    km[0, 5:9] = False
    km[1, L - 7:] = False
    P, buf = g.params(), g.buffers()
    allow = km[:, None, :].expand(B, L, L)
    with torch.no_grad():
        got = diff.backbone(xt, None, modality=mod, attention_mask=km).float()
        ref = O.dit_forward(g.cfg, P, buf, xt, None, mod, None, False, allow_mask=allow)
        base = diff.backbone(xt, None, modality=mod).float()
        x2 = xt.clone()
        x2[0, 5:9] = (x2[0, 5:9] + 3) % 17
        got2 = diff.backbone(x2, None, modality=mod, attention_mask=km).float()
    assert rel_err(got, ref) <= 3 * rel_err(g.t("bf16/logits"), g.t("fp32/logits")) + 5e-3
    assert rel_err(base, ref) > 10 * rel_err(got, ref)            # the mask matters
    valid0 = km[0]
    assert rel_err(got2[0][valid0], got[0][valid0]) < 1e-6 and torch.equal(got2[1], got[1])   # padded keys are invisible to the valid positions
    # together with modality attention dropout: both restrictions hold
    drop = ModalityMask(torch.tensor([True] + [False] * (B - 1)), torch.zeros(B, dtype=torch.bool), g.case["txt_length"])
    with torch.no_grad():
        both = diff.backbone(xt, None, modality=mod, attention_mask=km, block_mask=drop).float()
    is_txt = torch.arange(L) < g.case["txt_length"]
    allow2 = allow.clone()
    allow2[0] &= ~(is_txt[:, None] & ~is_txt[None, :])            # sample 0: text queries see text keys only
    with torch.no_grad():
        ref2 = O.dit_forward(g.cfg, P, buf, xt, None, mod, None, False, allow_mask=allow2)
    assert rel_err(both, ref2) <= 3 * rel_err(g.t("bf16/logits"), g.t("fp32/logits")) + 5e-3


@pytest.mark.parametrize("name", ["c_large", "d_adaln_mm", "f_interleaved"])
def test_gradient_checkpointing_recomputes_blocks_bit_identically(name, fake_k):
    """trainer.use_gradient_checkpointing (models/dit.py:1486-1490): only each block's input is kept, the block is re-run right before its backward.  Same loss
    and bit-identical gradients as the activation-keeping engine."""
    g = Golden(name)
    res = []
    for ck in (False, True):
        diff = build_product(g, device="cpu")
        diff.rng_device = "cpu"
        diff.backbone.use_gradient_checkpointing = ck
        torch.manual_seed(g.case["step_seed"])
        out = diff.training_step(g.batch(), 1)
        out.loss.backward()
        res.append((float(out.loss.detach()), {k: p.grad.clone() for k, p in diff.backbone.named_parameters() if p.grad is not None}))
    assert res[0][0] == res[1][0]
    assert set(res[0][1]) == set(res[1][1])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_low_precision_loss_is_accepted_and_changes_nothing(fake_k):
    """`trainer.low_precision_loss` (model.py:747, :924): under bf16 autocast the reference's loss is the SAME number with the flag on or off (checked against the
    imported reference on the c_large case when this test was written: loss 3.809645652770996 and the NLL sum identical to the last bit) - the product accepts the flag
    instead of raising, and computes the same loss."""
    g = Golden("c_large")
    vals = []
    for flag in (False, True):
        diff = build_product(g, device="cpu")
        diff.config.trainer.low_precision_loss = flag
        diff.rng_device = "cpu"
        from unidisc_amd import Diffusion

        diff2 = Diffusion(diff.config, None, "cpu", backbone=diff.backbone)   # (init re-reads the trainer flags: must not raise)
        diff2.rng_device = "cpu"
        torch.manual_seed(g.case["step_seed"])
        out = diff2.training_step(g.batch(), 1)
        vals.append((float(out.loss.detach()), out.nlls.detach().clone()))
    assert vals[0][0] == vals[1][0] and torch.equal(vals[0][1], vals[1][1])
