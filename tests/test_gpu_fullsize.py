"""Size-independent properties of the hot path at BASELINE.json's FULL sizes on the GPU (pytest -m gpu): the oracle cannot run these sizes in
seconds, so parity here is through identities the domain offers (SURVEY.md §8c analytic KATs, packing equivalence, gradient checksums).

  UniDisc-S : n=12 d=768 H=12 V=32001+8192, B=64, L=128+256                      (BASELINE configs[1])
  1.4 B     : n=24 d=2048 H=16 V=32001+16384, B=8, L=256+1024, 2-D rope          (configs[2])
  1.4 B interleaved: B=1 row of 4 packed samples of 128+1024 tokens, L=4608      (configs[4], bf16 attention)
"""
import importlib.util
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def bench():
    return _bench()


def _build(bench, workload, dropout=0.0, seed=0):
    torch.manual_seed(seed)
    cfg, diff = bench.build(workload, torch.device(DEV), dropout)
    return cfg, diff


@pytest.mark.parametrize("workload,B", [("unidisc-s-l384", 64), ("unidisc-1.4b-l1280", 8), ("unidisc-1.4b-l1280-adaln", 8), ("unidisc-1.4b-interleaved-l4608", 1)])
def test_zero_head_known_answer_and_mask_exactness(bench, workload, B):
    """Zero vocabulary head => uniform prediction over the ids SUBS leaves valid: log p(x0) = -log(Vt - 1) on masked text positions (the
    [MASK] id is excluded), -log(V - Vt) on masked image positions, exactly 0 on unmasked positions; xt == where(move, [MASK], x0) bit for bit;
    the antithetic t of row b lies in stratum [b/B, (b+1)/B)."""
    cfg, diff = _build(bench, workload)
    w = bench.WORKLOADS[workload]
    with torch.no_grad():
        diff.backbone.output_layer.linear.weight.zero_()
        diff.backbone.output_layer.linear.bias.zero_()
    batch = {k: v.to(DEV) for k, v in bench.synthetic_batch(workload, B, 3).items()}
    torch.manual_seed(11)
    out = diff.training_step(batch, 1)
    last = diff._last
    x0 = diff.update_batch({k: v.clone() for k, v in batch.items()})["input_ids"]
    move, xt, lp, t = last["move_indices"], last["xt"], last["log_p_theta"].float(), last["t"].float().cpu()
    assert torch.equal(xt, torch.where(move, torch.full_like(x0, diff.mask_index), x0))
    assert 0 < int(move.sum()) < move.numel()
    Vt, V = diff.text_vocab_size, diff.vocab_size
    is_img = x0 >= Vt
    want = torch.where(is_img, -math.log(V - Vt), -math.log(Vt - 1)).to(lp.dtype)
    assert torch.all(lp[~move] == 0)
    assert torch.allclose(lp[move], want[move], atol=2e-6, rtol=0), float((lp[move] - want[move]).abs().max())
    lo = torch.arange(B, dtype=torch.float32) / B
    eps = float(cfg.trainer.sampling_eps)
    u = (t - eps) / (1 - eps)
    assert torch.all(u >= lo - 1e-6) and torch.all(u <= lo + 1.0 / B + 1e-6)
    assert torch.isfinite(out.loss)


@pytest.mark.parametrize("workload,B", [("unidisc-s-l384", 64), ("unidisc-1.4b-l1280", 8), ("unidisc-1.4b-l1280-adaln", 8)])
def test_gradient_checksums(bench, workload, B):
    """Softmax minus one-hot sums to zero over the vocabulary, so the head-bias gradient sums to zero (a checksum over all masked rows and the
    whole joint vocabulary), is exactly zero at the [MASK] id, and - with modality-restricted SUBS - sums to zero over the text ids and over the
    image ids separately.  All gradients are finite, and the step is reproducible for a fixed seed (Philox dropout included)."""
    res = []
    for rep in range(2):
        cfg, diff = _build(bench, workload, dropout=0.1)
        batch = {k: v.to(DEV) for k, v in bench.synthetic_batch(workload, B, 5).items()}
        torch.manual_seed(21)
        out = diff.training_step(batch, 1)
        out.loss.backward()
        torch.cuda.synchronize()
        gb = diff.backbone.output_layer.linear.bias.grad.double()
        res.append((float(out.loss), gb.clone()))
        if rep == 0:
            Vt = diff.text_vocab_size
            scale = float(gb.abs().sum())
            assert scale > 0
            assert float(gb[diff.mask_index]) == 0.0
            assert abs(float(gb[:Vt].sum())) < 2e-3 * float(gb[:Vt].abs().sum())       # bf16 d-logits: each row's sum is zero to bf16 rounding
            assert abs(float(gb[Vt:].sum())) < 2e-3 * float(gb[Vt:].abs().sum())
            assert all(torch.isfinite(p.grad).all() for p in diff.backbone.parameters() if p.grad is not None)
        del diff
        torch.cuda.empty_cache()
    assert res[0][0] == res[1][0]
    assert rel_err(res[1][1], res[0][1]) < 1e-4        # (fp32 atomics in the column reductions: not bit-reproducible)


def test_batch_rows_are_independent_at_full_size(bench):
    """Sequences never interact (attention stays inside a row): permuting the rows of the batch permutes the per-token log-probabilities."""
    workload, B = "unidisc-1.4b-l1280", 8
    cfg, diff = _build(bench, workload)
    diff.backbone.eval()
    batch = diff.update_batch({k: v.to(DEV) for k, v in bench.synthetic_batch(workload, B, 9).items()})
    x0, mod = batch["input_ids"], batch["modality"]
    g = torch.Generator().manual_seed(1)
    xt = torch.where((torch.rand(x0.shape, generator=g) < 0.5).to(DEV), torch.full_like(x0, diff.mask_index), x0)
    perm = torch.randperm(B, generator=g).to(DEV)
    with torch.no_grad():
        a = diff.backbone.forward_logp(xt, x0, None, modality=mod, restrict_modality=True).float()
        b = diff.backbone.forward_logp(xt[perm], x0[perm], None, modality=mod[perm], restrict_modality=True).float()
    assert rel_err(b, a[perm]) < 2e-3


def test_packed_rows_equal_separate_rows_at_full_size(bench):
    """Configuration E: each of the 4 packed samples gets the logits it gets alone in a row of its own (document mask, per-sample rotary
    positions, image-count embedding), at the full 1.4 B width and L = 4608."""
    workload = "unidisc-1.4b-interleaved-l4608"
    cfg, diff = _build(bench, workload)
    diff.backbone.eval()
    b = bench.synthetic_batch(workload, 1, 13)
    ids, mod, sid = b["input_ids"][0], b["modality"][0], b["sample_ids"][0]
    L, n = ids.numel(), ids.numel() // 4
    rows_i = torch.zeros(4, L, dtype=torch.int64)
    rows_m = torch.zeros(4, L, dtype=torch.int64)
    rows_s = torch.full((4, L), -1, dtype=torch.int64)
    for s in range(4):
        rows_i[s, :n], rows_m[s, :n], rows_s[s, :n] = ids[s * n:(s + 1) * n], mod[s * n:(s + 1) * n], 0
    g = torch.Generator().manual_seed(2)
    masked = torch.rand(L, generator=g) < 0.5
    xt = torch.where(masked, torch.full_like(ids, diff.mask_index), ids)
    rows_x = rows_i.clone()
    for s in range(4):
        rows_x[s, :n] = xt[s * n:(s + 1) * n]
    with torch.no_grad():
        packed = diff.backbone.forward_logp(xt[None].to(DEV), ids[None].to(DEV), None, modality=mod[None].to(DEV), sample_ids=sid[None].to(DEV),
                                            restrict_modality=True).float().cpu()[0]
        sep = diff.backbone.forward_logp(rows_x.to(DEV), rows_i.to(DEV), None, modality=rows_m.to(DEV), sample_ids=rows_s.to(DEV),
                                         restrict_modality=True).float().cpu()
    for s in range(4):
        assert rel_err(sep[s, :n], packed[s * n:(s + 1) * n]) < 1e-2, s
    assert float(packed[masked].abs().sum()) > 0
