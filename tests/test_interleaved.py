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
