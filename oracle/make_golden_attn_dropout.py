"""Golden vectors for modality attention dropout (`model.flex_attention_{txt,img}_masking_prob`, shipped in
configs/experiments/small_scale_train_caching.yaml:34-35), made by running the IMPORTED reference (build container only).

    python -m oracle.make_golden_attn_dropout        # writes tests/golden/g_attn_dropout.npz

TEST INFRASTRUCTURE (same status as make_golden.py).  Reference code exercised (file:line in /root/reference): the two per-sample draws and the
`& ~should_mask_*` rule in `compute_loss` model.py:863-875, `get_block_mask` / `_attn_mask` model_utils.py:721-737, and the FlexAttention call of
`Attention.forward` models/dit.py:784-812 with that block mask.

One substitution, the same as in make_golden_interleaved.py: FlexAttention (un-vendored torch, no CPU backward) is replaced by its definition -
SDPA with the dense boolean mask `mask_mod` describes, evaluated by calling the reference's own `_attn_mask` on every (b, q, kv).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_shim
from oracle import make_golden as MG
from oracle.cases import ATTN_DROPOUT_CASES, lumina_rope_2d


def main():
    case = ATTN_DROPOUT_CASES["g_attn_dropout"]
    ref_shim.install()
    ref_shim.install_lumina_rope(lumina_rope_2d)
    import models.dit as refdit
    import model_utils as ref_utils

    class DenseMask:
        def __init__(self, m):
            self.m = m

    def create_block_mask(mask_mod, B, H, Q_LEN, KV_LEN, device=None, **kw):
        b = torch.arange(B)[:, None, None]
        q = torch.arange(Q_LEN)[None, :, None]
        kv = torch.arange(KV_LEN)[None, None, :]
        return DenseMask(mask_mod(b, None, q, kv).expand(B, Q_LEN, KV_LEN)[:, None])

    def flex_attention(q, k, v, block_mask=None, **kw):
        if block_mask is None:
            return F.scaled_dot_product_attention(q, k, v)
        return F.scaled_dot_product_attention(q, k, v, attn_mask=block_mask.m)

    refdit.flex_attention = refdit.compiled_flex_attention = flex_attention
    refdit.create_block_mask = create_block_mask
    import torch.nn.attention.flex_attention as fa
    fa.create_block_mask = create_block_mask
    ref_utils.create_block_mask = create_block_mask

    orig_cfg = MG._ref_cfg

    def cfg_with_dropout(c):
        cfg = orig_cfg(c)
        cfg.model.use_flex_attention = True
        cfg.model.flex_attention_txt_masking_prob = c["flex_attention_txt_masking_prob"]
        cfg.model.flex_attention_img_masking_prob = c["flex_attention_img_masking_prob"]
        return cfg

    MG._ref_cfg = cfg_with_dropout
    drawn = []
    orig_get = ref_utils.get_block_mask
    import model as refmodel

    def get_block_mask(txt_drop, img_drop, *a, **k):
        drawn.append((txt_drop.clone(), img_drop.clone()))
        return orig_get(txt_drop, img_drop, *a, **k)

    ref_utils.get_block_mask = get_block_mask
    if hasattr(refmodel, "get_block_mask"):
        refmodel.get_block_mask = get_block_mask
    out = {}
    batch, rec, params, grads, bufs = MG.run_reference(case, torch.float32)
    assert len(drawn) == 1, "the block mask was not built"
    rec["txt_attn_dropout"], rec["img_attn_dropout"] = drawn[0]
    for k, v in batch.items():
        out["batch/" + k] = MG._np(v)
    for k, v in params.items():
        out["param/" + k] = MG._np(v)
    for k, v in bufs.items():
        out["buffer/" + k] = MG._np(v)
    for k, v in rec.items():
        out[f"fp32/{k}"] = np.array(v) if isinstance(v, str) else MG._np(v)
    for k, v in grads.items():
        out[f"fp32/grad/{k}"] = MG._np(v)
    path = os.path.join(MG.GOLDEN_DIR, "g_attn_dropout.npz")
    np.savez_compressed(path, **out)
    print("loss", float(rec["loss"]), "txt_drop", drawn[0][0].tolist(), "img_drop", drawn[0][1].tolist(), "should_mask_txt", rec["should_mask_txt"].flatten().tolist(),
          "should_mask_img", rec["should_mask_img"].flatten().tolist(), "->", path, f"({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
