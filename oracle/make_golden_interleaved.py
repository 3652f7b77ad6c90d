"""Golden vectors for interleaved / packed batches, made by running the IMPORTED reference (build container only).

    python -m oracle.make_golden_interleaved        # writes tests/golden/f_interleaved.npz

TEST INFRASTRUCTURE (same status as make_golden.py).  Reference code exercised (file:line in /root/reference): `update_batch` interleaved tail
model.py:350-393; `q_xt` per-block modality masking :483-522; `compute_loss` with the document mask :876-878; `DIT.forward` interleaved rotary
dit.py:1421-1444 with `add_img_data_to_blocks` / `add_txt_data_to_blocks` :122-191 (per-image-block 2-D RoPE, image-count embedding, text
positions restarting at every packed sample) and `get_interleaved_block_mask` model_utils.py:740-771.

One substitution, stated here: FlexAttention (`torch.nn.attention.flex_attention`, un-vendored torch) is replaced by its definition - SDPA with the
dense boolean mask that `mask_mod` describes, rows without any allowed key returning zeros - because its CPU path has no backward.  The mask
itself is built by calling the reference's `_interleaved_attn_mask` on every (b, q, kv).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_shim
from oracle import make_golden as MG
from oracle.cases import CASES, lumina_rope_2d

from oracle.cases import INTERLEAVED_CASES

CASE = INTERLEAVED_CASES["f_interleaved"]
LAYOUT = CASE["layout"]


def make_batch(case):
    g = torch.Generator().manual_seed(case["data_seed"])
    Vt, V = case["text_vocab_size"], case["vocab_size"]
    L = case["txt_length"] + case["img_length"]
    ids, mod, sid, am = [], [], [], []
    for row in LAYOUT:
        i, m, s, a = [], [], [], []
        for (sample, modality, n) in row:
            if modality == 1:
                i.append(torch.randint(Vt, V, (n,), generator=g))
            elif modality == 0:
                i.append(torch.randint(0, Vt - 1, (n,), generator=g))
            else:
                i.append(torch.zeros(n, dtype=torch.int64))
            m.append(torch.full((n,), modality, dtype=torch.int64))
            s.append(torch.full((n,), sample, dtype=torch.int64))
            a.append(torch.full((n,), modality >= 0, dtype=torch.bool))
        ids.append(torch.cat(i)); mod.append(torch.cat(m)); sid.append(torch.cat(s)); am.append(torch.cat(a))
        assert ids[-1].numel() == L, ids[-1].numel()
    return dict(input_ids=torch.stack(ids), modality=torch.stack(mod), sample_ids=torch.stack(sid), attention_mask=torch.stack(am))


def main():
    case = CASE
    ref_shim.install()
    ref_shim.install_lumina_rope(lumina_rope_2d)
    import models.dit as refdit
    import model_utils as ref_utils
    import model as refmodel

    # --- FlexAttention -> dense-mask SDPA (see module docstring)
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
        m = block_mask.m
        any_key = m.any(-1, keepdim=True)
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=m | ~any_key)   # rows without keys: computed unmasked, then zeroed
        return torch.where(any_key, out, torch.zeros_like(out))

    refdit.flex_attention = refdit.compiled_flex_attention = flex_attention
    refdit.create_block_mask = create_block_mask
    import torch.nn.attention.flex_attention as fa
    fa.create_block_mask = create_block_mask
    ref_utils.create_block_mask = create_block_mask

    orig_cfg = MG._ref_cfg

    def cfg_with_interleaved(c):
        cfg = orig_cfg(c)
        cfg.model.use_flex_attention = True
        cfg.trainer.interleaved = True
        cfg.trainer.interleaved_training_flex_attention = True
        cfg.data.require_sample_ids = True
        return cfg

    MG._ref_cfg = cfg_with_interleaved
    MG.make_batch = make_batch
    out = {}
    for tag, dtype in (("fp32", torch.float32),):
        batch, rec, params, grads, bufs = MG.run_reference(case, dtype)
        for k, v in batch.items():
            out["batch/" + k] = MG._np(v)
        for k, v in params.items():
            out["param/" + k] = MG._np(v)
        for k, v in bufs.items():
            out["buffer/" + k] = MG._np(v)
        for k, v in rec.items():
            out[f"{tag}/{k}"] = np.array(v) if isinstance(v, str) else MG._np(v)
        for k, v in grads.items():
            out[f"{tag}/grad/{k}"] = MG._np(v)
    path = os.path.join(MG.GOLDEN_DIR, "f_interleaved.npz")
    np.savez_compressed(path, **out)
    print("loss", float(rec["loss"]), "masked", int(rec["move_indices"].sum()), "->", path, f"({os.path.getsize(path) / 1024:.0f} KiB)")
    print("grad of img_count_embedding rows touched:", (grads["img_count_embedding"].abs().sum(-1) > 0).nonzero().flatten().tolist())


if __name__ == "__main__":
    main()
