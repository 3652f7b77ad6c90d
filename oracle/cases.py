"""Tiny hot-path configurations shared by the golden generator and the tests (TEST INFRASTRUCTURE).

Each case is a flat dict of the flags the hot path reads, at
dimensions small enough that the imported reference, the oracle and the fixtures stay small.
"""
from __future__ import annotations

import torch


def lumina_rope_2d(embed_dim, len_h, len_w, linear_factor=1.0, ntk_factor=1.0):
    """Restatement of diffusers==0.32.2 ``get_2d_rotary_pos_embed_lumina`` (call site models/dit.py:1052-1060).

    PARITY UNPINNED: diffusers is an un-vendored third-party dependency (pyproject.toml:23) and is not
    installed here, so this is a restatement of its published algorithm.  The product takes rotary
    tables as data, so kernel parity does not depend on it.
    Returns complex64 [len_h, len_w, embed_dim // 2] with the last dim interleaved [h0, w0, h1, w1, ...].
    """
    assert embed_dim % 4 == 0
    half = embed_dim // 2
    theta = 10000.0 * ntk_factor
    freqs = 1.0 / (theta ** (torch.arange(0, half, 2, dtype=torch.float32)[: half // 2] / half)) / linear_factor
    fh = torch.outer(torch.arange(len_h, dtype=torch.float32), freqs)
    fw = torch.outer(torch.arange(len_w, dtype=torch.float32), freqs)
    eh = torch.polar(torch.ones_like(fh), fh).view(len_h, 1, half // 2, 1).repeat(1, len_w, 1, 1)
    ew = torch.polar(torch.ones_like(fw), fw).view(1, len_w, half // 2, 1).repeat(len_h, 1, 1, 1)
    return torch.cat([eh, ew], dim=-1).flatten(2)


_BASE = dict(
    hidden_size=64, n_heads=2, cond_dim=32, n_blocks=2, batch_size=4,
    txt_length=16, img_length=16, text_vocab_size=41, vocab_size=65,
    norm_type="rms", qk_norm=True, sandwich_normalization=True, modality_embed=True, rope_2d=False,
    time_conditioning=False, multimodal_batches=True, force_argmax_valid_indices=True,
    mask_entire_modality=None, softmin_snr=None, text_loss_weight=None, img_loss_weight=None,
    force_full_attention_mask_loss_only=None, force_full_attention_mask=None, set_max_txt_loss_ratio=None,
    param_seed=1234, data_seed=99, step_seed=7, ragged_text=False,
)


def _case(**kw):
    c = dict(_BASE)
    c.update(kw)
    return c


CASES = {
    # BASELINE config A flavour: text-only, MDLM-style block (LayerNorm, adaLN-Zero time conditioning), masked mean loss.
    "a_text_adaln": _case(
        img_length=0, txt_length=32, vocab_size=41, norm_type="layernorm", qk_norm=False, sandwich_normalization=False,
        modality_embed=False, time_conditioning=True, multimodal_batches=False, force_argmax_valid_indices=False,
        ragged_text=True,
    ),
    # UniDisc-S flavour (small_scale_train.yaml): masked-mean loss branch, soft-min SNR, whole-modality masking, 1-D rope.
    "b_small": _case(
        mask_entire_modality=0.6, softmin_snr=5, text_loss_weight=1.0, img_loss_weight=None,
        force_full_attention_mask_loss_only=True, ragged_text=True,
    ),
    # 1.4B flavour (large_scale_train[_high_res].yaml): 2-D rope on image rows, modality-weighted loss, full attention mask.
    "c_large": _case(
        rope_2d=True, linear_factor=2.0, mask_entire_modality=0.6, softmin_snr=5, text_loss_weight=1.0,
        img_loss_weight=0.5, force_full_attention_mask=True, step_seed=11,
    ),
    # north-star adaLN variant on multimodal batches: modulation/gating on image tokens only (dit.py:239-251,266-268).
    "d_adaln_mm": _case(
        time_conditioning=True, sandwich_normalization=False, text_loss_weight=1.0, img_loss_weight=0.6, step_seed=13,
    ),
    # sandwich + time conditioning (gate_msa unused quirk, dit.py:983) + set_max_txt_loss_ratio clamp.
    "e_adaln_sandwich": _case(
        time_conditioning=True, sandwich_normalization=True, text_loss_weight=1.0, img_loss_weight=0.2,
        set_max_txt_loss_ratio=1.0, mask_entire_modality=0.4, step_seed=17,
    ),
}


# Interleaved / packed batches (fixture made by oracle/make_golden_interleaved.py).  Kept out of CASES: the generic suites
# iterate over CASES with token-dataset batches, this one has its own batch layout (sample ids, padding, two images in one sample).
INTERLEAVED_CASES = {
    "f_interleaved": _case(
        txt_length=48, img_length=512, rope_2d=True, linear_factor=1.0, mask_entire_modality=0.6, softmin_snr=5, text_loss_weight=1.0, img_loss_weight=0.5,
        batch_size=2, step_seed=23, data_seed=101, interleaved=True,
        # rows of (sample id, modality, length); -1 = padding
        layout=[[(0, 0, 12), (0, 1, 256), (0, 0, 8), (1, 0, 10), (1, 1, 256), (1, 0, 6), (-1, -1, 12)],
                [(0, 0, 16), (0, 1, 256), (0, 0, 8), (0, 1, 256), (0, 0, 12), (-1, -1, 12)]],
    ),
}


# Modality attention dropout (model.flex_attention_{txt,img}_masking_prob; fixture made by oracle/make_golden_attn_dropout.py): the 1.4 B flavour with
# eight samples so the per-sample draws cover all four combinations, whole-modality masking on (its `& ~should_mask_*` interaction is part of the rule).
ATTN_DROPOUT_CASES = {
    "g_attn_dropout": _case(
        rope_2d=True, linear_factor=2.0, mask_entire_modality=0.6, softmin_snr=5, text_loss_weight=1.0, img_loss_weight=0.5, force_full_attention_mask=True,
        batch_size=8, step_seed=29, data_seed=103, flex_attention_txt_masking_prob=0.5, flex_attention_img_masking_prob=0.5,
    ),
}
