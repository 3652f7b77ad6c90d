"""Minimal config objects with the reference's Hydra tree shape (config.model.*, config.trainer.*, config.data.*).

The reference builds its config with Hydra/OmegaConf (configs/config.yaml + experiments); the hot path only
reads the keys listed in SURVEY.md §5.6 and reads most of them through ``getattr(cfg, key, default)``.
``Cfg`` is an attribute bag with that behaviour, so either an OmegaConf node or a ``Cfg`` works everywhere.
"""
from __future__ import annotations


class Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __getattr__(self, k):  # missing keys raise AttributeError so getattr(cfg, k, default) works
        raise AttributeError(k)

    def __contains__(self, k):
        return k in self.__dict__

    def get(self, k, d=None):
        return self.__dict__.get(k, d)

    def __repr__(self):
        return f"Cfg({self.__dict__})"


# model sizes: configs/model/small.yaml (UniDisc-S) and configs/model/extra_large.yaml (1.4 B) of the reference
MODEL_PRESETS = {
    "tiny": dict(hidden_size=64, n_heads=2, cond_dim=32, n_blocks=2),
    "plumbing": dict(hidden_size=256, n_heads=4, cond_dim=128, n_blocks=2),          # BASELINE configs[0]
    "small": dict(hidden_size=768, n_heads=12, cond_dim=128, n_blocks=12),           # UniDisc-S, ~115 M non-embedding
    "extra_large": dict(hidden_size=2048, n_heads=16, cond_dim=128, n_blocks=24),    # UniDisc 1.4 B
}


def make_config(*, hidden_size, n_heads, cond_dim, n_blocks, txt_length, img_length, norm_type="rms", qk_norm=True, sandwich_normalization=True,
                modality_embed=True, rope_2d=False, linear_factor=1.0, time_conditioning=False, multimodal_batches=True,
                force_argmax_valid_indices=True, dropout=0.0, zero_linear_init=False, image_vocab_size=None, precision="bf16", **trainer_kw):
    """Build a config tree with the hot-path keys (everything else at the reference's defaults)."""
    model = Cfg(hidden_size=hidden_size, n_heads=n_heads, cond_dim=cond_dim, n_blocks=n_blocks, dropout=dropout, length=txt_length + img_length,
                txt_length=txt_length, img_length=img_length, attn_type="flash", force_varlen_attn=False, norm_type=norm_type, qk_norm=qk_norm,
                sandwich_normalization=sandwich_normalization, full_attention=True, modality_embed=modality_embed, rope_2d=rope_2d,
                linear_factor=linear_factor, zero_linear_init=zero_linear_init, scale_by_sigma=False, use_attention_mask=False,
                force_argmax_valid_indices=force_argmax_valid_indices, image_model=img_length > 0, unified_model=img_length > 0,
                image_vocab_size=image_vocab_size)
    trainer = Cfg(precision=precision, image_mode="discrete", multimodal_batches=multimodal_batches, interleaved=False, antithetic_sampling=True,
                  importance_sampling=False, change_of_variables=False, sampling_eps=1e-3, allow_null_sigma=True,
                  log_seperate_modal_losses=img_length > 0, **{k: v for k, v in trainer_kw.items() if v is not None})
    data = Cfg(require_sample_ids=False, txt_only=False)
    return Cfg(model=model, trainer=trainer, data=data, eval=Cfg(), noise=Cfg(type="loglinear"), time_conditioning=time_conditioning,
               parameterization="subs", backbone="dit", mode="train", T=0)
