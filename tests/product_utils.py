"""Build the product objects (unidisc_amd.DIT / Diffusion) for a golden case."""
import torch

from unidisc_amd import DIT, Diffusion, make_config


def product_config(case):
    kw = dict(hidden_size=case["hidden_size"], n_heads=case["n_heads"], cond_dim=case["cond_dim"], n_blocks=case["n_blocks"],
              txt_length=case["txt_length"], img_length=case["img_length"], norm_type=case["norm_type"], qk_norm=case["qk_norm"],
              sandwich_normalization=case["sandwich_normalization"], modality_embed=case["modality_embed"], rope_2d=case["rope_2d"],
              linear_factor=case.get("linear_factor", 1.0), time_conditioning=case["time_conditioning"], multimodal_batches=case["multimodal_batches"],
              force_argmax_valid_indices=case["force_argmax_valid_indices"], dropout=0.0,
              image_vocab_size=case["vocab_size"] - case["text_vocab_size"] if case["img_length"] > 0 else None)
    for k in ("mask_entire_modality", "softmin_snr", "text_loss_weight", "img_loss_weight", "force_full_attention_mask_loss_only",
              "force_full_attention_mask", "set_max_txt_loss_ratio"):
        kw[k] = case.get(k)
    cfg = make_config(**kw)
    cfg.model.force_text_vocab_size = case["text_vocab_size"] - 1  # tokenizer-less: len(tokenizer) stand-in (model_setup.py:90-92)
    for k in ("flex_attention_txt_masking_prob", "flex_attention_img_masking_prob"):   # modality attention dropout
        if case.get(k) is not None:
            setattr(cfg.model, k, case[k])
    if case.get("interleaved"):  # configs of the interleaved checkpoints: packed samples, document mask from sample ids
        cfg.trainer.interleaved = True
        cfg.trainer.interleaved_training_flex_attention = True
        cfg.data.require_sample_ids = True
        cfg.model.use_flex_attention = True
    return cfg


def build_product(golden, device):
    cfg = product_config(golden.case)
    diff = Diffusion(cfg, None, device)
    assert diff.vocab_size == golden.case["vocab_size"] and diff.mask_index == golden.case["text_vocab_size"] - 1
    missing, unexpected = diff.backbone.load_state_dict(golden.params(), strict=True)
    diff.backbone.to(device)
    diff.backbone.train()
    return diff
