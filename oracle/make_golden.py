"""Generate golden vectors by running the IMPORTED reference (build container only).

    python -m oracle.make_golden            # writes tests/golden/*.npz

TEST INFRASTRUCTURE.  The reference (``/root/reference``, Python) cannot travel to the
GPU box, so its behaviour on the hot path is frozen here as small fixtures: inputs,
RNG-drawn quantities, logits / log-probs / per-token NLL / loss, and gradients, for a
fp32 run ("truth") and a CPU-bf16-autocast run ("ref_bf16": the reference's own bf16
noise floor).  The committed fixtures are data only; this script is the
recipe that made them.

Reference entry points exercised (file:line in /root/reference):
  model.py:157 update_batch, :589 _sample_t, :424 q_xt, :674 forward,
  :621 _subs_parameterization, :797 compute_loss; models/dit.py:1095 DIT (+blocks);
  models/noise_schedule.py:128 LogLinearNoise.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

from oracle import ref_shim  # noqa: E402
from oracle.cases import CASES, lumina_rope_2d  # noqa: E402


def _ref_cfg(case):
    """Map a flat hot-path case dict onto the reference's Hydra-style config tree."""
    C = ref_shim.Cfg
    m = case
    model = C(
        hidden_size=m["hidden_size"], n_heads=m["n_heads"], cond_dim=m["cond_dim"], n_blocks=m["n_blocks"],
        dropout=0.0, length=m["txt_length"] + m["img_length"], txt_length=m["txt_length"], img_length=m["img_length"],
        attn_type="flash", force_varlen_attn=False, norm_type=m["norm_type"], qk_norm=m["qk_norm"],
        sandwich_normalization=m["sandwich_normalization"], full_attention=True, modality_embed=m["modality_embed"],
        rope_2d=m["rope_2d"], linear_factor=m.get("linear_factor", 1.0), zero_linear_init=False, scale_by_sigma=False,
        use_spda_attn=True, force_optimized_native_attn=False, use_attention_mask=False,
        force_argmax_valid_indices=m["force_argmax_valid_indices"], flex_attention_img_masking_prob=None,
        flex_attention_txt_masking_prob=None, image_model=m["img_length"] > 0, unified_model=m["img_length"] > 0,
    )
    trainer = C(
        image_mode="discrete", multimodal_batches=m["multimodal_batches"], compile=False, compile_flag_pos_emb=True,
        interleaved=False, joint_ar_nar_timestep_warmup_steps=None, joint_ar_nar_prob=None, add_label=False,
        first_token_dropout=None, disable_forward_autocast_during_eval=False, force_bf16_eval=False, ar_shift=False,
        low_precision_loss=False, ar_llm_loss=False, allow_null_sigma=True, log_seperate_modal_losses=m["img_length"] > 0,
        text_loss_weight=m.get("text_loss_weight"), img_loss_weight=m.get("img_loss_weight"), ar_inpainting=False,
        ignore_text_in_unified=False,
    )
    for k in ("mask_entire_modality", "softmin_snr", "force_full_attention_mask_loss_only", "force_full_attention_mask",
              "set_max_txt_loss_ratio"):
        if m.get(k) is not None:
            setattr(trainer, k, m[k])
    # data.txt_only=True would trip the reference's own assert at model.py:311 (modality.max()==1) on text-only batches
    data = C(require_sample_ids=False, txt_only=False)
    return C(model=model, trainer=trainer, data=data, eval=C(), time_conditioning=m["time_conditioning"],
             parameterization="subs", backbone="dit", mode="train", T=0)


def build_reference(case, dtype):
    """Recipe B of SURVEY.md §8c: Diffusion(disable_init=True) wired by hand around DIT."""
    ref_shim.install()
    ref_shim.install_lumina_rope(lumina_rope_2d)
    import model as refmodel
    import models.dit as refdit
    import models.noise_schedule as refns

    cfg = _ref_cfg(case)
    V, Vt = case["vocab_size"], case["text_vocab_size"]
    static_txt = slice(None, case["txt_length"])
    static_img = slice(-case["img_length"], None) if case["img_length"] > 0 else slice(0, 0)
    torch.manual_seed(case["param_seed"])
    kw = dict(dtype=torch.float32) if dtype == torch.float32 else dict(autocast_dtype=dtype)  # F5 quirk
    backbone = refdit.DIT(cfg, vocab_size=V, text_vocab_size=Vt, mask_index=Vt - 1, device=torch.device("cpu"),
                          static_img_sl=static_img, static_txt_sl=static_txt, **kw)
    # Randomise every parameter (zero-init adaLN/head would hide those paths), deterministically by name.
    g = torch.Generator().manual_seed(case["param_seed"])
    with torch.no_grad():
        for name, p in sorted(backbone.named_parameters()):
            if name.endswith("norm1.weight") or name.endswith("norm2.weight") or "norm.weight" in name or name.endswith("norm_final.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "adaLN_modulation" in name:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
            elif "embed" in name:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p.shape[-1]))
    backbone.train()
    if case["qk_norm"]:  # F7: make eager CPU backward legal without changing numerics
        for blk in backbone.blocks:
            blk.attention.q_norm.register_forward_pre_hook(lambda mod, a: (a[0].clone(),))
            blk.attention.k_norm.register_forward_pre_hook(lambda mod, a: (a[0].clone(),))

    d = refmodel.Diffusion(cfg, None, torch.device("cpu"), disable_init=True)
    d.config, d.device, d.dtype = cfg, torch.device("cpu"), dtype
    d.parameterization, d.mask_index, d.vocab_size, d.text_vocab_size = "subs", Vt - 1, V, Vt
    d.neg_infinity, d.T, d.sampling_eps = -1000000.0, 0, 1e-3
    d.antithetic_sampling, d.importance_sampling, d.change_of_variables = True, False, False
    d.time_conditioning = case["time_conditioning"]
    d.global_step, d.current_run_fwd_bwd_pass = 0, 1
    d.image_model, d.unified_model = case["img_length"] > 0, case["img_length"] > 0
    d.noise = refns.LogLinearNoise()
    d.backbone = backbone
    d.visualize_samples = lambda *a, **k: None
    d.tokenizer = None
    return d


def make_batch(case):
    g = torch.Generator().manual_seed(case["data_seed"])
    B, Lt, Li, Vt, V = case["batch_size"], case["txt_length"], case["img_length"], case["text_vocab_size"], case["vocab_size"]
    batch = {}
    if Li > 0:
        batch["txt_input_ids"] = torch.randint(0, Vt - 1, (B, Lt), generator=g, dtype=torch.int32)
        batch["img_input_ids"] = torch.randint(0, V - Vt, (B, Li), generator=g, dtype=torch.int32).to(torch.int16)
        am = torch.ones(B, Lt, dtype=torch.bool)
        if case.get("ragged_text"):
            for b in range(B):
                am[b, Lt - b:] = False  # ragged padding at the tail of the text span
        batch["txt_attention_mask"] = am
    else:
        batch["input_ids"] = torch.randint(0, Vt - 1, (B, Lt), generator=g, dtype=torch.int64)
        am = torch.ones(B, Lt, dtype=torch.bool)
        if case.get("ragged_text"):
            for b in range(B):
                am[b, Lt - 2 * b:] = False
        batch["attention_mask"] = am
    return batch


def run_reference(case, dtype):
    d = build_reference(case, dtype)
    rec = {}
    orig_sample_t, orig_qxt, orig_subs = d._sample_t, d.q_xt, d._subs_parameterization

    def sample_t(n, device):
        t = orig_sample_t(n, device)
        rec["t"] = t.detach().clone()
        return t

    def q_xt(x, move_chance, **kw):
        out = orig_qxt(x, move_chance, **kw)
        xt, ign, _, smt, smi, move = out
        rec.update(x0=x.clone(), move_chance=move_chance.clone(), xt=xt.clone(), move_indices=move.clone())
        if ign is not None:
            rec.update(ignore_batch_mask=ign.clone(), should_mask_txt=smt.clone(), should_mask_img=smi.clone())
        return out

    def subs(logits, xt, **kw):
        rec["logits"] = logits.detach().float().clone()
        rec["logits_dtype"] = str(logits.dtype)
        out = orig_subs(logits, xt, **kw)
        rec["log_probs"] = out.detach().float().clone()
        return out

    d._sample_t, d.q_xt, d._subs_parameterization = sample_t, q_xt, subs
    batch = make_batch(case)
    torch.manual_seed(case["step_seed"])
    upd = d.update_batch({k: v.clone() for k, v in batch.items()})
    out = d.compute_loss(upd, "train", 1)
    out.loss.backward()
    rec.update(
        input_ids=upd["input_ids"], attention_mask=upd["attention_mask"],
        loss=out.loss.detach(), nlls=out.nlls.detach(), token_mask=out.token_mask,
    )
    if "modality" in upd:
        rec["modality"] = upd["modality"]
    for k in ("txt_loss", "img_loss", "txt_nlls", "img_nlls"):
        v = getattr(out, k)
        if torch.is_tensor(v):
            rec[k] = v.detach()
    for k, v in (out.extra_losses or {}).items():
        rec["extra/" + k] = torch.as_tensor(v).detach().float()
    grads = {n: p.grad.detach().clone() for n, p in d.backbone.named_parameters() if p.grad is not None}
    params = {n: p.detach().clone() for n, p in d.backbone.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in d.backbone.named_buffers() if "rotary" in n and torch.is_tensor(b)}
    return batch, rec, params, grads, bufs


def _np(v):
    v = v.detach() if torch.is_tensor(v) else torch.as_tensor(v)
    if v.dtype == torch.bfloat16:
        v = v.float()
    return v.cpu().numpy()


def main(names=None):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, case in CASES.items():
        if names and name not in names:
            continue
        out = {}
        batch, rec32, params, grads32, bufs = run_reference(case, torch.float32)
        _, rec16, params16, grads16, _ = run_reference(case, torch.bfloat16)
        for n in params:
            assert torch.equal(params[n], params16[n]), n
        for k, v in batch.items():
            out["batch/" + k] = _np(v)
        for k, v in params.items():
            out["param/" + k] = _np(v)
        for k, v in bufs.items():
            out["buffer/" + k] = _np(v)
        for tag, rec, grads in (("fp32", rec32, grads32), ("bf16", rec16, grads16)):
            for k, v in rec.items():
                if isinstance(v, str):
                    out[f"{tag}/{k}"] = np.array(v)
                else:
                    out[f"{tag}/{k}"] = _np(v)
            for k, v in grads.items():
                out[f"{tag}/grad/{k}"] = _np(v)
        # integer/boolean quantities must agree between the two reference runs (same RNG stream)
        for k in ("xt", "move_indices", "x0"):
            assert np.array_equal(out[f"fp32/{k}"], out[f"bf16/{k}"]), k
        path = os.path.join(GOLDEN_DIR, f"{name}.npz")
        np.savez_compressed(path, **out)
        l32, l16 = float(out["fp32/loss"]), float(out["bf16/loss"])
        print(f"{name}: loss fp32={l32:.6f} bf16={l16:.6f} rel={abs(l16 - l32) / abs(l32):.2e} "
              f"masked={int(out['fp32/move_indices'].sum())}/{out['fp32/move_indices'].size} -> {path} "
              f"({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
