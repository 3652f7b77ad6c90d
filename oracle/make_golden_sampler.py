"""Golden vectors for the sampler inner loop, made by running the IMPORTED reference (build container only).

    python -m oracle.make_golden_sampler        # writes tests/golden/sampler_<case>.npz

TEST INFRASTRUCTURE (same status as make_golden.py).  Reference entry points exercised (file:line in /root/reference):
  model_eval.py:2073 _ddpm_caching_update, :1761 _ddpm_forward (no-CFG branch), model_utils.py:95 _sample_categorical,
  model.py:674 forward (log-probs and return_logits=True), and the `ddpm_cache` branch of the loop in model_eval.py:2307-2444
  (timesteps = linspace(1, eps, steps+1), dt = (1-eps)/steps, cache reuse when nothing changed, x0/x0_unmask conditioning,
  noise_removal arg-max).  The loop around the reference's own update function is restated here because `_sample` itself needs a
  tokenizer, decode and logging stack; every model/sampling call inside it is the reference's.

Recorded per step: x before, t, the uniforms `torch.rand_like` drew (captured by wrapping torch.rand_like), the logits the backbone
produced, x after.  Plus the final (noise-removed) tokens.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from oracle import ref_shim
from oracle.cases import CASES
from oracle.make_golden import GOLDEN_DIR, build_reference, make_batch, _np

SAMPLER_CASES = {"c_large": dict(steps=6, eps=1e-5, seed=1234, conditional=True), "b_small": dict(steps=5, eps=1e-5, seed=77, conditional=False),
                 # classifier-free guidance (`_ddpm_forward` CFG branch :1763-1817, `get_cfg_weight` :1737-1758): text kept as conditioning, w = cfg (1 - t)
                 "c_large_cfg": dict(case="c_large", steps=6, eps=1e-5, seed=4321, conditional=True, cfg=2.0)}


def run(name, spec):
    case = CASES[spec.get("case", name)]
    d = build_reference(case, torch.float32)
    d.backbone.eval()
    C = ref_shim.Cfg
    d.config.noise = C(type="loglinear")
    d.config.eval = C(cfg=spec.get("cfg"), attention_caching=False)
    d.config.trainer.interleaved_training_flex_attention = False
    d.config.trainer.force_null_sigma = False
    d.config.sampling = C(predictor="ddpm_cache", steps=spec["steps"], noise_removal=True)
    d.sampler = "ddpm_cache"
    import model_utils as ref_utils
    import model_eval as ref_eval

    batch = d.update_batch({k: v.clone() for k, v in make_batch(case).items()})
    x0_data = batch["input_ids"]
    modality = batch.get("modality")
    B, L = x0_data.shape
    steps, eps = spec["steps"], spec["eps"]
    x0 = x0_unmask = None
    if spec["conditional"]:  # keep the text half as conditioning (x0 / x0_unmask branch of _sample)
        x0 = x0_data.clone()
        x0_unmask = torch.zeros(B, L, dtype=torch.bool)
        x0_unmask[:, : case["txt_length"]] = True
    x = d._sample_prior(B, L)
    if x0 is not None:
        x = torch.where(x0_unmask, x0, x)
    timesteps = torch.linspace(1, eps, steps + 1)
    dt = (1 - eps) / steps
    rec = {"x_init": x.clone(), "timesteps": timesteps.clone(), "dt": torch.tensor(dt)}
    if modality is not None:
        rec["modality"] = modality.clone()
    if x0 is not None:
        rec.update(x0=x0.clone(), x0_unmask=x0_unmask.clone())
    kwargs = dict(modality=modality) if modality is not None else {}
    drawn = []
    orig_rand_like = torch.rand_like

    def rand_like(t, *a, **k):
        u = orig_rand_like(t, *a, **k)
        drawn.append(u.detach().clone())
        return u

    p_x0_cache = None
    nfe = 0
    torch.manual_seed(spec["seed"])
    with torch.no_grad():
        for i in range(steps):
            t = timesteps[i] * torch.ones(B, 1)
            rec[f"step{i}/x"] = x.clone()
            rec[f"step{i}/reused_cache"] = torch.tensor(p_x0_cache is not None)
            if p_x0_cache is None:  # the logits this step's forward sees (side call: deterministic, consumes no RNG)
                sigma_t, _ = d.noise(t)
                rec[f"step{i}/logits"] = d.forward(x=x, sigma=sigma_t, return_logits=True, **kwargs).float().clone()
                if spec.get("cfg") is not None:  # the unconditional half (conditioning positions masked) and the guidance weight of this step
                    x_uncond = x.clone()
                    x_uncond[x0_unmask] = d.mask_index
                    rec[f"step{i}/logits_uncond"] = d.forward(x=x_uncond, sigma=sigma_t, return_logits=True, **kwargs).float().clone()
                    rec[f"step{i}/cfg_w"] = d.get_cfg_weight(t.squeeze(-1)).float().clone()
            torch.rand_like = rand_like
            ref_utils.torch.rand_like = rand_like
            try:
                p_x0_cache, x_next, n = d._ddpm_caching_update(x, t, dt, p_x0=p_x0_cache, x0=x0, x0_unmask=x0_unmask, **kwargs)
            finally:
                torch.rand_like = orig_rand_like
            nfe += n
            rec[f"step{i}/u"] = drawn.pop()
            assert not drawn
            rec[f"step{i}/p_x0"] = p_x0_cache.float().clone()
            if not torch.allclose(x_next, x) or d.time_conditioning:
                p_x0_cache = None
            x = x_next
            if x0 is not None:
                x = torch.where(x0_unmask, x0, x)
            rec[f"step{i}/x_next"] = x.clone()
        t = timesteps[-1] * torch.ones(B, 1)
        x_final = d.forward(x=x, sigma=d.noise(t)[0], **kwargs).argmax(dim=-1)   # (the reference's noise removal is unguided)
        if x0 is not None:
            x_final = torch.where(x0_unmask, x0, x_final)
    rec["x_before_noise_removal"] = x.clone()
    rec["x_final"] = x_final.clone()
    rec["nfe"] = torch.tensor(nfe)
    return rec


# `eval.attention_caching` (model_eval.py:2296-2366, :2425-2440; models/dit.py:784-812): every `ratio` steps a full joint update, the step after it a full
# update in which image queries see image keys only (the "cache-building" step), all other steps on the TEXT slice alone.  (The reference writes its
# per-layer flex-attention cache in those steps but never reads it back: dit.py:797-803 vs :812 - text-only steps attend to text keys only.)
CACHING_CASES = {"c_large_attn_caching": dict(case="c_large", steps=8, eps=1e-5, seed=777, ratio=3)}


def run_caching(name, spec):
    import torch.nn.functional as F
    case = CASES[spec["case"]]
    ref_shim.install()
    import models.dit as refdit
    import model_utils as ref_utils
    import model_eval as ref_eval

    class DenseMask:
        def __init__(self, m):
            self.m = m

    def create_block_mask(mask_mod, B, H, Q_LEN, KV_LEN, device=None, **kw):
        b = torch.arange(B)[:, None, None]
        q = torch.arange(Q_LEN)[None, :, None]
        kv = torch.arange(KV_LEN)[None, None, :]
        return DenseMask(mask_mod(b, None, q, kv).expand(B, Q_LEN, KV_LEN)[:, None])

    def flex_attention(q, k, v, block_mask=None, **kw):   # FlexAttention -> its definition (dense-mask SDPA), as in make_golden_interleaved.py
        return F.scaled_dot_product_attention(q, k, v, attn_mask=None if block_mask is None else block_mask.m)

    refdit.flex_attention = refdit.compiled_flex_attention = flex_attention
    refdit.create_block_mask = create_block_mask
    import torch.nn.attention.flex_attention as fa
    fa.create_block_mask = create_block_mask
    ref_utils.create_block_mask = create_block_mask

    import oracle.make_golden as MG
    orig_cfg = MG._ref_cfg

    def cfg_flex(c):
        cfg = orig_cfg(c)
        cfg.model.use_flex_attention = True
        return cfg

    MG._ref_cfg = cfg_flex
    try:
        d = build_reference(case, torch.float32)
    finally:
        MG._ref_cfg = orig_cfg
    d.backbone.eval()
    C = ref_shim.Cfg
    d.config.noise = C(type="loglinear")
    d.config.eval = C(cfg=None, attention_caching=True, attention_caching_txt_to_img_ratio=spec["ratio"])
    d.config.trainer.interleaved_training_flex_attention = False
    d.config.trainer.force_null_sigma = False
    d.config.sampling = C(predictor="ddpm_cache", steps=spec["steps"], noise_removal=True)
    d.sampler = "ddpm_cache"

    batch = d.update_batch({k: v.clone() for k, v in make_batch(case).items()})
    modality = batch.get("modality")
    B, L = batch["input_ids"].shape
    steps, eps, ratio = spec["steps"], spec["eps"], spec["ratio"]
    x = d._sample_prior(B, L)
    x0 = x0_unmask = None
    timesteps = torch.linspace(1, eps, steps + 1)
    dt = (1 - eps) / steps
    rec = {"x_init": x.clone(), "timesteps": timesteps.clone(), "dt": torch.tensor(dt), "ratio": torch.tensor(ratio), "modality": modality.clone()}
    kwargs = dict(modality=modality)
    drawn = []
    orig_rand_like = torch.rand_like

    def rand_like(t, *a, **k):
        u = orig_rand_like(t, *a, **k)
        drawn.append(u.detach().clone())
        return u

    d.backbone.set_flex_attention_cache(B, L, x.device, torch.float32)
    txt_sl = d.static_txt_sl
    p_x0_cache, x_next = None, None
    is_x_sliced, full_data = False, dict()
    nfe = 0
    torch.manual_seed(spec["seed"])
    with torch.no_grad():
        for i in range(steps):
            t = timesteps[i] * torch.ones(x.shape[0], 1)
            # ---- model_eval.py:2311-2366, restated
            if i % ratio == 0:
                if is_x_sliced:
                    def replace_new_data(_key, _new):
                        if full_data[_key] is not None:
                            full_data[_key][:, txt_sl] = _new
                        return full_data[_key]
                    x = replace_new_data("x", x)
                    p_x0_cache = replace_new_data("p_x0_cache", p_x0_cache)
                    kwargs["modality"] = replace_new_data("modality", kwargs.get("modality"))
                    full_data = dict()
                    is_x_sliced = False
                update_cache_slice, block_mask, mode = None, True, "full"
            elif (i - 1) % ratio == 0:
                update_cache_slice = slice(0, x.shape[1])
                block_mask = ref_utils.get_block_mask(txt_batch_attn_dropout=torch.zeros(x.shape[0], dtype=torch.bool),
                                                      img_batch_attn_dropout=torch.ones(x.shape[0], dtype=torch.bool), txt_length=d.config.model.txt_length,
                                                      batch_size=x.shape[0], seq_len=x.shape[1], device=x.device)
                mode = "build"
            else:
                update_cache_slice, block_mask, mode = txt_sl, True, "text"
                if not is_x_sliced:
                    is_x_sliced = True
                    cl = lambda v: None if v is None else v.clone()
                    sl = lambda v: None if v is None else v[:, txt_sl]
                    full_data.update(x=cl(x), modality=cl(kwargs.get("modality")), p_x0_cache=cl(p_x0_cache))
                    x, x_next, p_x0_cache = sl(x), sl(x_next), sl(p_x0_cache)
                    kwargs["modality"] = sl(kwargs.get("modality"))
            kwargs["update_cache_slice"], kwargs["block_mask"] = update_cache_slice, block_mask
            rec[f"step{i}/mode"] = np.array(mode)
            rec[f"step{i}/x"] = x.clone()
            rec[f"step{i}/reused_cache"] = torch.tensor(p_x0_cache is not None)
            torch.rand_like = rand_like
            ref_utils.torch.rand_like = rand_like
            try:
                p_x0_cache, x_next, n = d._ddpm_caching_update(x, t, dt, p_x0=p_x0_cache, x0=x0, x0_unmask=x0_unmask, **kwargs)
            finally:
                torch.rand_like = orig_rand_like
            nfe += n
            rec[f"step{i}/u"] = drawn.pop()
            assert not drawn
            rec[f"step{i}/p_x0"] = p_x0_cache.float().clone()
            if not torch.allclose(x_next, x) or d.time_conditioning:
                p_x0_cache = None
            x = x_next
            rec[f"step{i}/x_next"] = x.clone()
        if is_x_sliced:   # model_eval.py:2425-2440
            full_data["x"][:, txt_sl] = x
            x = full_data["x"]
            if full_data["modality"] is not None:
                full_data["modality"][:, txt_sl] = kwargs["modality"]
                kwargs["modality"] = full_data["modality"]
        kwargs.pop("update_cache_slice"), kwargs.pop("block_mask")
        d.backbone.use_flex_attention_cache = False
        for blk in d.backbone.blocks:
            blk.attention.use_flex_attention_cache = False
        t = timesteps[-1] * torch.ones(B, 1)
        x_final = d.forward(x=x, sigma=d.noise(t)[0], block_mask=True, **kwargs).argmax(dim=-1)
    rec["x_before_noise_removal"] = x.clone()
    rec["x_final"] = x_final.clone()
    rec["nfe"] = torch.tensor(nfe)
    return rec


def main(names=None):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, spec in SAMPLER_CASES.items():
        if names and name not in names:
            continue
        rec = run(name, spec)
        out = {k: _np(v) for k, v in rec.items()}
        out["steps"], out["eps"], out["seed"] = np.array(spec["steps"]), np.array(spec["eps"]), np.array(spec["seed"])
        path = os.path.join(GOLDEN_DIR, f"sampler_{name}.npz")
        np.savez_compressed(path, **out)
        left = int((rec["x_before_noise_removal"] == CASES[spec.get("case", name)]["text_vocab_size"] - 1).sum())
        print(f"sampler_{name}: steps={spec['steps']} nfe={int(rec['nfe'])} masks left before noise removal={left} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main_caching(names=None):
    for name, spec in CACHING_CASES.items():
        if names and name not in names:
            continue
        rec = run_caching(name, spec)
        out = {k: (v if isinstance(v, np.ndarray) else _np(v)) for k, v in rec.items()}
        out["steps"], out["eps"], out["seed"] = np.array(spec["steps"]), np.array(spec["eps"]), np.array(spec["seed"])
        path = os.path.join(GOLDEN_DIR, f"sampler_{name}.npz")
        np.savez_compressed(path, **out)
        left = int((rec["x_before_noise_removal"] == CASES[spec["case"]]["text_vocab_size"] - 1).sum())
        print(f"sampler_{name}: steps={spec['steps']} nfe={int(rec['nfe'])} modes={[str(rec[f'step{i}/mode']) for i in range(spec['steps'])]} masks left={left} -> {path} "
              f"({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or any(a in SAMPLER_CASES for a in args):
        main(args or None)
    if not args or any(a in CACHING_CASES for a in args):
        main_caching(args or None)
