"""Golden vectors for the `maskgit` predictor, made by running the IMPORTED reference (build container only).

    python -m oracle.make_golden_maskgit        # writes tests/golden/maskgit_<case>.npz

TEST INFRASTRUCTURE.  Reference entry points exercised (file:line in /root/reference): model_eval.py:2964 adap_sche ('arccos'),
:3046 _maskgit_update (through :1761 _ddpm_forward, no CFG), and the loop shape of :2274-2370 / :2425-2447 (schedule from the initial x,
timesteps = linspace(1, eps, steps + 1), noise-removal arg-max, x0 / x0_unmask conditioning) restated around the reference's own update.
Recorded per step: x before, the token `torch.multinomial` drew per position (captured by wrapping it), the Gumbel noise `np.random.gumbel`
produced (captured likewise), the backbone's logits, x after; plus the schedule and the final tokens.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from oracle import ref_shim
from oracle.cases import CASES
from oracle.make_golden import GOLDEN_DIR, build_reference, make_batch, _np

MASKGIT_CASES = {"c_large": dict(steps=6, eps=1e-5, seed=2024, conditional=True, r_temp=10.0),
                 "b_small": dict(steps=5, eps=1e-5, seed=99, conditional=False, r_temp=4.5)}
# `first_hitting` predictor (model_eval.py:3005-3043, linear schedule): recorded draws = the two torch.rand_like calls of a step (token race
# uniforms [B, L, V] inside _sample_categorical, position lottery [B, L])
FIRST_HITTING_CASES = {"c_large": dict(steps=6, eps=1e-5, seed=555, conditional=True), "b_small": dict(steps=4, eps=1e-5, seed=556, conditional=False)}


def run_first_hitting(name, spec):
    case = CASES[name]
    d = build_reference(case, torch.float32)
    d.backbone.eval()
    C = ref_shim.Cfg
    d.config.noise = C(type="loglinear")
    d.config.eval = C(cfg=None, attention_caching=False)
    d.config.trainer.force_null_sigma = False
    d.config.trainer.interleaved_training_flex_attention = False
    d.config.sampling = C(predictor="first_hitting", steps=spec["steps"], noise_removal=True)
    d.sampler = "first_hitting"
    import model_eval as ref_eval
    import model_utils as ref_utils

    batch = d.update_batch({k: v.clone() for k, v in make_batch(case).items()})
    x0_data, modality = batch["input_ids"], batch.get("modality")
    B, L = x0_data.shape
    steps, eps = spec["steps"], spec["eps"]
    x0 = x0_unmask = None
    if spec["conditional"]:
        x0 = x0_data.clone()
        x0_unmask = torch.zeros(B, L, dtype=torch.bool)
        x0_unmask[:, : case["txt_length"]] = True
    x = d._sample_prior(B, L)
    if x0 is not None:
        x = torch.where(x0_unmask, x0, x)
    schedule = ref_eval.adap_sche(x=x, step=steps, mask_index=d.mask_index, mode="linear")
    timesteps = torch.linspace(1, eps, steps + 1)
    dt = (1 - eps) / steps
    rec = {"x_init": x.clone(), "timesteps": timesteps.clone(), "dt": torch.tensor(dt), "schedule": schedule.clone()}
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

    nfe = 0
    torch.manual_seed(spec["seed"])
    with torch.no_grad():
        for i in range(steps):
            t = timesteps[i] * torch.ones(B, 1)
            rec[f"step{i}/x"] = x.clone()
            rec[f"step{i}/logits"] = d.forward(x=x, sigma=d.noise(t)[0], return_logits=True, **kwargs).float().clone()
            torch.rand_like = rand_like
            ref_utils.torch.rand_like = rand_like
            try:
                x, n = d._first_hitting_update(x, t, dt, x0=x0, x0_unmask=x0_unmask, schedule=schedule, step=i, **kwargs)
            finally:
                torch.rand_like = orig_rand_like
            nfe += n
            rec[f"step{i}/u"] = drawn[0]
            if len(drawn) > 1:
                rec[f"step{i}/pos_u"] = drawn[1]
            drawn.clear()
            rec[f"step{i}/x_next"] = x.clone()
        t = timesteps[-1] * torch.ones(B, 1)
        x_final = d.forward(x=x, sigma=d.noise(t)[0], **kwargs).argmax(dim=-1)
        if x0 is not None:
            x_final = torch.where(x0_unmask, x0, x_final)
    rec["x_before_noise_removal"] = x.clone()
    rec["x_final"] = x_final.clone()
    rec["nfe"] = torch.tensor(nfe)
    return rec


# `maskgit_nucleus` predictor (model_eval.py:3118-3167, nucleus_sampling_batch :2642-2685).  The reference's early exit `if num_unmask <= 0`
# is a tensor truth test, so the update only runs at batch size 1: the fixture takes the first row of the case's batch.
NUCLEUS_CASES = {"c_large": dict(steps=6, eps=1e-5, seed=4321, conditional=True, r_temp=10.0, top_p=0.9, temperature=0.8)}


def run_nucleus(name, spec):
    case = CASES[name]
    d = build_reference(case, torch.float32)
    d.backbone.eval()
    C = ref_shim.Cfg
    d.config.noise = C(type="loglinear")
    d.config.eval = C(cfg=None, attention_caching=False, maskgit_r_temp=spec["r_temp"], top_p=spec["top_p"], temperature=spec["temperature"])
    d.config.trainer.force_null_sigma = False
    d.config.trainer.interleaved_training_flex_attention = False
    d.config.sampling = C(predictor="maskgit_nucleus", steps=spec["steps"], noise_removal=True)
    d.sampler = "maskgit_nucleus"
    import model_eval as ref_eval

    batch = d.update_batch({k: v[:1].clone() for k, v in make_batch(case).items()})
    x0_data, modality = batch["input_ids"], batch.get("modality")
    B, L = x0_data.shape
    steps, eps = spec["steps"], spec["eps"]
    x0 = x0_unmask = None
    if spec["conditional"]:
        x0 = x0_data.clone()
        x0_unmask = torch.zeros(B, L, dtype=torch.bool)
        x0_unmask[:, : case["txt_length"]] = True
    x = d._sample_prior(B, L)
    if x0 is not None:
        x = torch.where(x0_unmask, x0, x)
    schedule = ref_eval.adap_sche(x=x, step=steps, mask_index=d.mask_index, mode="arccos")
    timesteps = torch.linspace(1, eps, steps + 1)
    dt = (1 - eps) / steps
    rec = {"x_init": x.clone(), "timesteps": timesteps.clone(), "dt": torch.tensor(dt), "schedule": schedule.clone(), "r_temp": torch.tensor(spec["r_temp"]),
           "top_p": torch.tensor(spec["top_p"]), "temperature": torch.tensor(spec["temperature"])}
    if modality is not None:
        rec["modality"] = modality.clone()
    if x0 is not None:
        rec.update(x0=x0.clone(), x0_unmask=x0_unmask.clone())
    kwargs = dict(modality=modality) if modality is not None else {}
    drawn, gum = [], []
    orig_nucleus, orig_gumbel = ref_eval.nucleus_sampling_batch, np.random.gumbel

    def nucleus(p_x0, *a, **k):
        out = orig_nucleus(p_x0, *a, **k)
        drawn.append((out.detach().clone(), p_x0.detach().clone()))
        return out

    def gumbel(*a, **k):
        out = orig_gumbel(*a, **k)
        gum.append(np.array(out))
        return out

    nfe = 0
    torch.manual_seed(spec["seed"])
    np.random.seed(spec["seed"])
    with torch.no_grad():
        for i in range(steps):
            t = timesteps[i] * torch.ones(B, 1)
            rec[f"step{i}/x"] = x.clone()
            rec[f"step{i}/logits"] = d.forward(x=x, sigma=d.noise(t)[0], return_logits=True, **kwargs).float().clone()
            ref_eval.nucleus_sampling_batch, np.random.gumbel = nucleus, gumbel
            try:
                x, n = d._maskgit_nucleus_update(x, t, dt, x0=x0, x0_unmask=x0_unmask, schedule=schedule, step=i, **kwargs)
            finally:
                ref_eval.nucleus_sampling_batch, np.random.gumbel = orig_nucleus, orig_gumbel
            nfe += n
            if n:
                pred, p_x0 = drawn.pop()
                rec[f"step{i}/pred"] = pred.reshape(B, L)
                rec[f"step{i}/p_x0"] = p_x0.reshape(B, L, -1)
                rec[f"step{i}/gumbel"] = torch.from_numpy(gum.pop()).reshape(B, L)
            assert not drawn and not gum
            if x0 is not None:
                x = torch.where(x0_unmask, x0, x)
            rec[f"step{i}/x_next"] = x.clone()
        t = timesteps[-1] * torch.ones(B, 1)
        x_final = d.forward(x=x, sigma=d.noise(t)[0], **kwargs).argmax(dim=-1)
        if x0 is not None:
            x_final = torch.where(x0_unmask, x0, x_final)
    rec["x_before_noise_removal"] = x.clone()
    rec["x_final"] = x_final.clone()
    rec["nfe"] = torch.tensor(nfe)
    return rec


def run(name, spec):
    case = CASES[name]
    d = build_reference(case, torch.float32)
    d.backbone.eval()
    C = ref_shim.Cfg
    d.config.noise = C(type="loglinear")
    d.config.eval = C(cfg=None, attention_caching=False, maskgit_r_temp=spec["r_temp"])
    d.config.trainer.force_null_sigma = False
    d.config.trainer.interleaved_training_flex_attention = False
    d.config.sampling = C(predictor="maskgit", steps=spec["steps"], noise_removal=True)
    d.sampler = "maskgit"
    import model_eval as ref_eval

    batch = d.update_batch({k: v.clone() for k, v in make_batch(case).items()})
    x0_data = batch["input_ids"]
    modality = batch.get("modality")
    B, L = x0_data.shape
    steps, eps = spec["steps"], spec["eps"]
    x0 = x0_unmask = None
    if spec["conditional"]:
        x0 = x0_data.clone()
        x0_unmask = torch.zeros(B, L, dtype=torch.bool)
        x0_unmask[:, : case["txt_length"]] = True
    x = d._sample_prior(B, L)
    if x0 is not None:
        x = torch.where(x0_unmask, x0, x)
    schedule = ref_eval.adap_sche(x=x, step=steps, mask_index=d.mask_index, mode="arccos")
    timesteps = torch.linspace(1, eps, steps + 1)
    dt = (1 - eps) / steps
    rec = {"x_init": x.clone(), "timesteps": timesteps.clone(), "dt": torch.tensor(dt), "schedule": schedule.clone(), "r_temp": torch.tensor(spec["r_temp"])}
    if modality is not None:
        rec["modality"] = modality.clone()
    if x0 is not None:
        rec.update(x0=x0.clone(), x0_unmask=x0_unmask.clone())
    kwargs = dict(modality=modality) if modality is not None else {}
    drawn, gum = [], []
    orig_multinomial, orig_gumbel = torch.multinomial, np.random.gumbel

    def multinomial(*a, **k):
        out = orig_multinomial(*a, **k)
        drawn.append(out.detach().clone())
        return out

    def gumbel(*a, **k):
        out = orig_gumbel(*a, **k)
        gum.append(np.array(out))
        return out

    nfe = 0
    torch.manual_seed(spec["seed"])
    np.random.seed(spec["seed"])
    with torch.no_grad():
        for i in range(steps):
            t = timesteps[i] * torch.ones(B, 1)
            rec[f"step{i}/x"] = x.clone()
            rec[f"step{i}/logits"] = d.forward(x=x, sigma=d.noise(t)[0], return_logits=True, **kwargs).float().clone()
            torch.multinomial, np.random.gumbel = multinomial, gumbel
            try:
                x, n = d._maskgit_update(x, t, dt, x0=x0, x0_unmask=x0_unmask, schedule=schedule, step=i, **kwargs)
            finally:
                torch.multinomial, np.random.gumbel = orig_multinomial, orig_gumbel
            nfe += n
            if n:
                rec[f"step{i}/pred"] = drawn.pop().reshape(B, L)
                rec[f"step{i}/gumbel"] = torch.from_numpy(gum.pop()).reshape(B, L)
            assert not drawn and not gum
            rec[f"step{i}/x_next"] = x.clone()
        t = timesteps[-1] * torch.ones(B, 1)
        x_final = d.forward(x=x, sigma=d.noise(t)[0], **kwargs).argmax(dim=-1)
        if x0 is not None:
            x_final = torch.where(x0_unmask, x0, x_final)
    rec["x_before_noise_removal"] = x.clone()
    rec["x_final"] = x_final.clone()
    rec["nfe"] = torch.tensor(nfe)
    return rec


def main(names=None):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, spec in MASKGIT_CASES.items():
        if names and name not in names:
            continue
        rec = run(name, spec)
        out = {k: _np(v) for k, v in rec.items()}
        out["steps"], out["eps"], out["seed"] = np.array(spec["steps"]), np.array(spec["eps"]), np.array(spec["seed"])
        path = os.path.join(GOLDEN_DIR, f"maskgit_{name}.npz")
        np.savez_compressed(path, **out)
        left = int((rec["x_before_noise_removal"] == CASES[name]["text_vocab_size"] - 1).sum())
        print(f"maskgit_{name}: steps={spec['steps']} nfe={int(rec['nfe'])} schedule={rec['schedule'].tolist()} masks left={left} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    for name, spec in NUCLEUS_CASES.items():
        if names and ("nucleus_" + name) not in names and name not in names:
            continue
        rec = run_nucleus(name, spec)
        out = {k: _np(v) for k, v in rec.items()}
        out["steps"], out["eps"], out["seed"] = np.array(spec["steps"]), np.array(spec["eps"]), np.array(spec["seed"])
        path = os.path.join(GOLDEN_DIR, f"maskgit_nucleus_{name}.npz")
        np.savez_compressed(path, **out)
        left = int((rec["x_before_noise_removal"] == CASES[name]["text_vocab_size"] - 1).sum())
        print(f"maskgit_nucleus_{name}: steps={spec['steps']} nfe={int(rec['nfe'])} schedule={rec['schedule'].tolist()} masks left={left} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    for name, spec in FIRST_HITTING_CASES.items():
        if names and ("fh_" + name) not in names and name not in names:
            continue
        rec = run_first_hitting(name, spec)
        out = {k: _np(v) for k, v in rec.items()}
        out["steps"], out["eps"], out["seed"] = np.array(spec["steps"]), np.array(spec["eps"]), np.array(spec["seed"])
        path = os.path.join(GOLDEN_DIR, f"first_hitting_{name}.npz")
        np.savez_compressed(path, **out)
        left = int((rec["x_before_noise_removal"] == CASES[name]["text_vocab_size"] - 1).sum())
        print(f"first_hitting_{name}: steps={spec['steps']} nfe={int(rec['nfe'])} schedule={rec['schedule'].tolist()} masks left={left} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main(sys.argv[1:] or None)
