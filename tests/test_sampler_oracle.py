"""the oracle's restatement of the `ddpm_cache` sampler loop against golden vectors recorded from the imported reference
(oracle/make_golden_sampler.py): identical uniforms in, identical tokens out at every step."""
import os

import numpy as np
import pytest
import torch

from golden_utils import GOLDEN_DIR, Golden
from oracle import unidisc_oracle as O


def oracle_setup(g):
    cfg = g.cfg
    return cfg, g.params(), g.buffers(), O.update_batch(cfg, g.batch())

SAMPLER_CASES = ["c_large", "b_small"]


def load_sampler(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"sampler_{name}.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if z[k].dtype.kind != "U"}   # (string entries: read them from the npz directly)


@pytest.mark.parametrize("name", SAMPLER_CASES)
def test_oracle_sampler_matches_reference_tokens(name):
    g = Golden(name)
    s = load_sampler(name)
    cfg, P, buffers, batch = oracle_setup(g)
    steps = int(s["steps"])
    us = [s[f"step{i}/u"] for i in range(steps)]
    modality = s["modality"] if "modality" in s else None
    x0, x0_unmask = (s["x0"], s["x0_unmask"].bool()) if "x0" in s else (None, None)
    x_final, xs, x_last, nfe = O.sample_ddpm_cache(cfg, P, buffers, s["x_init"], s["timesteps"], float(s["dt"]), us, x0=x0, x0_unmask=x0_unmask,
                                                   modality=modality, batch=batch)
    for i in range(steps):
        assert torch.equal(xs[i], s[f"step{i}/x_next"]), f"step {i}"
    assert torch.equal(x_last, s["x_before_noise_removal"]) and torch.equal(x_final, s["x_final"])
    assert nfe == int(s["nfe"])
    # per-step probabilities of the restated forward agree with the reference's to fp32 accuracy
    p0, _ = O.ddpm_forward(cfg, P, buffers, s["step0/x"], O.loglinear_noise(s["timesteps"][0] * torch.ones(s["x_init"].shape[0]))[0], modality, batch)
    assert torch.allclose(p0, s["step0/p_x0"], atol=2e-6, rtol=1e-4)


def test_oracle_sampler_with_guidance_matches_reference_tokens():
    """CFG branch of `_ddpm_forward` (model_eval.py:1763-1817) + `get_cfg_weight`: same uniforms in, same tokens out; the restated guidance
    weight and both logits halves agree with what the reference produced."""
    g = Golden("c_large")
    s = load_sampler("c_large_cfg")
    cfg, P, buffers, batch = oracle_setup(g)
    steps = int(s["steps"])
    us = [s[f"step{i}/u"] for i in range(steps)]
    x0, x0_unmask = s["x0"], s["x0_unmask"].bool()
    x_final, xs, x_last, nfe = O.sample_ddpm_cache(cfg, P, buffers, s["x_init"], s["timesteps"], float(s["dt"]), us, x0=x0, x0_unmask=x0_unmask,
                                                   modality=s["modality"], batch=batch, cfg_scale=2.0)
    for i in range(steps):
        assert torch.equal(xs[i], s[f"step{i}/x_next"]), f"step {i}"
    assert torch.equal(x_final, s["x_final"]) and nfe == int(s["nfe"])
    B = s["x_init"].shape[0]
    for i in range(steps):
        if f"step{i}/cfg_w" not in s:
            continue
        t = s["timesteps"][i] * torch.ones(B)
        w = O.cfg_weight(2.0, t)
        assert torch.allclose(w, s[f"step{i}/cfg_w"], atol=1e-7)
        p, lg = O.ddpm_forward(cfg, P, buffers, s[f"step{i}/x"], O.loglinear_noise(t)[0], s["modality"], batch, x0_unmask=x0_unmask, w=w)
        if (w > 0).any():   # (t = 1 gives w = 0: the reference takes the unguided branch there)
            lc, lu = lg
            assert torch.allclose(lc, s[f"step{i}/logits"], atol=2e-5, rtol=1e-4) and torch.allclose(lu, s[f"step{i}/logits_uncond"], atol=2e-5, rtol=1e-4)
        else:
            assert i == 0 and torch.allclose(lg, s[f"step{i}/logits"], atol=2e-5, rtol=1e-4)
        assert torch.allclose(p, s[f"step{i}/p_x0"], atol=2e-6, rtol=1e-4)
    # guidance changes the outcome: the unguided loop on the same uniforms ends elsewhere
    x_plain, *_ = O.sample_ddpm_cache(cfg, P, buffers, s["x_init"], s["timesteps"], float(s["dt"]), us, x0=x0, x0_unmask=x0_unmask,
                                      modality=s["modality"], batch=batch)
    assert not torch.equal(x_plain, x_final)


def test_cfg_weight_windows():
    t = torch.tensor([0.1, 0.5, 0.9])
    assert torch.allclose(O.cfg_weight(3.0, t), (3.0 * (1 - t))[:, None])
    w = O.cfg_weight(3.0, t, cfg_min_timestep=0.2, cfg_max_timestep=0.8)
    assert w.shape == (3, 3) or w.shape == (3, 1) or w.ndim >= 1   # (the reference broadcasts [B,1] against [B]: kept as is)
    assert float(O.cfg_weight(1.5, t, force_cfg_value=True)) == 1.5


@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_oracle_maskgit_matches_reference_tokens(name):
    """`maskgit` predictor (model_eval.py:3046-3114, schedule :2964-3001): with the reference's multinomial draws and Gumbel noise replayed,
    the restated update reveals exactly the same tokens at every step."""
    g = Golden(name)
    z = np.load(os.path.join(GOLDEN_DIR, f"maskgit_{name}.npz"))
    s = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    cfg, P, buffers, batch = oracle_setup(g)
    steps = int(s["steps"])
    preds = [s.get(f"step{i}/pred") for i in range(steps)]
    gums = [s[f"step{i}/gumbel"].float() if f"step{i}/gumbel" in s else None for i in range(steps)]
    modality = s["modality"] if "modality" in s else None
    x0, x0_unmask = (s["x0"], s["x0_unmask"].bool()) if "x0" in s else (None, None)
    x_final, xs, x_last, nfe, schedule = O.sample_maskgit(cfg, P, buffers, s["x_init"], s["timesteps"], float(s["dt"]), preds, gums, float(s["r_temp"]),
                                                          x0=x0, x0_unmask=x0_unmask, modality=modality, batch=batch)
    assert torch.equal(schedule, s["schedule"].to(schedule.dtype))
    for i in range(steps):
        assert torch.equal(xs[i], s[f"step{i}/x_next"]), f"step {i}"
    assert torch.equal(x_final, s["x_final"]) and nfe == int(s["nfe"])
    # the number of tokens revealed per step follows the schedule
    m = cfg.mask_index
    prev = s["x_init"]
    for i in range(steps):
        revealed = ((prev == m) & (xs[i] != m)).sum(-1)
        assert torch.equal(revealed, torch.minimum(schedule[:, i].long(), (prev == m).sum(-1)))
        prev = xs[i]


def test_adap_sche_modes_and_edges():
    x = torch.tensor([[9, 9, 9, 9, 9, 9, 9, 9, 9, 9], [1, 9, 2, 9, 3, 9, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7, 8, 1, 2]])
    for mode in ("arccos", "linear", "cosine", "root", "square"):
        s = O.adap_sche(x, 4, 9, mode)
        assert s.shape == (3, 4) and (s >= 0).all()
    s = O.adap_sche(x, 4, 9, "arccos")
    assert int(s[0].sum()) == 10 and int(s[2, -1]) == 0      # nothing masked: the lifted ones are paid back by the last step (clamped at 0)


@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_oracle_first_hitting_matches_reference_tokens(name):
    """`first_hitting` predictor (model_eval.py:3005-3043, linear schedule): both rand_like draws replayed, identical tokens at every step."""
    g = Golden(name)
    z = np.load(os.path.join(GOLDEN_DIR, f"first_hitting_{name}.npz"))
    s = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    cfg, P, buffers, batch = oracle_setup(g)
    steps = int(s["steps"])
    us = [s[f"step{i}/u"] for i in range(steps)]
    pos = [s.get(f"step{i}/pos_u") for i in range(steps)]
    modality = s["modality"] if "modality" in s else None
    x0, x0_unmask = (s["x0"], s["x0_unmask"].bool()) if "x0" in s else (None, None)
    x_final, xs, x_last, nfe, schedule = O.sample_first_hitting(cfg, P, buffers, s["x_init"], s["timesteps"], float(s["dt"]), us, pos, x0=x0,
                                                                x0_unmask=x0_unmask, modality=modality, batch=batch)
    assert torch.equal(schedule, s["schedule"].to(schedule.dtype))
    for i in range(steps):
        assert torch.equal(xs[i], s[f"step{i}/x_next"]), f"step {i}"
    assert torch.equal(x_final, s["x_final"]) and nfe == int(s["nfe"])


def test_oracle_sampler_with_attention_caching_matches_reference_tokens():
    """eval.attention_caching (full / cache-building / text-only steps, model_eval.py:2296-2366) replayed from the uniforms of a reference run."""
    g = Golden("c_large")
    s = load_sampler("c_large_attn_caching")
    cfg, P, buffers, batch = oracle_setup(g)
    steps, ratio = int(s["steps"]), int(s["ratio"])
    us = [s[f"step{i}/u"] for i in range(steps)]
    x_final, xs, x_last, nfe, modes = O.sample_ddpm_cache_attention_caching(cfg, P, buffers, s["x_init"], s["timesteps"], float(s["dt"]), us, ratio,
                                                                            modality=s["modality"], batch=batch)
    z = np.load(os.path.join(GOLDEN_DIR, "sampler_c_large_attn_caching.npz"))
    assert modes == [str(z[f"step{i}/mode"]) for i in range(steps)]
    assert {"full", "build", "text"} <= set(modes)
    for i in range(steps):
        assert xs[i].shape == s[f"step{i}/x_next"].shape and torch.equal(xs[i], s[f"step{i}/x_next"]), f"step {i}"
    assert torch.equal(x_last, s["x_before_noise_removal"]) and torch.equal(x_final, s["x_final"])
    assert nfe == int(s["nfe"])
    # the three kinds of forward agree with the reference's probabilities to fp32 accuracy (step 1 = image queries masked from text keys,
    # step 2 = the text slice alone)
    B, Lt = s["x_init"].shape[0], cfg.txt_length
    sig = lambda i: O.loglinear_noise(s["timesteps"][i] * torch.ones(B))[0]
    allow = O.modality_dropout_mask(torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool), Lt, s["x_init"].shape[1])
    p1, _ = O.ddpm_forward(cfg, P, buffers, s["step1/x"], sig(1), s["modality"], batch, allow_mask=allow)
    assert torch.allclose(p1, s["step1/p_x0"], atol=2e-6, rtol=1e-4)
    p1_nomask, _ = O.ddpm_forward(cfg, P, buffers, s["step1/x"], sig(1), s["modality"], batch)
    assert not torch.allclose(p1_nomask, s["step1/p_x0"], atol=1e-4, rtol=1e-3)   # the mask matters
    p2, _ = O.ddpm_forward(cfg, P, buffers, s["step2/x"], sig(2), s["modality"][:, :Lt], batch)
    assert torch.allclose(p2, s["step2/p_x0"], atol=2e-6, rtol=1e-4)
