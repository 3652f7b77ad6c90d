"""the fused sampler step (udm_ddpm_sample_rows) and the `ddpm_cache` loop of unidisc_amd.Diffusion against the oracle and
the golden vectors recorded from the imported reference (oracle/make_golden_sampler.py)."""
import os

import numpy as np
import pytest
import torch

import fake_kernels
from golden_utils import GOLDEN_DIR, Golden
from oracle import unidisc_oracle as O
from product_utils import build_product

DEV = "cuda"

# Token agreement of a product sampling loop with the recorded reference run (bf16 backbone / kernel doubles vs the fp32 reference: a near-tie draw may flip and
# the flip propagates).  Recorded in the parity ledger; the bound per test is the disagreement achieved (profiles/r04_parity_ledger.json) x 3 + 2 % (was a flat 10 %).
AGREE_BOUNDS = {   # achieved 0.0078 (one token of 128) on CPU doubles and on the GPU; every other loop: 0.0 -> the default of 2 % (two tokens of the small cases)
    "test_first_hitting_host_logic_replays_reference_run[b_small]": 0.045,
    "test_first_hitting_loop_on_gpu[b_small]": 0.045,
    "test_attention_caching_sampler_host_logic_replays_reference_run": 0.045,
    "test_attention_caching_sampler_loop_on_gpu": 0.045,
}


def _agree(test, case, x, ref):
    from ledger import check

    key = test if case is None else f"{test}[{case}]"
    dis = 1.0 - (x.cpu() == ref.cpu()).float().mean().item()
    check(key, "token_disagreement_vs_reference_run", dis, AGREE_BOUNDS.get(key, 0.02))



def load_sampler(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"sampler_{name}.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if z[k].dtype.kind != "U"}   # (string entries: read them from the npz directly)


def _oracle_tokens_from_logits(cfg, logits_bf16, x, t, dt, u, modality, batch):
    """The reference update on GIVEN logits (model.py:621-658 + model_eval.py:2090-2096) in fp32 on the CPU."""
    lp = O.subs_parameterization(cfg, logits_bf16.float(), x, modality, batch, bf16=False).float()
    p = lp.exp()
    q = p * (t[:, None, None] - (t - dt)[:, None, None])
    q[:, :, cfg.mask_index] = (t - dt)[:, None]
    tok = O.sample_categorical(q, u)
    keep = x != cfg.mask_index
    return torch.where(keep, x, tok), lp


@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_sampler_loop_host_logic_replays_reference_run(name, monkeypatch):
    """CPU: Diffusion.sample with kernel doubles, fed the uniforms the reference drew, reproduces the reference's tokens step by step."""
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g, s = Golden(name), load_sampler(name)
    diff = build_product(g, device="cpu")
    diff.backbone.eval()
    steps = int(s["steps"])
    noise = [s[f"step{i}/u"] for i in range(steps)]
    modality = s["modality"] if "modality" in s else None
    x0, x0_unmask = (s["x0"], s["x0_unmask"].bool()) if "x0" in s else (None, None)
    B, L = s["x_init"].shape
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=B, modality=modality, noise=noise, return_nfe=True)
    _agree("test_sampler_loop_host_logic_replays_reference_run", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1  # + the noise-removal forward
    assert not (x == diff.mask_index).any()
    if x0 is not None:
        assert torch.equal(x[x0_unmask], x0[x0_unmask])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_sample_rows_kernel_matches_oracle_on_reference_logits(name):
    """GPU kernel parity, token-exact: the reference's own logits (rounded to bf16) and uniforms of every recorded step go through
    udm_ddpm_sample_rows; the oracle applies the reference's update to the same bf16 logits on the CPU."""
    from unidisc_amd import kernels as K

    g, s = Golden(name), load_sampler(name)
    cfg = g.cfg
    batch = O.update_batch(cfg, g.batch())
    modality = s["modality"] if "modality" in s else None
    dt = float(s["dt"])
    V, Vt, mask = cfg.vocab_size, cfg.text_vocab_size, cfg.mask_index
    for i in range(int(s["steps"])):
        if f"step{i}/logits" not in s:
            continue
        x, u = s[f"step{i}/x"], s[f"step{i}/u"]
        B, L = x.shape
        t = s["timesteps"][i] * torch.ones(B)
        lb = s[f"step{i}/logits"].bfloat16()
        want, lp = _oracle_tokens_from_logits(cfg, lb, x, t, dt, u, modality, batch)
        rows = (x.reshape(-1) == mask).nonzero().reshape(-1)
        Vp = (V + 7) // 8 * 8
        lg = torch.zeros((rows.numel(), Vp), dtype=torch.bfloat16)
        lg[:, :V] = lb.reshape(B * L, V)[rows]
        b_of = rows // L
        rm = None
        if cfg.force_argmax_valid_indices:
            mod = modality if modality is not None else torch.cat([torch.zeros(B, cfg.txt_length), torch.ones(B, L - cfg.txt_length)], 1).long()
            rm = mod.reshape(-1)[rows].long().to(DEV)
        tok = K.ddpm_sample_rows(lg.to(DEV), V, Vt, mask, t=t[b_of].to(DEV), s=(t - dt)[b_of].to(DEV), modality=rm,
                                 restrict=cfg.force_argmax_valid_indices, u=u.reshape(B * L, V)[rows].contiguous().to(DEV)).cpu()
        assert torch.equal(tok, want.reshape(-1)[rows]), f"step {i}"
        greedy = K.ddpm_sample_rows(lg.to(DEV), V, Vt, mask, modality=rm, restrict=cfg.force_argmax_valid_indices, greedy=True).cpu()
        assert torch.equal(greedy, lp.argmax(-1).reshape(-1)[rows]), f"greedy step {i}"


@pytest.mark.gpu
def test_sample_rows_philox_draws_follow_the_step_distribution():
    """Philox path: 40 000 independent rows with the same logits; empirical frequencies match q = p (t - s) on tokens, s on [MASK]."""
    from unidisc_amd import kernels as K

    V, Vt, mask = 24, 24, 23
    z = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5] + [-3.0] * (V - 5))
    R = 40000
    lg = z.bfloat16()[None].repeat(R, 1).contiguous().to(DEV)
    t, s = torch.full((R,), 0.6, device=DEV), torch.full((R,), 0.45, device=DEV)
    tok = K.ddpm_sample_rows(lg, V, Vt, mask, t=t, s=s, seed=123).cpu()
    zz = z.bfloat16().float().clone()
    zz[mask] = float("-inf")
    q = torch.softmax(zz, -1) * 0.15
    q[mask] = 0.45
    want = q / q.sum()
    freq = torch.bincount(tok, minlength=V).float() / R
    assert torch.allclose(freq, want, atol=0.01), (freq, want)
    tok2 = K.ddpm_sample_rows(lg, V, Vt, mask, t=t, s=s, seed=124).cpu()
    assert not torch.equal(tok, tok2)
    assert torch.equal(tok, K.ddpm_sample_rows(lg, V, Vt, mask, t=t, s=s, seed=123).cpu())  # reproducible for a seed


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_sampler_loop_on_gpu(name):
    g, s = Golden(name), load_sampler(name)
    diff = build_product(g, device=DEV)
    diff.backbone.eval()
    steps = int(s["steps"])
    noise = [s[f"step{i}/u"].to(DEV) for i in range(steps)]
    modality = s["modality"].to(DEV) if "modality" in s else None
    x0, x0_unmask = (s["x0"].to(DEV), s["x0_unmask"].bool().to(DEV)) if "x0" in s else (None, None)
    B, L = s["x_init"].shape
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=B, modality=modality, noise=noise, return_nfe=True)
    x = x.cpu()
    _agree("test_sampler_loop_on_gpu", locals().get("name"), x, s["x_final"])
    assert not (x == diff.mask_index).any() and nfe == int(s["nfe"]) + 1
    if x0 is not None:
        assert torch.equal(x[s["x0_unmask"].bool()], s["x0"][s["x0_unmask"].bool()])
    # Philox-driven run: complete, reproducible for a seed, different across seeds
    a = diff.sample(num_steps=steps, x0=x0, x0_unmask=x0_unmask, batch_size=B, modality=modality, seed=5)
    b = diff.sample(num_steps=steps, x0=x0, x0_unmask=x0_unmask, batch_size=B, modality=modality, seed=5)
    c = diff.sample(num_steps=steps, x0=x0, x0_unmask=x0_unmask, batch_size=B, modality=modality, seed=6)
    assert torch.equal(a, b) and not torch.equal(a, c) and not (a == diff.mask_index).any()


# ------------------------------------------------------------------------------------------------ classifier-free guidance (config.eval.cfg)
def _cfg_product(device):
    g, s = Golden("c_large"), load_sampler("c_large_cfg")
    diff = build_product(g, device=device)
    diff.backbone.eval()
    from unidisc_amd.config import Cfg
    diff.config.eval = Cfg(cfg=2.0)
    return g, s, diff


def test_guided_sampler_host_logic_replays_reference_run(monkeypatch):
    """CPU: the guided loop (one [x ; x_uncond] backbone pass per step, mix inside the row kernel) with kernel doubles replays the reference's
    CFG run (`_ddpm_forward` CFG branch) from the recorded uniforms."""
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g, s, diff = _cfg_product("cpu")
    steps = int(s["steps"])
    noise = [s[f"step{i}/u"] for i in range(steps)]
    x0, x0_unmask = s["x0"], s["x0_unmask"].bool()
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=x0.shape[0], modality=s["modality"], noise=noise,
                         return_nfe=True)
    _agree("test_guided_sampler_host_logic_replays_reference_run", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any() and torch.equal(x[x0_unmask], x0[x0_unmask])
    # the guidance weight is the reference's
    t = s["timesteps"][2] * torch.ones(x0.shape[0])
    assert torch.allclose(diff.get_cfg_weight(t), s["step2/cfg_w"], atol=1e-7)
    # and it matters: without config.eval.cfg the same uniforms give other tokens
    diff.config.eval.cfg = None
    y = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=x0.shape[0], modality=s["modality"], noise=noise)
    assert not torch.equal(x, y)


def test_guided_sampler_split_cfg_batches_is_the_same_sampler(monkeypatch):
    """config.eval.split_cfg_batches (model_eval.py:1770-1784): two backbone passes instead of one over [x ; x_uncond]; rows of a batch never
    interact, so every token of the replayed run is the same."""
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g, s, diff = _cfg_product("cpu")
    steps = int(s["steps"])
    noise = [s[f"step{i}/u"] for i in range(steps)]
    x0, x0_unmask = s["x0"], s["x0_unmask"].bool()
    kw = dict(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=x0.shape[0], modality=s["modality"], noise=noise, return_nfe=True)
    x1, nfe1 = diff.sample(**kw)
    diff.config.eval.split_cfg_batches = True
    x2, nfe2 = diff.sample(**kw)
    assert torch.equal(x1, x2) and nfe1 == nfe2


@pytest.mark.gpu
def test_guided_sampler_split_cfg_batches_on_gpu():
    g, s, diff = _cfg_product(DEV)
    steps = int(s["steps"])
    noise = [s[f"step{i}/u"].to(DEV) for i in range(steps)]
    x0, x0_unmask = s["x0"].to(DEV), s["x0_unmask"].bool().to(DEV)
    kw = dict(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=x0.shape[0], modality=s["modality"].to(DEV), noise=noise)
    x1 = diff.sample(**kw)
    diff.config.eval.split_cfg_batches = True
    x2 = diff.sample(**kw)
    assert (x1 == x2).float().mean().item() >= 0.99      # same kernels per row; a different GEMM tile family may flip a near-tie draw


@pytest.mark.gpu
def test_guided_sample_rows_kernel_matches_oracle_on_reference_logits():
    """Token-exact: both logits halves of the reference's CFG run (rounded to bf16), its weights and uniforms through udm_ddpm_sample_rows_cfg;
    the oracle mixes the same bf16 logits in fp32 and applies SUBS (xt=None) + the update on the CPU."""
    from unidisc_amd import kernels as K

    g, s = Golden("c_large"), load_sampler("c_large_cfg")
    cfg = g.cfg
    batch = O.update_batch(cfg, g.batch())
    modality, dt = s["modality"], float(s["dt"])
    V, Vt, mask = cfg.vocab_size, cfg.text_vocab_size, cfg.mask_index
    checked = 0
    for i in range(int(s["steps"])):
        if f"step{i}/logits_uncond" not in s or not bool((s[f"step{i}/cfg_w"] > 0).any()):
            continue
        x, u, w = s[f"step{i}/x"], s[f"step{i}/u"], s[f"step{i}/cfg_w"].reshape(-1)
        B, L = x.shape
        t = s["timesteps"][i] * torch.ones(B)
        lc, lu = s[f"step{i}/logits"].bfloat16(), s[f"step{i}/logits_uncond"].bfloat16()
        mixed = (1 + w[:, None, None]) * lc.float() - w[:, None, None] * lu.float()
        lp = O.subs_parameterization(cfg, mixed, None, modality, batch, bf16=False).float()
        q = lp.exp() * dt
        q[:, :, mask] = (t - dt)[:, None]
        want = torch.where(x != mask, x, O.sample_categorical(q, u))
        rows = (x.reshape(-1) == mask).nonzero().reshape(-1)
        Vp = (V + 7) // 8 * 8
        pad = lambda z: torch.cat([z.reshape(B * L, V)[rows], torch.zeros(rows.numel(), Vp - V, dtype=z.dtype)], 1).contiguous()
        b_of = rows // L
        rm = modality.reshape(-1)[rows].long().to(DEV) if cfg.force_argmax_valid_indices else None
        tok = K.ddpm_sample_rows(pad(lc).to(DEV), V, Vt, mask, t=t[b_of].to(DEV), s=(t - dt)[b_of].to(DEV), modality=rm, restrict=cfg.force_argmax_valid_indices,
                                 u=u.reshape(B * L, V)[rows].contiguous().to(DEV), logits_u=pad(lu).to(DEV), w=w[b_of].float().contiguous().to(DEV)).cpu()
        assert torch.equal(tok, want.reshape(-1)[rows]), f"step {i}"
        plain = K.ddpm_sample_rows(pad(lc).to(DEV), V, Vt, mask, t=t[b_of].to(DEV), s=(t - dt)[b_of].to(DEV), modality=rm, restrict=cfg.force_argmax_valid_indices,
                                   u=u.reshape(B * L, V)[rows].contiguous().to(DEV)).cpu()
        checked += int(not torch.equal(plain, tok))
    assert checked >= 1   # guidance changed at least one step's draw
    with pytest.raises(ValueError):
        K.ddpm_sample_rows(pad(lc).to(DEV), V, Vt, mask, greedy=True, logits_u=pad(lu).to(DEV))


@pytest.mark.gpu
def test_guided_sampler_loop_on_gpu():
    g, s, diff = _cfg_product(DEV)
    steps = int(s["steps"])
    noise = [s[f"step{i}/u"].to(DEV) for i in range(steps)]
    x0, x0_unmask = s["x0"].to(DEV), s["x0_unmask"].bool().to(DEV)
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=x0.shape[0], modality=s["modality"].to(DEV), noise=noise,
                         return_nfe=True)
    x = x.cpu()
    _agree("test_guided_sampler_loop_on_gpu", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    assert torch.equal(x[s["x0_unmask"].bool()], s["x0"][s["x0_unmask"].bool()])


# ------------------------------------------------------------------------------------------------ `maskgit` predictor
def _maskgit_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"maskgit_{name}.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if z[k].dtype.kind != "U"}   # (string entries: read them from the npz directly)


def _maskgit_run(diff, s, device):
    steps = int(s["steps"])
    from unidisc_amd.config import Cfg
    diff.config.eval = Cfg(maskgit_r_temp=float(s["r_temp"]))
    replay = [(s[f"step{i}/pred"].to(device), s[f"step{i}/gumbel"].float().to(device)) if f"step{i}/pred" in s else (None, None) for i in range(steps)]
    modality = s["modality"].to(device) if "modality" in s else None
    x0, x0_unmask = (s["x0"].to(device), s["x0_unmask"].bool().to(device)) if "x0" in s else (None, None)
    B = s["x_init"].shape[0]
    return diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=B, modality=modality, predictor="maskgit", replay=replay,
                       return_nfe=True)


@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_maskgit_host_logic_replays_reference_run(name, monkeypatch):
    """CPU: the maskgit loop with kernel doubles, fed the reference's multinomial draws and Gumbel noise, reveals the reference's tokens."""
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g, s = Golden(name), _maskgit_golden(name)
    diff = build_product(g, device="cpu")
    diff.backbone.eval()
    assert torch.equal(diff.adap_sche(s["x_init"], int(s["steps"]), diff.mask_index), s["schedule"].to(torch.int32))
    x, nfe = _maskgit_run(diff, s, "cpu")
    _agree("test_maskgit_host_logic_replays_reference_run", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    if "x0" in s:
        assert torch.equal(x[s["x0_unmask"].bool()], s["x0"][s["x0_unmask"].bool()])
    with pytest.raises(NotImplementedError):
        diff.sample(num_steps=2, batch_size=2, predictor="ddpm_tweedie")


def _nucleus_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "maskgit_nucleus_c_large.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if z[k].dtype.kind != "U"}   # (string entries: read them from the npz directly)


def test_nucleus_filter_oracle_matches_reference_draw_support():
    """The oracle's restatement of `nucleus_sampling_batch` on the reference's own p_x0: every token the reference drew lies in the kept set, the kept
    set is the top of the distribution with cumulative p / temperature <= top_p (+ the arg-max), and the filtered distribution is normalised."""
    s = _nucleus_golden()
    top_p, temp = float(s["top_p"]), float(s["temperature"])
    seen = 0
    for i in range(int(s["steps"])):
        if f"step{i}/p_x0" not in s:
            continue
        p, pred, x = s[f"step{i}/p_x0"].float(), s[f"step{i}/pred"], s[f"step{i}/x"]
        fp = O.nucleus_filter(p, top_p, temp)
        assert torch.allclose(fp.sum(-1), torch.ones_like(fp.sum(-1)), atol=1e-5)
        masked = x == int(Golden("c_large").cfg.mask_index)
        assert bool((fp.gather(-1, pred[..., None]).squeeze(-1)[masked] > 0).all())          # the reference's draws come from the kept set
        kept = fp > 0
        assert bool(kept.gather(-1, p.argmax(-1, keepdim=True)).all())                          # the arg-max always stays
        worst_kept = torch.where(kept, p, torch.full_like(p, 2.0)).min(-1).values
        best_dropped = torch.where(kept, torch.zeros_like(p), p).max(-1).values
        assert bool((worst_kept >= best_dropped).all())                                         # a prefix of the sorted order
        # at least the rows whose top-1 probability alone is below the budget keep a cumulative mass of no more than top_p * temperature
        cm = torch.where(kept, p, torch.zeros_like(p)).sum(-1)
        ok = (cm <= top_p * temp + 1e-5) | (kept.sum(-1) == 1)
        assert bool(ok.all())
        seen += 1
    assert seen >= 3


def test_maskgit_nucleus_host_logic_replays_reference_run(monkeypatch):
    """CPU: the maskgit_nucleus loop (batch size 1: the reference's early exit is a tensor truth test) with kernel doubles, fed the reference's nucleus
    draws and Gumbel noise, reveals the reference's tokens; free-running draws stay inside the kept set."""
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod
    from unidisc_amd.config import Cfg

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g, s = Golden("c_large"), _nucleus_golden()
    diff = build_product(g, device="cpu")
    diff.backbone.eval()
    steps = int(s["steps"])
    diff.config.eval = Cfg(maskgit_r_temp=float(s["r_temp"]), top_p=float(s["top_p"]), temperature=float(s["temperature"]))
    replay = [(s[f"step{i}/pred"], s[f"step{i}/gumbel"].float()) if f"step{i}/pred" in s else (None, None) for i in range(steps)]
    x0, x0_unmask = s["x0"], s["x0_unmask"].bool()
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=1, modality=s["modality"], predictor="maskgit_nucleus",
                         replay=replay, return_nfe=True)
    _agree("test_maskgit_nucleus_host_logic_replays_reference_run", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    assert torch.equal(x[x0_unmask], x0[x0_unmask])
    # free draw of one step: tokens come from the nucleus of the step's own distribution
    xs, t = s["step1/x"], s["timesteps"][1] * torch.ones(1, 1)
    logits, rows, n = diff.backbone.forward_masked_logits(xs, None, modality=s["modality"])
    rm = diff._row_modality(rows[:n], 1, xs.shape[1], s["modality"])
    tok = diff._nucleus_draw(logits[:n], None, None, rm, float(s["top_p"]), float(s["temperature"]), seed=5)
    lp = O.subs_parameterization(g.cfg, logits[:n, : g.cfg.vocab_size].float()[None], torch.full((1, n), g.cfg.mask_index), rm[None], None).float()[0]
    fp = O.nucleus_filter(lp.exp(), float(s["top_p"]), float(s["temperature"]))
    assert bool((fp.gather(-1, tok[:, None]) > 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_categorical_rows_kernel_logp_matches_oracle(name):
    """udm_categorical_sample_rows with replayed tokens: log p(token) equals the oracle's SUBS log-prob of the same bf16 logits (fp32 rounding)."""
    from unidisc_amd import kernels as K

    g, s = Golden(name), _maskgit_golden(name)
    cfg = g.cfg
    batch = O.update_batch(cfg, g.batch())
    modality = s["modality"] if "modality" in s else None
    V, Vt, mask = cfg.vocab_size, cfg.text_vocab_size, cfg.mask_index
    for i in range(int(s["steps"])):
        if f"step{i}/pred" not in s:
            continue
        x, pred = s[f"step{i}/x"], s[f"step{i}/pred"]
        B, L = x.shape
        lb = s[f"step{i}/logits"].bfloat16()
        lp = O.subs_parameterization(cfg, lb.float(), x, modality, batch, bf16=False).float()
        rows = (x.reshape(-1) == mask).nonzero().reshape(-1)
        Vp = (V + 7) // 8 * 8
        lg = torch.zeros((rows.numel(), Vp), dtype=torch.bfloat16)
        lg[:, :V] = lb.reshape(B * L, V)[rows]
        rm = None
        if cfg.force_argmax_valid_indices:
            mod = modality if modality is not None else torch.cat([torch.zeros(B, cfg.txt_length), torch.ones(B, L - cfg.txt_length)], 1).long()
            rm = mod.reshape(-1)[rows].long().to(DEV)
        given = pred.reshape(-1)[rows].contiguous()
        tok, logp = K.categorical_sample_rows(lg.to(DEV), V, Vt, mask, modality=rm, restrict=cfg.force_argmax_valid_indices, given=given.to(DEV))
        assert torch.equal(tok.cpu(), given)
        want = lp.reshape(B * L, V)[rows].gather(-1, given[:, None]).squeeze(-1)
        assert torch.allclose(logp.cpu(), want, atol=2e-5, rtol=1e-5), f"step {i}"


@pytest.mark.gpu
def test_categorical_rows_kernel_draws_follow_p():
    from unidisc_amd import kernels as K

    V, Vt, mask = 24, 24, 23
    z = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5] + [-3.0] * (V - 5))
    R = 40000
    lg = z.bfloat16()[None].repeat(R, 1).contiguous().to(DEV)
    tok, logp = K.categorical_sample_rows(lg, V, Vt, mask, seed=11)
    zz = z.bfloat16().float().clone()
    zz[mask] = float("-inf")
    want = torch.softmax(zz, -1)
    freq = torch.bincount(tok.cpu(), minlength=V).float() / R
    assert torch.allclose(freq, want, atol=0.01) and freq[mask] == 0
    assert torch.allclose(logp.cpu(), torch.log_softmax(zz, -1)[tok.cpu()], atol=1e-5)
    tok2, _ = K.categorical_sample_rows(lg, V, Vt, mask, seed=12)
    assert not torch.equal(tok, tok2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_maskgit_loop_on_gpu(name):
    g, s = Golden(name), _maskgit_golden(name)
    diff = build_product(g, device=DEV)
    diff.backbone.eval()
    x, nfe = _maskgit_run(diff, s, DEV)
    x = x.cpu()
    _agree("test_maskgit_loop_on_gpu", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    # free-running (Philox draws + device Gumbel noise): complete, reproducible for a seed
    modality = s["modality"].to(DEV) if "modality" in s else None
    B = s["x_init"].shape[0]
    a = diff.sample(num_steps=int(s["steps"]), batch_size=B, modality=modality, predictor="maskgit", seed=3)
    b = diff.sample(num_steps=int(s["steps"]), batch_size=B, modality=modality, predictor="maskgit", seed=3)
    c = diff.sample(num_steps=int(s["steps"]), batch_size=B, modality=modality, predictor="maskgit", seed=4)
    assert torch.equal(a, b) and not torch.equal(a, c) and not (a == diff.mask_index).any()


@pytest.mark.gpu
def test_maskgit_nucleus_loop_on_gpu():
    from unidisc_amd.config import Cfg

    g, s = Golden("c_large"), _nucleus_golden()
    diff = build_product(g, device=DEV)
    diff.backbone.eval()
    steps = int(s["steps"])
    diff.config.eval = Cfg(maskgit_r_temp=float(s["r_temp"]), top_p=float(s["top_p"]), temperature=float(s["temperature"]))
    replay = [(s[f"step{i}/pred"].to(DEV), s[f"step{i}/gumbel"].float().to(DEV)) if f"step{i}/pred" in s else (None, None) for i in range(steps)]
    x0, x0_unmask, mod = s["x0"].to(DEV), s["x0_unmask"].bool().to(DEV), s["modality"].to(DEV)
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=1, modality=mod, predictor="maskgit_nucleus",
                         replay=replay, return_nfe=True)
    _agree("test_maskgit_nucleus_loop_on_gpu", locals().get("name"), x.cpu(), s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    # free-running on a batch (the generalisation `all(num_unmask <= 0)` of the reference's batch-1 early exit): complete, reproducible per seed
    B = 4
    mod4 = mod.expand(B, -1).contiguous()
    a = diff.sample(num_steps=steps, batch_size=B, modality=mod4, predictor="maskgit_nucleus", seed=3)
    b = diff.sample(num_steps=steps, batch_size=B, modality=mod4, predictor="maskgit_nucleus", seed=3)
    c = diff.sample(num_steps=steps, batch_size=B, modality=mod4, predictor="maskgit_nucleus", seed=4)
    assert torch.equal(a, b) and not torch.equal(a, c) and not (a == diff.mask_index).any()


# ------------------------------------------------------------------------------------------------ `first_hitting` predictor
def _fh_run(diff, name, device):
    z = np.load(os.path.join(GOLDEN_DIR, f"first_hitting_{name}.npz"))
    s = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    steps = int(s["steps"])
    replay = [(s[f"step{i}/u"].to(device), s[f"step{i}/pos_u"].to(device) if f"step{i}/pos_u" in s else None) for i in range(steps)]
    modality = s["modality"].to(device) if "modality" in s else None
    x0, x0_unmask = (s["x0"].to(device), s["x0_unmask"].bool().to(device)) if "x0" in s else (None, None)
    out = diff.sample(num_steps=steps, eps=float(s["eps"]), x0=x0, x0_unmask=x0_unmask, batch_size=s["x_init"].shape[0], modality=modality,
                      predictor="first_hitting", replay=replay, return_nfe=True)
    return s, out


@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_first_hitting_host_logic_replays_reference_run(name, monkeypatch):
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    diff = build_product(Golden(name), device="cpu")
    diff.backbone.eval()
    s, (x, nfe) = _fh_run(diff, name, "cpu")
    assert torch.equal(diff.adap_sche(s["x_init"], int(s["steps"]), diff.mask_index, "linear"), s["schedule"].to(torch.int32))
    _agree("test_first_hitting_host_logic_replays_reference_run", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    # the revealed POSITIONS are exact at every step (integer lottery); only near-tie token draws may differ under bf16
    if "x0" in s:
        assert torch.equal(x[s["x0_unmask"].bool()], s["x0"][s["x0_unmask"].bool()])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_first_hitting_loop_on_gpu(name):
    diff = build_product(Golden(name), device=DEV)
    diff.backbone.eval()
    s, (x, nfe) = _fh_run(diff, name, DEV)
    x = x.cpu()
    _agree("test_first_hitting_loop_on_gpu", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    modality = s["modality"].to(DEV) if "modality" in s else None
    B = s["x_init"].shape[0]
    a = diff.sample(num_steps=int(s["steps"]), batch_size=B, modality=modality, predictor="first_hitting", seed=3)
    b = diff.sample(num_steps=int(s["steps"]), batch_size=B, modality=modality, predictor="first_hitting", seed=3)
    assert torch.equal(a, b) and not (a == diff.mask_index).any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c_large", "b_small"])
def test_categorical_rows_kernel_token_exact_on_reference_uniforms(name):
    """`_sample_categorical(p_x0)` of the first-hitting update: the reference's logits (rounded to bf16) and its [B, L, V] uniforms through
    udm_categorical_sample_rows give exactly the tokens the oracle's fp32 restatement picks from the same bf16 logits."""
    from unidisc_amd import kernels as K

    g = Golden(name)
    z = np.load(os.path.join(GOLDEN_DIR, f"first_hitting_{name}.npz"))
    s = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    cfg = g.cfg
    batch = O.update_batch(cfg, g.batch())
    modality = s["modality"] if "modality" in s else None
    V, Vt, mask = cfg.vocab_size, cfg.text_vocab_size, cfg.mask_index
    for i in range(int(s["steps"])):
        x, u = s[f"step{i}/x"], s[f"step{i}/u"]
        B, L = x.shape
        lb = s[f"step{i}/logits"].bfloat16()
        p = O.subs_parameterization(cfg, lb.float(), x, modality, batch, bf16=False).float().exp()
        want = O.sample_categorical(p, u)
        rows = (x.reshape(-1) == mask).nonzero().reshape(-1)
        if rows.numel() == 0:
            continue
        Vp = (V + 7) // 8 * 8
        lg = torch.zeros((rows.numel(), Vp), dtype=torch.bfloat16)
        lg[:, :V] = lb.reshape(B * L, V)[rows]
        rm = None
        if cfg.force_argmax_valid_indices:
            mod = modality if modality is not None else torch.cat([torch.zeros(B, cfg.txt_length), torch.ones(B, L - cfg.txt_length)], 1).long()
            rm = mod.reshape(-1)[rows].long().to(DEV)
        tok, _ = K.categorical_sample_rows(lg.to(DEV), V, Vt, mask, modality=rm, restrict=cfg.force_argmax_valid_indices,
                                           u=u.reshape(B * L, V)[rows].contiguous().to(DEV))
        assert torch.equal(tok.cpu(), want.reshape(-1)[rows]), f"step {i}"


# ------------------------------------------------------------------------------------------------ eval.attention_caching (model_eval.py:2296-2366)
def _caching_product(device):
    g, s = Golden("c_large"), load_sampler("c_large_attn_caching")
    diff = build_product(g, device=device)
    diff.backbone.eval()
    from unidisc_amd.config import Cfg
    diff.config.eval = Cfg(cfg=None, attention_caching=True, attention_caching_txt_to_img_ratio=int(s["ratio"]))
    z = np.load(os.path.join(GOLDEN_DIR, "sampler_c_large_attn_caching.npz"))
    modes = [str(z[f"step{i}/mode"]) for i in range(int(s["steps"]))]
    return g, s, diff, modes


def test_attention_caching_sampler_host_logic_replays_reference_run(monkeypatch):
    """CPU: the three kinds of step (full / image queries on image keys only / text slice alone) with kernel doubles, fed the reference's uniforms"""
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g, s, diff, modes = _caching_product("cpu")
    steps = int(s["steps"])
    B, L = s["x_init"].shape
    seen = []
    orig = diff._ddpm_caching_update

    def spy(x, t, dt, **kw):
        out = orig(x, t, dt, **kw)
        seen.append((tuple(x.shape), out[1].clone()))
        return out

    diff._ddpm_caching_update = spy
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), batch_size=B, modality=s["modality"], noise=[s[f"step{i}/u"] for i in range(steps)], return_nfe=True)
    assert diff.sample_step_modes == modes and {"full", "build", "text"} <= set(modes)
    for i in range(steps):   # every step ran in the reference's view (full sequence or the text slice) and mostly drew the reference's tokens
        assert seen[i][0] == tuple(s[f"step{i}/x"].shape), i
    _agree("test_attention_caching_sampler_host_logic_replays_reference_run", locals().get("name"), x, s["x_final"])
    assert nfe == int(s["nfe"]) + 1 and not (x == diff.mask_index).any()
    assert x.shape == (B, L)


def test_attention_caching_cache_moves_between_views():
    """the logits cache of the [MASK] rows follows the reference's p_x0 slicing: to the text slice and back"""
    from unidisc_amd.diffusion import Diffusion
    L, Lt, B = 6, 2, 2
    rows = torch.tensor([0, 3, 7, 8, 11])          # b0: l = 0, 3; b1: l = 1, 2, 5
    logits = torch.arange(5, dtype=torch.float32)[:, None].repeat(1, 4)
    cache = (torch.cat([logits, torch.zeros(3, 4)]), torch.cat([rows, torch.tensor([1, 2, 4])]), 5)   # padded to 8 rows like the product's
    tl, tr, tn = Diffusion._cache_to_text(cache, L, Lt)
    assert tn == 2 and tr.tolist() == [0, 3] and tl[:, 0].tolist() == [0.0, 2.0]      # (b0, l0) -> 0, (b1, l1) -> 1 * Lt + 1
    new_text = (torch.full((1, 4), 9.0), torch.tensor([3]), 1)                          # the text slice was re-evaluated: one [MASK] left at (b1, l1)
    fl, fr, fn = Diffusion._cache_to_full(cache, new_text, L, Lt)
    assert fn == 4 and fr.tolist() == [3, 8, 11, 7] and fl[:, 0].tolist() == [1.0, 3.0, 4.0, 9.0]
    assert Diffusion._cache_to_full(None, new_text, L, Lt) is None and Diffusion._cache_to_full(cache, None, L, Lt) is None
    assert Diffusion._cache_to_text(None, L, Lt) is None


@pytest.mark.gpu
def test_attention_caching_sampler_loop_on_gpu():
    g, s, diff, modes = _caching_product(DEV)
    steps = int(s["steps"])
    B, L = s["x_init"].shape
    x, nfe = diff.sample(num_steps=steps, eps=float(s["eps"]), batch_size=B, modality=s["modality"].to(DEV), noise=[s[f"step{i}/u"].to(DEV) for i in range(steps)],
                         return_nfe=True)
    assert diff.sample_step_modes == modes
    x = x.cpu()
    _agree("test_attention_caching_sampler_loop_on_gpu", locals().get("name"), x, s["x_final"])
    assert not (x == diff.mask_index).any() and nfe == int(s["nfe"]) + 1
    a = diff.sample(num_steps=steps, batch_size=B, modality=s["modality"].to(DEV), seed=5)
    b = diff.sample(num_steps=steps, batch_size=B, modality=s["modality"].to(DEV), seed=5)
    assert torch.equal(a, b) and not (a == diff.mask_index).any()
    # the kernels' view of the cache-building step: image queries masked from text keys changes the image rows' logits, not the text rows' inputs
    from unidisc_amd.dit import ModalityMask
    xq = s["step1/x"].to(DEV)
    sig = diff._process_sigma(diff.noise(s["timesteps"][1].to(DEV) * torch.ones(B, device=DEV))[0])
    bm = ModalityMask(torch.zeros(B, dtype=torch.bool, device=DEV), torch.ones(B, dtype=torch.bool, device=DEV), g.case["txt_length"])
    lg_m, rows_m, n_m = diff.backbone.forward_masked_logits(xq, sig, modality=s["modality"].to(DEV), block_mask=bm)
    lg, rows, n = diff.backbone.forward_masked_logits(xq, sig, modality=s["modality"].to(DEV))
    assert n == n_m and torch.equal(rows, rows_m) and not torch.equal(lg[:n], lg_m[:n])
