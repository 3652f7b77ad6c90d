"""FusedAdamW (+ global-norm clipping, bf16 shadow maintenance) against torch.optim.AdamW + clip_grad_norm_, which is what
the reference's loop runs (model_setup.py:385-424, model.py:1516-1545).  CPU part: the host logic with kernel doubles; GPU part: the HIP
kernels themselves."""
import copy

import pytest
import torch

import fake_kernels
from golden_utils import Golden, rel_err
from product_utils import build_product

DEV = "cuda"


def _torch_reference_steps(params, grads_per_step, lr, betas, eps, wd, max_norm):
    ps = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = torch.optim.AdamW(ps, lr=lr, betas=betas, eps=eps, weight_decay=wd)
    norms = []
    for grads in grads_per_step:
        for p, g in zip(ps, grads):
            p.grad = g.clone()
        if max_norm is not None:
            norms.append(torch.nn.utils.clip_grad_norm_(ps, max_norm))
        opt.step()
    return [p.detach() for p in ps], norms


@pytest.fixture()
def fake_k(monkeypatch):
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod, optim as optim_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    monkeypatch.setattr(optim_mod, "K", fake_kernels)
    return fake_kernels


@pytest.mark.parametrize("max_norm", [None, 0.05])
def test_fused_adamw_host_logic_matches_torch(fake_k, max_norm):
    from unidisc_amd import FusedAdamW

    g = Golden("c_large")
    diff = build_product(g, device="cpu")
    diff.rng_device = "cpu"
    bb = diff.backbone
    ref_model = copy.deepcopy(bb)
    opt = FusedAdamW(bb, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, max_grad_norm=max_norm)
    ref_opt = torch.optim.AdamW(ref_model.parameters(), lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    for step in range(3):
        torch.manual_seed(100 + step)
        out = diff.training_step(g.batch(), step)
        out.loss.backward()
        for (n, p), (_, q) in zip(bb.named_parameters(), ref_model.named_parameters()):
            q.grad = p.grad.detach().clone() if p.grad is not None else None
        if max_norm is not None:
            tn = torch.nn.utils.clip_grad_norm_(ref_model.parameters(), max_norm)
        ref_opt.step()
        opt.step()
        if max_norm is not None:
            assert torch.allclose(opt.grad_norm.reshape(()), tn, rtol=1e-5)
        opt.zero_grad()
        ref_opt.zero_grad()
        for (n, p), (_, q) in zip(bb.named_parameters(), ref_model.named_parameters()):
            assert torch.allclose(p, q, rtol=2e-5, atol=1e-7), (step, n)
        # the optimizer keeps the bf16 shadows current and switches the per-forward re-cast off
        assert bb.recast_every_forward is False
        for lin in bb._lins.values():
            assert torch.equal(lin.w16[: lin.out], lin.weight.detach().bfloat16())
            assert torch.equal(lin.w16t[:, : lin.out], lin.weight.detach().t().bfloat16())
    assert opt.step_count == 3


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 4096, 100003])
@pytest.mark.parametrize("clip", [False, True])
def test_adamw_kernel_matches_torch(n, clip):
    from unidisc_amd import kernels as K

    gen = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=gen)
    grads = [torch.randn(n, generator=gen) * (10.0 if clip else 1.0) for _ in range(4)]
    lr, betas, eps, wd, mx = 3e-3, (0.9, 0.99), 1e-8, 0.05, (1.0 if clip else None)
    (ref,), norms = _torch_reference_steps([p0], [[g] for g in grads], lr, betas, eps, wd, mx)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    gsq = torch.zeros(1, device=DEV)
    for t, g in enumerate(grads, 1):
        gd = g.to(DEV)
        if clip:
            K.sumsq(gd, gsq)
            assert torch.allclose(gsq.sqrt().cpu().reshape(()), norms[t - 1], rtol=1e-5)
        K.adamw_step(p, gd, m, v, lr, betas[0], betas[1], eps, wd, t, gsq if clip else None, mx)
    assert torch.allclose(p.cpu(), ref, rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("R,C,pad", [(64, 64, 0), (200, 328, 0), (2048, 512, 0), (97, 130, 31), (48, 2048, 80)])
def test_adamw_shadow_kernel(R, C, pad):
    from unidisc_amd import kernels as K

    gen = torch.Generator().manual_seed(R * 1000 + C)
    p0 = torch.randn(R, C, generator=gen)
    grads = [torch.randn(R, C, generator=gen) for _ in range(3)]
    lr, betas, eps, wd = 1e-2, (0.9, 0.999), 1e-8, 0.0
    (ref,), _ = _torch_reference_steps([p0], [[g] for g in grads], lr, betas, eps, wd, None)
    Rp = R + pad
    p, m, v = p0.clone().to(DEV), torch.zeros(R, C, device=DEV), torch.zeros(R, C, device=DEV)
    w16, w16t = torch.zeros((Rp, C), dtype=torch.bfloat16, device=DEV), torch.zeros((C, Rp), dtype=torch.bfloat16, device=DEV)
    for t, g in enumerate(grads, 1):
        K.adamw_step_shadow(p, g.to(DEV), m, v, lr, betas[0], betas[1], eps, wd, t, None, None, w16, w16t)
    assert torch.allclose(p.cpu(), ref, rtol=2e-5, atol=1e-7)
    assert torch.equal(w16[:R].cpu(), p.cpu().bfloat16()) and torch.equal(w16t[:, :R].cpu(), p.cpu().t().bfloat16())
    assert not w16[R:].any() and not w16t[:, R:].any()  # padding rows / columns stay zero


@pytest.mark.gpu
def test_training_steps_with_fused_adamw_match_torch_adamw_with_recast():
    """Three optimisation steps of the product model on the GPU: FusedAdamW (shadows maintained, no re-cast) vs torch.optim.AdamW +
    clip_grad_norm_ with the forward re-casting the weights -- same losses, same parameters.  (eps = 1e-3: with the default 1e-8 Adam
    normalises gradients that are pure accumulation-order noise, e.g. the k-norm bias whose true gradient is zero, to +-lr steps, and two
    runs of the SAME code then differ in those parameters.)"""
    from unidisc_amd import FusedAdamW

    g = Golden("c_large")
    res = []
    for fused in (True, False):
        diff = build_product(g, device=DEV)
        bb = diff.backbone
        batch = {k: v.to(DEV) for k, v in g.batch().items()}
        if fused:
            opt = FusedAdamW(bb, lr=2e-3, eps=1e-3, weight_decay=0.01, max_grad_norm=0.5)
        else:
            opt = torch.optim.AdamW(bb.parameters(), lr=2e-3, eps=1e-3, weight_decay=0.01)
        losses = []
        for step in range(3):
            torch.manual_seed(7 + step)
            out = diff.training_step(batch, step)
            out.loss.backward()
            if not fused:
                torch.nn.utils.clip_grad_norm_(bb.parameters(), 0.5)
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(float(out.loss))
        res.append((losses, {n: p.detach().float().cpu().clone() for n, p in bb.named_parameters()}))
    (l1, p1), (l0, p0) = res
    assert all(abs(a - b) <= 2e-3 * abs(b) for a, b in zip(l1, l0)), (l1, l0)
    for n in p0:
        assert rel_err(p1[n], p0[n]) < 2e-3, n


# ------------------------------------------------------------------------------------------------ parameter EMA (models/ema.py:44-53)
class _RefEMA:
    """The reference's ExponentialMovingAverage.update / copy_to / store / restore arithmetic, restated for the check."""

    def __init__(self, params, decay, use_num_updates=True):
        self.decay, self.n = decay, 0 if use_num_updates else None
        self.shadow = [p.clone().detach() for p in params]

    def update(self, params):
        d = self.decay
        if self.n is not None:
            self.n += 1
            d = min(d, (1 + self.n) / (10 + self.n))
        for s, p in zip(self.shadow, params):
            s.sub_((1.0 - d) * (s - p))


def test_fused_adamw_ema_host_logic(fake_k):
    from unidisc_amd import FusedAdamW

    def net():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(12, 20), torch.nn.LayerNorm(20), torch.nn.Linear(20, 7))

    def grads(mod, seed):
        g = torch.Generator().manual_seed(seed)
        for p in mod.parameters():
            p.grad = torch.randn(p.shape, generator=g)

    model, ref = net(), net()
    opt = FusedAdamW(model, lr=1e-2, weight_decay=0.01, max_grad_norm=0.05, maintain_shadows=False, ema_decay=0.999)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ema = _RefEMA(list(ref.parameters()), 0.999)
    for it in range(12):
        grads(model, 50 + it)
        grads(ref, 50 + it)
        opt.step()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
        topt.step()
        ema.update(list(ref.parameters()))
    for p, s_ in zip(opt.params, ema.shadow):
        torch.testing.assert_close(opt.ema[id(p)], s_, rtol=2e-5, atol=1e-7)
    assert opt.ema_num_updates == 12
    # store / copy_to / restore
    before = [p.detach().clone() for p in opt.params]
    opt.ema_store_and_copy()
    for p in opt.params:
        assert torch.equal(p, opt.ema[id(p)])
    opt.ema_restore()
    for p, b in zip(opt.params, before):
        assert torch.equal(p, b)
    # state dict round trip keeps the EMA and its warm-up counter
    opt2 = FusedAdamW(model, lr=1e-2, maintain_shadows=False, ema_decay=0.999)
    opt2.load_state_dict(opt.state_dict())
    assert opt2.ema_num_updates == 12 and all(torch.equal(opt2.ema[id(p)], opt.ema[id(p)]) for p in opt.params)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 4096, 100003])
def test_adamw_kernel_ema(n):
    from unidisc_amd import kernels as K
    g = torch.Generator().manual_seed(n)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v, e = torch.zeros(n), torch.zeros(n), torch.randn(n, generator=g)
    pd, gd, md, vd, ed = (t.clone().cuda() for t in (p, gr, m, v, e))
    for step in (1, 2, 3):
        K.adamw_step(pd, gd, md, vd, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, ema=ed, ema_decay=0.9 + 0.03 * step)
        fake_kernels.adamw_step(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, ema=e, ema_decay=0.9 + 0.03 * step)
    torch.testing.assert_close(pd.cpu(), p, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(ed.cpu(), e, rtol=2e-6, atol=1e-7)   # fp32 arithmetic in the same order: rounding-level agreement


@pytest.mark.gpu
@pytest.mark.parametrize("R,C", [(64, 64), (200, 328), (97, 130)])
def test_adamw_shadow_kernel_ema(R, C):
    from unidisc_amd import kernels as K
    g = torch.Generator().manual_seed(R * 1000 + C)
    p, gr, e = torch.randn(R, C, generator=g), torch.randn(R, C, generator=g), torch.randn(R, C, generator=g)
    m, v = torch.zeros(R, C), torch.zeros(R, C)
    w16, w16t = torch.zeros(R, C, dtype=torch.bfloat16), torch.zeros(C, R, dtype=torch.bfloat16)
    pd, gd, md, vd, ed, w16d, w16td = (t.clone().cuda() for t in (p, gr, m, v, e, w16, w16t))
    K.adamw_step_shadow(pd, gd, md, vd, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None, None, w16d, w16td, ema=ed, ema_decay=0.95)
    fake_kernels.adamw_step_shadow(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None, None, w16, w16t, ema=e, ema_decay=0.95)
    torch.testing.assert_close(ed.cpu(), e, rtol=2e-6, atol=1e-7)
    assert torch.equal(w16d.cpu(), pd.cpu().bfloat16()) and torch.equal(w16td.cpu(), pd.cpu().t().bfloat16())
