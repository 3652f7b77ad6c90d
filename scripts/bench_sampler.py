"""Sampler step cost on the 1.4 B workload (B=8, L=1280, V=48385): fused update (logits of [MASK] rows -> udm_ddpm_sample_rows) vs the
reference-shaped update (full [B,L,V] SUBS log-probs -> exp -> q -> rand_like -> argmax, model_eval.py:2073-2106) on the same backbone."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg, diff = bench.build("unidisc-1.4b-l1280", dev, 0.0)
diff.backbone.eval()
B, L = 8, 1280
mask = diff.mask_index
modality = torch.zeros(B, L, dtype=torch.int64, device=dev)
modality[:, 256:] = 1   # 256 text + 1024 image positions

def timed(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

res = {}
for frac in (1.0, 0.5, 0.1):
    x = torch.randint(0, diff.text_vocab_size - 1, (B, L), device=dev)
    x[:, 256:] += diff.text_vocab_size  # image ids live above the text vocabulary
    m = torch.rand(B, L, device=dev) < frac
    x = torch.where(m, torch.full_like(x, mask), x)
    t = torch.full((B, 1), 0.5, device=dev)
    dt = 1.0 / 64
    def fused():
        with torch.no_grad():
            diff._ddpm_caching_update(x, t, dt, modality=modality, seed=1)
    def reference_shaped():
        with torch.no_grad():
            sigma_t, _ = diff.noise(t.squeeze(-1))
            p = diff.forward(x, sigma_t, modality=modality).float().exp()
            q = p * dt
            q[:, :, mask] = float(t[0, 0] - dt)
            g = 1e-10 - (torch.rand_like(q) + 1e-10).log()
            _x = (q / g).argmax(-1)
            keep = (x != mask).to(x.dtype)
            return keep * x + (1 - keep) * _x
    res[f"masked_{frac}"] = dict(fused_ms=round(timed(fused), 2), reference_shaped_ms=round(timed(reference_shaped), 2))
# classifier-free guidance (text kept as conditioning, image positions generated): one [x ; x_uncond] pass + in-kernel mix, against the
# reference-shaped form (full [2B, L, V] logits -> fp32 mix -> SUBS -> exp -> q -> rand_like -> argmax, model_eval.py:1787-1817, 2090-2096)
from unidisc_amd.config import Cfg
diff.config.eval = Cfg(cfg=2.0)
x0 = torch.randint(0, diff.text_vocab_size - 1, (B, L), device=dev)
x0[:, 256:] += diff.text_vocab_size
x0_unmask = modality == 0
for frac in (1.0, 0.5):
    m = (torch.rand(B, L, device=dev) < frac) & ~x0_unmask
    x = torch.where(m, torch.full_like(x0, mask), x0)
    t = torch.full((B, 1), 0.5, device=dev)
    dt = 1.0 / 64
    def fused_cfg():
        with torch.no_grad():
            diff._ddpm_caching_update(x, t, dt, x0=x0, x0_unmask=x0_unmask, modality=modality, seed=1)
    def reference_shaped_cfg():
        with torch.no_grad():
            sigma_t, _ = diff.noise(t.squeeze(-1))
            xu = x.clone(); xu[x0_unmask] = mask
            lg = diff.forward(torch.cat([x, xu]), torch.cat([sigma_t, sigma_t]), modality=torch.cat([modality, modality]), return_logits=True)
            lc, lu = lg.chunk(2, 0)
            w = diff.get_cfg_weight(t.squeeze(-1)).unsqueeze(-1)
            z = (1 + w) * lc - w * lu
            z[..., mask] = -1e6
            p = torch.log_softmax(z, -1).exp()
            q = p * dt
            q[:, :, mask] = float(t[0, 0] - dt)
            g = 1e-10 - (torch.rand_like(q) + 1e-10).log()
            _x = (q / g).argmax(-1)
            keep = (x != mask).to(x.dtype)
            return keep * x + (1 - keep) * _x
    res[f"cfg_masked_{frac}_of_image"] = dict(fused_ms=round(timed(fused_cfg), 2), reference_shaped_ms=round(timed(reference_shaped_cfg), 2))
diff.config.eval = Cfg(cfg=None)
steps = 16
t0 = time.perf_counter(); out, nfe = diff.sample(num_steps=steps, batch_size=B, modality=modality, seed=3, return_nfe=True); torch.cuda.synchronize()
t0 = time.perf_counter(); out, nfe = diff.sample(num_steps=steps, batch_size=B, modality=modality, seed=4, return_nfe=True); torch.cuda.synchronize()
dt_all = time.perf_counter() - t0
res["sample_16_steps"] = dict(ms_total=round(dt_all * 1e3, 1), nfe=nfe, ms_per_forward=round(dt_all * 1e3 / nfe, 2), tokens_out=B * L,
                              masks_left=int((out == mask).sum()))
print(json.dumps(res))
