#!/usr/bin/env python3
"""Headline benchmark: denoising tokens/s (fwd+bwd) of the UniDisc 1.4 B DiT at seq_len 1280 on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch: ``Diffusion.training_step`` (update_batch → _sample_t →
q_xt → DiT forward → fused SUBS cross-entropy → weighted loss) followed by ``loss.backward()`` — including, when N > 1,
the bf16 gradient all-reduce overlapped with the backward.  The optimizer is excluded (BASELINE.json metric).  The fp32 →
bf16 weight cast that autocast performs every forward in the reference is INSIDE the timed region.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (AMD's 5 PF figure is 2:1 sparse)
PEAK_HBM_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable with a plain copy)

WORKLOADS = {
    # BASELINE.json configs[2]: large_scale_train + large_scale_train_high_res, data.block_size=256
    "unidisc-1.4b-l1280": dict(preset="extra_large", txt_length=256, img_length=1024, text_vocab=32001, image_vocab=16384, batch=8,
                               desc="UniDisc 1.4B non-interleaved, seq_len=1280 (256 text + 1024 image), bf16"),
    # SURVEY §8(d) secondary workload: the same model with adaLN-Zero timestep modulation on (`time_conditioning=True`: models/dit.py:922-925, 966-967; no shipped
    # config sets it, north_star names it).  F_tok as above: the modulation GEMMs are [B, cond_dim] x [cond_dim, 6 d] per block - not counted, like the reference's 6 N D
    "unidisc-1.4b-l1280-adaln": dict(preset="extra_large", txt_length=256, img_length=1024, text_vocab=32001, image_vocab=16384, batch=8, time_conditioning=True,
                                     desc="UniDisc 1.4B non-interleaved, seq_len=1280 (256 text + 1024 image), bf16, adaLN-Zero time conditioning ON"),
    # BASELINE.json configs[1]: UniDisc-S
    # BASELINE.json configs[4] with bf16 attention (the fp8 forward of rounds 2-4 never paid inside the step and was removed): large_scale_train_high_res_interleaved with model.length=4608:
    # every row packs 4 samples of 128 text + 1024 image tokens; attention stays inside a sample (document mask from sample_ids)
    "unidisc-1.4b-interleaved-l4608": dict(preset="extra_large", txt_length=512, img_length=4096, text_vocab=32001, image_vocab=16384, batch=2,
                                           packed=dict(samples=4, txt=128, img=1024),
                                           desc="UniDisc 1.4B interleaved, seq_len=4608 (4 packed samples of 128 text + 1024 image), bf16 attention"),
    "unidisc-s-l384": dict(preset="small", txt_length=128, img_length=256, text_vocab=32001, image_vocab=8192, batch=64,
                           desc="UniDisc-S (~115M) DiT, joint 128 text + 256 image VQ tokens, bf16"),
}


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel (launch-weighted mean over its template instances) from the committed PMC pass
    (scripts/gpu_pmc_bench.sh: separate FETCH_SIZE / WRITE_SIZE runs of this same command, unit and gfx950 corrections of the guide).
    PMC counters cannot be collected from inside the timed run, so the figure is read from profiles/; None if the file is absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic_per_kernel*.json")))
    if not files:
        return None, None
    try:
        data = json.load(open(files[-1]))
        rows = [v for k, v in data.items() if k.startswith(kernel_prefix)]
        n = sum(v["launches"] for v in rows)
        return (sum(v["launches"] * v["hbm_bytes_per_launch"] for v in rows) / n if n else None), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def sustained_mfma_reference():
    """Context for `roofline` (NOT its peak): what a bare v_mfma_f32_32x32x16_bf16 stream sustains on random bf16 operands on this part, read from the committed
    run of experiments/ubench/mfma_power.hip (the mean of its "random bf16" 32x32x16 rows) - a number measured on ANOTHER box of the pool, so the line names
    the file it came from.  (None, None) when no such log is committed."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*mfma_power.log")))
    if not files:
        return None, None
    vals = [float(m.group(1)) for ln in open(files[-1]) if ln.startswith("random bf16,") and "16x16x32" not in ln for m in [re.search(r"-> (\d+) TF", ln)] if m]
    return (sum(vals) / len(vals), os.path.relpath(files[-1], ROOT)) if vals else (None, None)


def flops_per_token(n_blocks, d, V, L_att):
    """F_tok = 6·P_mm + 12·n·L_att·d with P_mm = n·12d² + d·V (matmul weights only; no recompute credit)."""
    p_mm = n_blocks * 12 * d * d + d * V
    return 6 * p_mm + 12 * n_blocks * L_att * d


def build(workload, device, dropout):
    from unidisc_amd import MODEL_PRESETS, Diffusion, make_config

    w = WORKLOADS[workload]
    large = w["preset"] == "extra_large"
    cfg = make_config(**MODEL_PRESETS[w["preset"]], txt_length=w["txt_length"], img_length=w["img_length"], norm_type="rms", qk_norm=True,
                      sandwich_normalization=True, modality_embed=True, rope_2d=large, linear_factor=2.0 if large else 1.0,
                      time_conditioning=bool(w.get("time_conditioning", False)),
                      multimodal_batches=True, force_argmax_valid_indices=True, dropout=dropout, zero_linear_init=False,
                      image_vocab_size=w["image_vocab"], mask_entire_modality=0.1, softmin_snr=5, text_loss_weight=1.0,
                      img_loss_weight=0.5 if large else None, force_full_attention_mask=True if large else None,
                      force_full_attention_mask_loss_only=None if large else True)
    cfg.model.force_text_vocab_size = w["text_vocab"] - 1
    if "packed" in w:  # large_scale_train_high_res_interleaved.yaml
        cfg.trainer.interleaved = True
        cfg.trainer.interleaved_training_flex_attention = True
        cfg.data.require_sample_ids = True
        cfg.model.use_flex_attention = True
        cfg.trainer.img_loss_weight, cfg.trainer.mask_entire_modality = 0.2, 0.2
    diff = Diffusion(cfg, None, device)
    diff.backbone.train()
    return cfg, diff


def synthetic_batch(workload, B, seed):
    w = WORKLOADS[workload]
    g = torch.Generator().manual_seed(seed)
    if "packed" in w:
        pk = w["packed"]
        ids, mod, sid = [], [], []
        for s_ in range(pk["samples"]):
            ids += [torch.randint(0, w["text_vocab"] - 1, (B, pk["txt"]), generator=g), torch.randint(w["text_vocab"], w["text_vocab"] + w["image_vocab"], (B, pk["img"]), generator=g)]
            mod += [torch.zeros(B, pk["txt"], dtype=torch.int64), torch.ones(B, pk["img"], dtype=torch.int64)]
            sid += [torch.full((B, pk["txt"] + pk["img"]), s_, dtype=torch.int64)]
        ids = torch.cat(ids, 1)
        return dict(input_ids=ids, modality=torch.cat(mod, 1), sample_ids=torch.cat(sid, 1), attention_mask=torch.ones_like(ids, dtype=torch.bool))
    return dict(txt_input_ids=torch.randint(0, w["text_vocab"] - 1, (B, w["txt_length"]), generator=g, dtype=torch.int32),
                img_input_ids=torch.randint(0, w["image_vocab"], (B, w["img_length"]), generator=g, dtype=torch.int32).to(torch.int16),
                txt_attention_mask=torch.ones(B, w["txt_length"], dtype=torch.bool))


# fraction of the L x L query/key pairs a query may attend to (document masks of packed rows: sum_doc len^2 / L^2); set per workload in main()
ATTN_PAIR_FRACTION = 1.0

GEMM_ENTRY_POINTS = ("udm_gemm_nt_bf16", "udm_gemm_tn_bf16", "udm_gemm_nn_bf16", "udm_gemm_nt_splitk_bf16", "udm_gemm_tn_splitk_bf16", "udm_gemm_tn_pair_bf16",
                     "udm_gemm_tn_multi_bf16")   # (A, B, C, M, N, K, ...)


def _work(name, a):
    """Algorithmic work of one launch from the C-ABI arguments: ("flop" | "byte", amount) or None.  Bytes follow DESIGN.md §4 (per-element
    figures of the HBM-bound kernels: what the op must read and write once), flops count what the MFMA pipe is asked to do."""
    if name == "udm_gemm_tn_pair_bf16":   # (A0, B0, C0, M0, lda0, ldb0, ldc0, A1, B1, C1, M1, lda1, ldb1, ldc1, N, K, beta, ws, ws_elems)
        return "flop", 2.0 * (a[3] + a[10]) * a[14] * a[15]
    if name == "udm_gemm_tn_multi_bf16":  # (nprob, A[], B[], C[], M[], N[], lda[], ldb[], K, ...)
        return "flop", 2.0 * a[8] * sum(float(a[4][i]) * float(a[5][i]) for i in range(a[0]))
    if name in GEMM_ENTRY_POINTS:
        return "flop", 2.0 * a[3] * a[4] * a[5]
    if name == "udm_attention_fwd":           # only the pairs inside a document count (a packed row of 4 x 1152 is a quarter of 4608^2)
        return "flop", 4.0 * a[7] * a[8] * a[9] * a[9] * a[10] * ATTN_PAIR_FRACTION
    if name == "udm_attention_bwd":          # dQ pass 6 (S, dP, dQ) + dK/dV pass 8 (S, dP, dV, dK): executed, recompute included
        return "flop", 14.0 * a[12] * a[13] * a[14] * a[14] * a[15] * ATTN_PAIR_FRACTION
    if name == "udm_norm_fwd":
        return "byte", 6.0 * a[10] * a[11]
    if name == "udm_norm_bwd":
        return "byte", (14.0 if a[18] else 10.0) * a[14] * a[15]
    if name == "udm_norm_residual_bwd":   # dy 2 + x 4 + dx 4 (when accumulating) + branch 2 read, dx 4 + d branch 2 written
        return "byte", (18.0 if a[7] else 14.0) * a[15] * a[16]
    if name == "udm_residual_fwd":
        return "byte", 10.0 * a[9] * a[10]
    if name == "udm_residual_norm_fwd":
        return "byte", 12.0 * a[9] * a[10]
    if name == "udm_residual_bwd":
        return "byte", 8.0 * a[11] * a[12]
    if name == "udm_qknorm_rope_fwd":
        return "byte", 8.0 * a[10] * a[11]
    if name == "udm_qknorm_rope_bwd":
        return "byte", 12.0 * a[13] * a[14]
    if name == "udm_cast_transpose_f32_bf16":
        return "byte", 8.0 * a[3] * a[4]
    if name == "udm_cast_transpose_multi_f32_bf16":   # 64 x 64 tiles of fp32 read once, bf16 written twice (edge tiles counted whole)
        return "byte", 8.0 * 4096 * a[2]
    if name == "udm_transpose_bf16":
        return "byte", (4.0 if a[1] else 2.0) * a[2] * a[3]
    if name == "udm_embedding_fwd":
        return "byte", 8.0 * a[5] * a[6]
    if name == "udm_embedding_bwd":
        return "byte", 8.0 * a[5] * a[6]
    if name in ("udm_subs_ce_fwd", "udm_subs_ce_bwd"):   # placeholder (whole rows): the table pass replaces it by the bytes of the valid id range of the step's [MASK] rows
        return "byte", (2.0 if name.endswith("fwd") else 4.0) * a[7] * a[8]
    if name in ("udm_cast_f32_bf16", "udm_cast_bf16_f32"):
        return "byte", 6.0 * a[2]
    return None


class KernelTimer:
    """HIP-event timing of C-ABI launches on the stream they are launched on (torch's current stream).  `names` = None times every entry
    point (the post-pass that builds `roofline_table`); the timed region only brackets the GEMM family (`roofline`)."""

    def __init__(self, names=GEMM_ENTRY_POINTS, sample=1):
        self.names = names
        self.records = []
        self.enabled = False
        # An event pair is not free on this stack: each record is a barrier + signal packet, ~5 us of bubble on the stream (measured: bracketing
        # all ~170 GEMM launches of a step costs 1.6-1.75 ms of an 88 ms step).  Inside the timed region only 1 launch in `sample` is bracketed, picked
        # by a seeded generator so that over the K steps every launch site is hit; the average duration is over the sampled launches.
        self.sample = max(1, int(sample))
        self.seen = 0
        self._rng = __import__("random").Random(1234)

    def install(self):
        from unidisc_amd import _lib

        orig = _lib.call
        timer = self

        def call(name, *args):
            hit = timer.enabled and (timer.names is None or name in timer.names)
            if hit:
                timer.seen += 1
                hit = timer.sample == 1 or timer._rng.randrange(timer.sample) == 0
            if hit:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                orig(name, *args)
                e.record()
                timer.records.append((s, e, name, _work(name, args)))
            else:
                orig(name, *args)

        self._orig = orig
        _lib.call = call

    def uninstall(self):
        from unidisc_amd import _lib

        _lib.call = self._orig

    def summary(self):
        if not self.records:
            return None
        ms = sum(s.elapsed_time(e) for s, e, _, _ in self.records)
        fl = sum(w[1] for _, _, _, w in self.records if w and w[0] == "flop")
        return dict(launches=len(self.records), launches_seen=self.seen, total_ms=ms, flops=fl)

    def table(self, steps):
        """Per entry point: launches and ms per step, achieved TFLOP/s against the dense bf16 MFMA peak or TB/s against the 8 TB/s HBM peak."""
        agg = {}
        for s, e, name, w in self.records:
            r = agg.setdefault(name, dict(launches=0, ms=0.0, flop=0.0, byte=0.0))
            r["launches"] += 1
            r["ms"] += s.elapsed_time(e)
            if w:
                r[w[0]] += w[1]
        rows = []
        for name, r in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            row = dict(entry_point=name, launches_per_step=r["launches"] / steps, ms_per_step=r["ms"] / steps)
            if r["flop"]:
                ach = r["flop"] / (r["ms"] * 1e-3) / 1e12
                row.update(bound="mfma", achieved=ach, peak=PEAK_BF16_DENSE_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_DENSE_TFLOPS)
            elif r["byte"]:
                ach = r["byte"] / (r["ms"] * 1e-3) / 1e12
                row.update(bound="hbm", achieved=ach, peak=PEAK_HBM_TBS, unit="TB/s", frac=ach / PEAK_HBM_TBS)
            rows.append(row)
        return rows


def cpu_baseline(workload, cfg, diff, seed):
    """The oracle (CPU restatement of the reference path) timed on this host's cores on a BOUNDED sample of the same workload: ONE sequence (B=1, full
    length, width, depth and vocabulary head) through fwd+bwd, measured.  A 1-block probe runs first; only if it predicts more than ~150 s for the full
    depth (a very slow host) is the full-depth time extrapolated from 1 and 2 blocks instead, and the sample text says so."""
    from oracle import unidisc_oracle as O
    from oracle.cases import lumina_rope_2d

    w = WORKLOADS[workload]
    m = cfg.model
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    L = w["txt_length"] + w["img_length"]
    nseq = 2 if "packed" in w else 1     # (one packed row alone trips the reference's own `.squeeze(-1)` on the interleaved ignore mask, which the oracle restates)

    def run(nb):
        case = dict(hidden_size=m.hidden_size, n_heads=m.n_heads, cond_dim=m.cond_dim, n_blocks=nb, txt_length=w["txt_length"],
                    img_length=w["img_length"], vocab_size=diff.vocab_size, text_vocab_size=diff.text_vocab_size, norm_type="rms", qk_norm=True,
                    sandwich_normalization=True, modality_embed=True, rope_2d=m.rope_2d, linear_factor=m.linear_factor,
                    time_conditioning=bool(w.get("time_conditioning", False)), interleaved="packed" in w,
                    multimodal_batches=True, force_argmax_valid_indices=True, mask_entire_modality=cfg.trainer.get("mask_entire_modality", 0.1), softmin_snr=5,
                    text_loss_weight=1.0, img_loss_weight=cfg.trainer.get("img_loss_weight"), force_full_attention_mask=cfg.trainer.get("force_full_attention_mask"),
                    force_full_attention_mask_loss_only=cfg.trainer.get("force_full_attention_mask_loss_only"))
        ocfg = O.OracleConfig.from_case(case)
        keep = lambda k: not k.startswith("blocks.") or int(k.split(".")[1]) < nb
        P = {k: v.detach().float().cpu().requires_grad_() for k, v in diff.backbone.named_parameters() if keep(k)}
        bufs = O.make_buffers(ocfg, lumina_rope_2d)
        batch = O.update_batch(ocfg, synthetic_batch(workload, nseq, seed))
        t0 = time.perf_counter()
        out = O.compute_loss(ocfg, P, bufs, batch, torch.Generator().manual_seed(seed), bf16=True)   # the reference's bf16-autocast numerics
        out.loss.backward()
        return time.perf_counter() - t0

    t1 = run(1)
    what = f"oracle (torch CPU, the reference's bf16-autocast rounding points emulated) fwd+bwd of {nseq} sequence(s) (B={nseq}, L={L}, d={m.hidden_size}, V={diff.vocab_size})"
    if t1 * (1 + 0.6 * (m.n_blocks - 1)) <= 150:
        total = run(m.n_blocks)
        sample = f"{what} through all {m.n_blocks} blocks: {total:.1f} s MEASURED (1-block probe {t1:.1f} s; {threads} threads of {cores} available cores)"
    else:
        t2 = run(2)
        total = t1 + (m.n_blocks - 1) * max(t2 - t1, 1e-6)
        sample = (f"{what} through 1 and 2 of the {m.n_blocks} blocks: {t1:.1f} s and {t2:.1f} s; EXTRAPOLATED to {m.n_blocks} blocks = {total:.1f} s "
                  f"(host too slow for the full depth inside the bench's time bound; {threads} threads of {cores} available cores)")
    return dict(value=nseq * L / total, unit="tokens/s", cores=threads, kind="port", sample=sample)


def cpu_baseline_legs(seed):
    """SURVEY §8(d) flavour of the CPU comparator, beside the workload's own leg: the oracle with the reference's bf16-autocast rounding points
    emulated (`bf16=True`), on min(cores, 32) threads, for BASELINE configs[0] (config A: n=2, d=256, H=4, L=128, V=1001, B=32; fwd+loss and fwd+bwd)
    and configs[1] (config B: UniDisc-S, n=12, d=768, L=128+256, V=40193, B=8; 1 step fwd+bwd).  Bounded: a few seconds each."""
    from oracle import unidisc_oracle as O
    from oracle.cases import lumina_rope_2d
    from unidisc_amd import Diffusion, make_config

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # torch's CPU thread pool is counter-productive far below the core count of a 256-core host at these sizes (measured on the GPU box: config A
    # forward 41 k tokens/s on 8 threads here vs 0.18 k tokens/s on 256 threads there), so the legs use min(cores, 32) threads and say so
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    legs = {}
    shapes = {
        "config_a": (dict(hidden_size=256, n_heads=4, cond_dim=128, n_blocks=2, txt_length=128, img_length=0, text_vocab_size=1001, vocab_size=1001,
                          norm_type="layernorm", qk_norm=False, sandwich_normalization=False, modality_embed=False, rope_2d=False, time_conditioning=True,
                          multimodal_batches=False, force_argmax_valid_indices=False), 32, 3),
        "config_b": (dict(hidden_size=768, n_heads=12, cond_dim=128, n_blocks=12, txt_length=128, img_length=256, text_vocab_size=32001, vocab_size=40193,
                          norm_type="rms", qk_norm=True, sandwich_normalization=True, modality_embed=True, rope_2d=False, time_conditioning=False,
                          multimodal_batches=True, force_argmax_valid_indices=True, mask_entire_modality=0.1, softmin_snr=5, text_loss_weight=1.0,
                          force_full_attention_mask_loss_only=True), 8, 1),
    }
    for name, (case, B, steps) in shapes.items():
        kw = {k: case[k] for k in ("hidden_size", "n_heads", "cond_dim", "n_blocks", "txt_length", "img_length", "norm_type", "qk_norm", "sandwich_normalization",
                                    "modality_embed", "rope_2d", "time_conditioning", "multimodal_batches", "force_argmax_valid_indices")}
        pc = make_config(**kw, image_vocab_size=(case["vocab_size"] - case["text_vocab_size"]) or None)
        pc.model.force_text_vocab_size = case["text_vocab_size"] - 1
        torch.manual_seed(seed)
        P = {k: v.detach().float().requires_grad_() for k, v in Diffusion(pc, None, "cpu").backbone.named_parameters()}   # parameter containers only (no compute)
        ocfg = O.OracleConfig.from_case(case)
        bufs = O.make_buffers(ocfg, lumina_rope_2d)
        g = torch.Generator().manual_seed(seed)
        Lt, Li, Vt = case["txt_length"], case["img_length"], case["text_vocab_size"]
        if Li:
            raw = dict(txt_input_ids=torch.randint(0, Vt - 1, (B, Lt), generator=g, dtype=torch.int32),
                       img_input_ids=torch.randint(0, case["vocab_size"] - Vt, (B, Li), generator=g, dtype=torch.int32).to(torch.int16),
                       txt_attention_mask=torch.ones(B, Lt, dtype=torch.bool))
        else:
            raw = dict(input_ids=torch.randint(0, Vt - 1, (B, Lt), generator=g), attention_mask=torch.ones(B, Lt, dtype=torch.bool))
        batch = O.update_batch(ocfg, raw)
        L = Lt + Li
        with torch.no_grad():   # fwd + loss
            O.compute_loss(ocfg, P, bufs, batch, torch.Generator().manual_seed(seed), bf16=True)
            t0 = time.perf_counter()
            for _ in range(steps):
                O.compute_loss(ocfg, P, bufs, batch, torch.Generator().manual_seed(seed), bf16=True)
            t_fwd = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            for p_ in P.values():
                p_.grad = None
            O.compute_loss(ocfg, P, bufs, batch, torch.Generator().manual_seed(seed), bf16=True).loss.backward()
        t_fb = (time.perf_counter() - t0) / steps
        legs[name] = dict(batch=B, seq_len=L, steps=steps, fwd_loss_tokens_per_s=B * L / t_fwd, fwd_bwd_tokens_per_s=B * L / t_fb, cores=threads, cores_available=cores,
                          numerics="oracle with bf16-autocast rounding points emulated")
    return legs


def launcher_command(n_gpus, argv, port=None):
    """The command that runs this script on `n_gpus` ranks of one node (one process per GPU over RCCL): what `python bench.py --gpus N` starts
    by itself when it was not launched by torch.distributed.run (main.py:641-656 is the reference's accelerate launch of the same shape)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """Start the N ranks as CHILD processes (this process has not touched the GPU and never will: no exec of a GPU-initialised process),
    relay their output, exit with the launcher's code."""
    import subprocess

    from unidisc_amd.ddp import rccl_channel_env

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    rccl_channel_env(env)     # bound RCCL's CU footprint (NCCL_MAX_NCHANNELS, default 32) unless the caller chose otherwise
    proc = subprocess.run(launcher_command(n_gpus, argv), env=env)
    raise SystemExit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="unidisc-1.4b-l1280", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: workload's)")
    ap.add_argument("--dropout", type=float, default=0.1, help="model.dropout (reference extra_large.yaml: 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-any-world", action="store_true", help="also time the CPU comparator on rank 0 of an N > 1 run (after every rank left the process group)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-every", type=int, default=16, help="bracket 1 GEMM launch in this many with HIP events inside the timed region (1 = all: costs ~1.7 ms/step)")
    ap.add_argument("--hog-cus", type=int, default=0, help="diagnostics: hold this many CUs with a spinning kernel for the whole run (stand-in for RCCL's channel kernels; "
                    "combine with UDM_GEMM_CUS = 256 - n so the GEMMs plan for the remaining CUs)")
    ap.add_argument("--ddp-mode", default=None, choices=["auto", "overlap", "overlap_planned", "serialized"],
                    help="gradient all-reduce schedule for --gpus > 1 (default: UDM_DDP_MODE or auto = timed in warm-up on all ranks, fastest kept)")
    ap.add_argument("--table-steps", type=int, default=2, help="extra untimed steps after the timed region with every launch event-timed (roofline_table); 0 = off")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            self_launch(args.gpus, sys.argv[1:])    # (before anything here initialises the GPU)
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    local_rank %= torch.cuda.device_count()  # one rank per GPU on a node; wraps only in the single-GPU rehearsal of the N > 1 path (below)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if args.hog_cus > 0 and world > 1:
        raise SystemExit("--hog-cus is a single-GPU diagnostic (it cannot be fenced by a collective barrier)")
    rccl_channels = 0
    if world > 1:
        from unidisc_amd.ddp import rccl_channel_env

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rccl_channels = rccl_channel_env()     # before the communicator exists (the driver launches torch.distributed.run itself: self_launch is not on that path)
        backend = os.environ.get("UDM_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only to rehearse the N > 1 path with two ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    L = w["txt_length"] + w["img_length"]
    global ATTN_PAIR_FRACTION
    if "packed" in w:   # documents of a packed row: a query attends to its own sample only
        pk = w["packed"]
        ATTN_PAIR_FRACTION = pk["samples"] * (pk["txt"] + pk["img"]) ** 2 / float(L * L)
    seed = 42 + rank  # reference seeding: main.py:1058-1068
    torch.manual_seed(seed)
    cfg, diff = build(args.workload, device, args.dropout)
    sync = None
    if world > 1:
        from unidisc_amd import ddp

        ddp.broadcast_parameters(diff.backbone)
        sync = ddp.wrap(diff.backbone, mode=args.ddp_mode)
    batch = {k: v.to(device) for k, v in synthetic_batch(args.workload, B, seed).items()}
    timer = KernelTimer(sample=args.time_every)
    if not args.no_kernel_timing:
        timer.install()

    def step(i):
        diff.backbone.zero_grad(set_to_none=True)
        out = diff.training_step(batch, i)
        out.loss.backward()
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    hog = None
    if args.hog_cus > 0:
        from unidisc_amd import _lib
        hog_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        hog_stream = torch.cuda.Stream(device=device)
        _lib.call("udm_debug_cu_hog", int(args.hog_cus), hog_flag.data_ptr(), hog_stream.cuda_stream)
        hog = (hog_flag, hog_stream)
    try:
        for i in range(args.warmup):
            out = step(i)
        if sync is not None:
            # comm policy: with "auto" every schedule is timed for a few EXTRA untimed steps on all ranks and the fastest kept (ddp.py)
            fence()
            sync.autotune(lambda: step(0), steps=4, settle=1, sync_device=device)    # >= 3 % for the slowest rank or "overlap" stays; skipped past 60 s (ddp.py)
        if hog is None:
            fence()
        else:   # (a device-wide synchronise would wait for the spinning kernel)
            torch.cuda.current_stream().synchronize()
        if sync is not None:
            sync.measure_exposed = True
            wire0 = sync.bytes_on_wire
        timer.enabled = True
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one event per step boundary (median beside the mean)
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            out = step(args.warmup + i)
            marks[i + 1].record()
        if hog is None:
            fence()
        else:
            torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
    finally:
        if hog is not None:    # also on an exception / interrupt: a kernel still polling a freed flag hangs or faults the teardown
            hog[0][0] = 1      # release the held CUs
            hog[1].synchronize()
    timer.enabled = False
    per_step_ms = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
    median_ms = per_step_ms[len(per_step_ms) // 2] if len(per_step_ms) % 2 else 0.5 * (per_step_ms[len(per_step_ms) // 2 - 1] + per_step_ms[len(per_step_ms) // 2])
    loss = float(out.loss.detach())
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    tokens = world * B * L * args.steps
    value = tokens / dt
    m = cfg.model
    L_att = (w["packed"]["txt"] + w["packed"]["img"]) if "packed" in w else L   # keys a query attends to
    f_tok = flops_per_token(m.n_blocks, m.hidden_size, diff.vocab_size, L_att)
    result = {
        "metric": "denoising tokens/sec (fwd+bwd), 1.4B DiT seq_len=1280" if args.workload == "unidisc-1.4b-l1280" else f"denoising tokens/sec (fwd+bwd), {args.workload}",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "ms_per_step_median": median_ms, "ms_per_step_min": per_step_ms[0], "ms_per_step_max": per_step_ms[-1],   # GPU time between step-boundary events on this rank
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": w["desc"], "per_gpu_batch": B, "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}",
                   "dropout": args.dropout, "weights": "random init (zero_linear_init=false)",
                   "attention_forward": "bf16", **({"cus_held_by_a_spinning_kernel": args.hog_cus} if args.hog_cus else {})},
        "tokens_per_s_per_gpu": value / world, "loss": loss, "flops_per_token": f_tok,
        "step_mfu": (value / world) * f_tok / (PEAK_BF16_DENSE_TFLOPS * 1e12),   # dense head, no recompute credit
    }
    gs = timer.summary()
    if gs:
        ach = gs["flops"] / (gs["total_ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic("gemm_") if args.workload == "unidisc-1.4b-l1280" else (None, None)
        sustained, sustained_src = sustained_mfma_reference()
        result["roofline"] = {"bound": "mfma", "kernel": "GEMM family (udm_gemm_nt_bf16 / udm_gemm_tn_bf16 / udm_gemm_nn_bf16 and the split-K forms)", "achieved": ach,
                              "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach / PEAK_BF16_DENSE_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "launches": gs["launches"],
                              "launches_in_timed_region": gs["launches_seen"], "timing": f"HIP events around 1 launch in {timer.sample} (seeded pick) inside the timed region",
                              "avg_launch_ms": gs["total_ms"] / gs["launches"],
                              "share_of_step_time": gs["total_ms"] * (gs["launches_seen"] / gs["launches"]) * 1e-3 / dt,
                              # context, not the contract's peak: what a bare v_mfma_f32_32x32x16_bf16 stream SUSTAINS on this part (power-limited clock), measured by
                              # experiments/ubench/mfma_power.hip (profiles/r05_mfma_power.log; 1.78-1.88 PF over the pool's boxes); 2.36-2.41 PF on all-zero operands
                              # the constant is READ from that committed log (another box of the pool), not measured in this run: its source travels with it
                              "sustained_mfma_random_bf16_tflops": sustained, "sustained_mfma_source": sustained_src,
                              "frac_of_sustained": (ach / sustained) if sustained else None}
    result["rccl_world"] = world if (world > 1 and os.environ.get("UDM_DIST_BACKEND", "nccl") == "nccl") else (1 if world == 1 else 0)   # ranks joined over RCCL (0: gloo rehearsal)
    if sync is not None:
        result["allreduce_bytes_per_step"] = (sync.bytes_on_wire - wire0) // args.steps
        result["ddp_mode"] = sync.mode
        result["ddp_mode_requested"] = sync.requested_mode
        result["ddp_mode_warmup_ms"] = sync.mode_timings_ms       # {mode: ms per step, max over ranks} from the auto-selection, else null
        result["ddp_reserved_cus"] = sync.reserved_cus            # what overlap_planned leaves to the collective
        result["ddp_mode_decision"] = sync.autotune_report        # margin of the fastest schedule over "overlap", steps per mode, or why the selection was skipped
        result["rccl_max_nchannels"] = rccl_channels              # NCCL_MAX_NCHANNELS in force (0 = RCCL's default)
        # time the compute stream spent waiting for the comm stream at the end of backward (events around BucketedGradSync.finish)
        result["exposed_comm_ms_per_step"] = sync.exposed_ms() / args.steps
    if not args.no_kernel_timing and args.table_steps > 0:
        # post-pass, OUTSIDE the timed region: the same step with EVERY launch bracketed by events -> one roofline row per C-ABI entry point
        timer.uninstall()
        full = KernelTimer(names=None)
        full.install()
        full.enabled = True
        for i in range(args.table_steps):
            n0 = len(full.records)
            step(args.warmup + args.steps + i)
            # SUBS cross-entropy: the algorithmic bytes are ONE bf16 read of the valid id range of every [MASK] row (text rows: the text ids, image rows:
            # the image ids; backward: one read + one write of the same range), not whole padded rows
            last = diff._last
            masked = last["xt"] == diff.mask_index
            mod = last.get("modality")
            if mod is not None and diff._restrict():
                n_img = int((masked & (mod > 0)).sum())
                n_txt = int(masked.sum()) - n_img
                ce_bytes = 2.0 * (n_txt * diff.text_vocab_size + n_img * (diff.vocab_size - diff.text_vocab_size))
            else:
                ce_bytes = 2.0 * int(masked.sum()) * diff.vocab_size
            for j in range(n0, len(full.records)):
                s_, e_, name_, w_ = full.records[j]
                if name_ == "udm_subs_ce_fwd":
                    full.records[j] = (s_, e_, name_, ("byte", ce_bytes))
                elif name_ == "udm_subs_ce_bwd":
                    full.records[j] = (s_, e_, name_, ("byte", 2.0 * ce_bytes))
        fence()
        full.enabled = False
        full.uninstall()
        result["roofline_table"] = full.table(args.table_steps)
        ex = sum(w[1] for _, _, _, w in full.records if w and w[0] == "flop") / args.table_steps   # everything the matrix pipe executed in a step
        result["executed_mfma_flops_per_step"] = ex
        result["executed_mfma_utilization"] = ex / (dt / args.steps) / (PEAK_BF16_DENSE_TFLOPS * 1e12)
    if world > 1:
        # Every rank leaves the process group BEFORE rank 0's CPU comparator (round 6): a 30 s host-only leg must not sit between the peers' communicator
        # teardown and rank 0's.  The contract times the CPU path "on rank 0 at N = 1 only"; `--cpu-baseline-any-world` adds it to an N > 1 line (rehearsals).
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    want_cpu = not args.no_cpu_baseline and (world == 1 or args.cpu_baseline_any_world)
    if rank == 0 and want_cpu:
        try:
            result["cpu_baseline"] = cpu_baseline(args.workload, cfg, diff, seed)
            if world == 1 and "packed" not in w:
                result["cpu_baseline"]["legs"] = cpu_baseline_legs(seed)
        except Exception as e:  # the GPU number stands on its own; say why the comparator is missing
            result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    elif rank == 0 and world > 1:
        result["cpu_baseline"] = None      # measured by the N = 1 line of the same command (contract: rank 0 at N = 1 only)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
