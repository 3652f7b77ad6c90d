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

WORKLOADS = {
    # BASELINE.json configs[2]: large_scale_train + large_scale_train_high_res, data.block_size=256
    "unidisc-1.4b-l1280": dict(preset="extra_large", txt_length=256, img_length=1024, text_vocab=32001, image_vocab=16384, batch=8,
                               desc="UniDisc 1.4B non-interleaved, seq_len=1280 (256 text + 1024 image), bf16"),
    # BASELINE.json configs[1]: UniDisc-S
    # BASELINE.json configs[4] in bf16 (no fp8 attention yet): large_scale_train_high_res_interleaved with model.length=4608:
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


def flops_per_token(n_blocks, d, V, L_att):
    """F_tok = 6·P_mm + 12·n·L_att·d with P_mm = n·12d² + d·V (matmul weights only; no recompute credit)."""
    p_mm = n_blocks * 12 * d * d + d * V
    return 6 * p_mm + 12 * n_blocks * L_att * d


def build(workload, device, dropout):
    from unidisc_amd import MODEL_PRESETS, Diffusion, make_config

    w = WORKLOADS[workload]
    large = w["preset"] == "extra_large"
    cfg = make_config(**MODEL_PRESETS[w["preset"]], txt_length=w["txt_length"], img_length=w["img_length"], norm_type="rms", qk_norm=True,
                      sandwich_normalization=True, modality_embed=True, rope_2d=large, linear_factor=2.0 if large else 1.0, time_conditioning=False,
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


class GemmTimer:
    """HIP-event timing of every udm_gemm_nt_bf16 launch, on the stream it is launched on (torch's current stream)."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def install(self):
        from unidisc_amd import _lib

        orig = _lib.call
        timer = self

        def call(name, *args):
            if timer.enabled and name in ("udm_gemm_nt_bf16", "udm_gemm_tn_bf16"):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                orig(name, *args)
                e.record()
                timer.records.append((s, e, 2.0 * args[3] * args[4] * args[5]))
            else:
                orig(name, *args)

        _lib.call = call

    def summary(self):
        if not self.records:
            return None
        ms = sum(s.elapsed_time(e) for s, e, _ in self.records)
        fl = sum(f for _, _, f in self.records)
        return dict(launches=len(self.records), total_ms=ms, flops=fl)


def cpu_baseline(workload, cfg, diff, seed):
    """The oracle (CPU restatement of the reference path) timed on this host's cores on a BOUNDED sample of the same workload:
    one sequence (B=1, full L and width, full vocabulary head) through the first 1 and the first 2 DiT blocks, fwd+bwd.
    Blocks are identical, so the 24-block time is extrapolated as t(1) + (n-1)·(t(2) - t(1)); the sample itself is ~10-30 s."""
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
    times = {}
    losses = {}
    for nb in (1, 2):
        case = dict(hidden_size=m.hidden_size, n_heads=m.n_heads, cond_dim=m.cond_dim, n_blocks=nb, txt_length=w["txt_length"],
                    img_length=w["img_length"], vocab_size=diff.vocab_size, text_vocab_size=diff.text_vocab_size, norm_type="rms", qk_norm=True,
                    sandwich_normalization=True, modality_embed=True, rope_2d=m.rope_2d, linear_factor=m.linear_factor, time_conditioning=False,
                    multimodal_batches=True, force_argmax_valid_indices=True, mask_entire_modality=0.1, softmin_snr=5, text_loss_weight=1.0,
                    img_loss_weight=cfg.trainer.get("img_loss_weight"), force_full_attention_mask=cfg.trainer.get("force_full_attention_mask"),
                    force_full_attention_mask_loss_only=cfg.trainer.get("force_full_attention_mask_loss_only"))
        ocfg = O.OracleConfig.from_case(case)
        keep = lambda k: not k.startswith("blocks.") or int(k.split(".")[1]) < nb
        P = {k: v.detach().float().cpu().requires_grad_() for k, v in diff.backbone.named_parameters() if keep(k)}
        bufs = O.make_buffers(ocfg, lumina_rope_2d)
        batch = O.update_batch(ocfg, synthetic_batch(workload, 1, seed))
        t0 = time.perf_counter()
        out = O.compute_loss(ocfg, P, bufs, batch, torch.Generator().manual_seed(seed))
        out.loss.backward()
        times[nb] = time.perf_counter() - t0
        losses[nb] = float(out.loss)
        del P, out
        if times[nb] > 90:  # very slow host: do not run the second point, assume the head costs as much as one block
            times[2] = 1.5 * times[1]
            break
    per_block = max(times[2] - times[1], 1e-6)
    total = times[1] + (m.n_blocks - 1) * per_block
    return dict(value=L / total, unit="tokens/s", cores=threads, kind="port",
                sample=(f"oracle (fp32 torch CPU) fwd+bwd of 1 sequence (B=1, L={L}, d={m.hidden_size}, V={diff.vocab_size}) through 1 and 2 of the "
                        f"{m.n_blocks} blocks: {times[1]:.1f} s and {times[2]:.1f} s; extrapolated to {m.n_blocks} blocks = {total:.1f} s "
                        f"({threads} threads of {cores} available cores)"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="unidisc-1.4b-l1280", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: workload's)")
    ap.add_argument("--dropout", type=float, default=0.1, help="model.dropout (reference extra_large.yaml: 0.1)")
    ap.add_argument("--fp8-attention", action="store_true", help="attention forward through the fp8 kernel (config E option; changes numerics, not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed.run launch with {args.gpus} ranks (WORLD_SIZE is 1)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    local_rank %= torch.cuda.device_count()  # one rank per GPU on a node; wraps only in the single-GPU rehearsal of the N > 1 path (below)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("UDM_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only to rehearse the N > 1 path with two ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    L = w["txt_length"] + w["img_length"]
    seed = 42 + rank  # reference seeding: main.py:1058-1068
    torch.manual_seed(seed)
    cfg, diff = build(args.workload, device, args.dropout)
    diff.backbone.fp8_attention = bool(args.fp8_attention)
    sync = None
    if world > 1:
        from unidisc_amd import ddp

        ddp.broadcast_parameters(diff.backbone)
        sync = ddp.wrap(diff.backbone)
    batch = {k: v.to(device) for k, v in synthetic_batch(args.workload, B, seed).items()}
    timer = GemmTimer()
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

    for i in range(args.warmup):
        out = step(i)
    fence()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
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
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": w["desc"], "per_gpu_batch": B, "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}",
                   "dropout": args.dropout, "weights": "random init (zero_linear_init=false)",
                   "attention_forward": "fp8 e4m3" if args.fp8_attention else "bf16"},
        "tokens_per_s_per_gpu": value / world, "loss": loss, "flops_per_token": f_tok,
        "step_mfu": (value / world) * f_tok / (PEAK_BF16_DENSE_TFLOPS * 1e12),
    }
    gs = timer.summary()
    if gs:
        ach = gs["flops"] / (gs["total_ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic("gemm_nt_stagger_kernel") if args.workload == "unidisc-1.4b-l1280" else (None, None)
        result["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_stagger_kernel (udm_gemm_nt_bf16 / udm_gemm_tn_bf16)", "achieved": ach, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach / PEAK_BF16_DENSE_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "launches": gs["launches"],
                              "avg_launch_ms": gs["total_ms"] / gs["launches"], "share_of_step_time": gs["total_ms"] * 1e-3 / dt}
    if sync is not None:
        result["allreduce_bytes_per_step"] = sync.bytes_on_wire // (args.steps + args.warmup)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and "packed" not in w:
        try:
            result["cpu_baseline"] = cpu_baseline(args.workload, cfg, diff, seed)
        except Exception as e:  # the GPU number stands on its own; say why the comparator is missing
            result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
