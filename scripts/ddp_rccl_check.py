#!/usr/bin/env python3
"""One rank, REAL RCCL: the data-parallel gradient path of `unidisc_amd.ddp` with backend "nccl" (= RCCL on ROCm) at world_size 1.

    python scripts/ddp_rccl_check.py            (prints one JSON line; exit code 1 on a failed check)

A one-GPU box cannot show scaling, but it can prove the path RUNS under RCCL: process-group init on the device, buckets reported from inside the HIP
backward, bf16 compress on the comm stream, `all_reduce` through RCCL's own kernels (world 1 still launches them), decompress, the event hand-offs
between the compute and comm streams, the accumulate-then-sync path and the persistent-GEMM CU reservation (`UDM_GEMM_CUS`).  At world 1 the
reduction is the identity, so the synchronised gradients must equal bf16(local gradients) EXACTLY (the reference hook's compression, main.py:641-656).
tests/test_gpu_ddp_rccl.py runs this script in a child process.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from product_utils import product_config
    from unidisc_amd import Diffusion, ddp

    ddp.rccl_channel_env()      # NCCL_MAX_NCHANNELS before the communicator exists, as bench.py does
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    case = dict(hidden_size=768, n_heads=12, cond_dim=128, n_blocks=3, batch_size=4, txt_length=128, img_length=256, text_vocab_size=32001,
                vocab_size=40193, norm_type="rms", qk_norm=True, sandwich_normalization=True, modality_embed=True, rope_2d=False,
                time_conditioning=False, multimodal_batches=True, force_argmax_valid_indices=True, mask_entire_modality=0.1, softmin_snr=5,
                text_loss_weight=1.0, img_loss_weight=None, force_full_attention_mask_loss_only=True)
    torch.manual_seed(0)
    diff = Diffusion(product_config(case), None, dev)
    diff.backbone.train()
    diff.rng_device = "cpu"
    ddp.broadcast_parameters(diff.backbone)
    gen = torch.Generator().manual_seed(5)
    B = 4

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        return dict(txt_input_ids=torch.randint(0, 32000, (B, 128), generator=g, dtype=torch.int32),
                    img_input_ids=torch.randint(0, 8192, (B, 256), generator=g, dtype=torch.int32).to(torch.int16),
                    txt_attention_mask=torch.ones(B, 128, dtype=torch.bool))

    def grads(seed, zero=True):
        if zero:
            diff.backbone.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        out = diff.training_step(batch(seed), 1)
        out.loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in diff.backbone.named_parameters()}

    def bf16(x):
        return x.to(torch.bfloat16).float()

    def worst(a, b):
        return max(float((a[k] - b[k]).abs().max() / (b[k].abs().max() + 1e-30)) for k in a)

    res, fails = {}, []
    local = grads(1)
    for min_bucket in (1, 4 * 1024 * 1024, 1 << 30):
        if getattr(diff.backbone, "_grad_sync", None) is not None:
            del diff.backbone._grad_sync
        sync = ddp.wrap(diff.backbone, min_bucket_elems=min_bucket, force_single_rank=True)
        sync.measure_exposed = True
        synced = grads(1)
        # the backward is not bit-reproducible run to run (fp32 atomics in column reductions), so two statements: the synchronised values ARE
        # bf16-representable (they went through the wire format), and they equal bf16(local) up to that run-to-run noise
        repr_ok = all(torch.equal(synced[k], bf16(synced[k])) for k in synced)
        err = worst(synced, {k: bf16(v) for k, v in local.items()})
        res[f"bucket{min_bucket}"] = dict(bf16_representable=repr_ok, worst_rel_vs_bf16_local=err, bytes_on_wire=sync.bytes_on_wire,
                                         exposed_ms=sync.exposed_ms())
        if not repr_ok or err > 1e-2 or sync.bytes_on_wire == 0:
            fails.append(("sync", min_bucket, repr_ok, err))
    # comm policy (round 4): every schedule under real RCCL - same wire-format gradients; the planned one switches the GEMM CU plan on at the first
    # bucket and off at the end of the backward; autotune times the three and keeps one
    if getattr(diff.backbone, "_grad_sync", None) is not None:
        del diff.backbone._grad_sync
    sync = ddp.wrap(diff.backbone, min_bucket_elems=4 * 1024 * 1024, force_single_rank=True, mode="auto")
    from unidisc_amd import kernels as K
    for mode in ddp.MODES:
        sync.set_mode(mode)
        w0 = sync.bytes_on_wire
        got = grads(1)
        repr_ok = all(torch.equal(got[k], bf16(got[k])) for k in got)
        err = worst(got, {k: bf16(v) for k, v in local.items()})
        res[f"mode_{mode}"] = dict(bf16_representable=repr_ok, worst_rel_vs_bf16_local=err, bytes_on_wire=sync.bytes_on_wire - w0, gemm_cus_after=K._CUS[0])
        if not repr_ok or err > 1e-2 or sync.bytes_on_wire == w0 or K._CUS[0] != 0:
            fails.append(("mode", mode, repr_ok, err, K._CUS[0]))
    table = sync.autotune(lambda: grads(1), steps=2, settle=1, sync_device=dev)
    res["autotune"] = dict(table_ms=table, chosen=sync.mode, reserved_cus=sync.reserved_cus, nccl_max_nchannels=os.environ.get("NCCL_MAX_NCHANNELS"))
    if table is None or set(table) != set(ddp.MODES) or sync.mode not in ddp.MODES:
        fails.append(("autotune", table, sync.mode))
    # no_sync keeps gradients local (NOT bf16-rounded), then accumulate-then-sync reduces the accumulated sum
    sync.enabled = False
    g1 = grads(1)
    if all(torch.equal(g1[k], bf16(g1[k])) for k in g1):
        fails.append(("no_sync gradients went through the wire",))
    other = grads(2)
    grads(1)
    sync.enabled = True
    acc = grads(2, zero=False)
    exp = {k: bf16(g1[k] + other[k]) for k in g1}
    acc_err = worst(acc, exp)
    acc_repr = all(torch.equal(acc[k], bf16(acc[k])) for k in acc)
    res["accumulate_then_sync"] = dict(bf16_representable=acc_repr, worst_rel_vs_bf16_sum=acc_err)
    if not acc_repr or acc_err > 1e-2:
        fails.append(("accumulate", acc_repr, acc_err))
    # persistent GEMM grid with CUs left free for RCCL's channels: same results
    K.gemm_set_cus(224)
    red = grads(1)
    K.gemm_set_cus(0)
    cu_err = worst(red, {k: bf16(v) for k, v in local.items()})
    res["gemm_cus_224"] = dict(worst_rel_vs_bf16_local=cu_err)
    if cu_err > 1e-2:
        fails.append(("gemm_cus", cu_err))
    # sharded path (unidisc_amd/zero.py: reduce-to-owner, owner-only AdamW, broadcast of the masters) under RCCL: one rank owns every bucket, so three
    # steps must track the replicated path's (all-reduce + FusedAdamW) parameters from the same start and batches
    from unidisc_amd import FusedAdamW, zero
    start = {k: v.detach().clone() for k, v in diff.backbone.state_dict().items()}

    def three_steps(make):
        diff.backbone.load_state_dict(start)
        diff.backbone.invalidate_shadows()
        if getattr(diff.backbone, "_grad_sync", None) is not None:
            del diff.backbone._grad_sync
        opt = make()
        for it in range(3):
            diff.backbone.zero_grad(set_to_none=True)
            torch.manual_seed(20 + it)
            diff.training_step(batch(20 + it), 1).loss.backward()
            opt.step()
        torch.cuda.synchronize()
        return {k: p.detach().clone() for k, p in diff.backbone.named_parameters()}, opt

    def replicated():
        ddp.wrap(diff.backbone, min_bucket_elems=4 * 1024 * 1024, force_single_rank=True)
        return FusedAdamW(diff.backbone, lr=1e-3, max_grad_norm=1.0)

    def sharded():
        sy = zero.wrap_sharded(diff.backbone, min_bucket_elems=4 * 1024 * 1024, force_single_rank=True)
        return zero.ShardedAdamW(diff.backbone, sy, lr=1e-3, max_grad_norm=1.0)

    p_rep, _ = three_steps(replicated)
    p_rep2, _ = three_steps(replicated)
    p_sh, opt_sh = three_steps(sharded)

    def rms_rel(a, b):   # ||a - b|| over all parameters, relative to the size of the three updates
        num = sum(float(((a[k] - b[k]).double() ** 2).sum()) for k in a)
        den = sum(float(((b[k] - start[k]).double() ** 2).sum()) for k in a)
        return (num / (den + 1e-300)) ** 0.5

    # the backward is not bit-reproducible run to run (fp32 atomics) and Adam turns a sign flip of a near-zero gradient into a full-size update of that
    # element: the yardstick is the replicated path against ITSELF
    noise, z_err = rms_rel(p_rep2, p_rep), rms_rel(p_sh, p_rep)
    res["sharded_optimizer"] = dict(rms_diff_vs_replicated_rel_to_update=z_err, replicated_run_to_run=noise, buckets=len(opt_sh.sync.ranges),
                                    grad_norm=float(opt_sh.grad_norm), bytes_on_wire=opt_sh.sync.bytes_on_wire)
    if z_err > 2.0 * noise + 0.02 or opt_sh.sync.bytes_on_wire == 0:
        fails.append(("sharded", z_err, noise))
    dist.barrier()
    dist.destroy_process_group()
    res["backend"] = "nccl (RCCL)"
    res["ok"] = not fails
    res["fails"] = [str(f) for f in fails]
    print(json.dumps(res), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
