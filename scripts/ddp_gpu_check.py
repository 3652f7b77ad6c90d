#!/usr/bin/env python3
"""Two ranks on ONE MI355X: the data-parallel gradient path with the real HIP kernels, device tensors and the comm stream.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 scripts/ddp_gpu_check.py

RCCL refuses two ranks on one device, so the process group here is gloo (it moves device tensors through the host); everything else is the product
path that runs under RCCL on an 8-GPU node: `BucketedGradSync` buckets reported from inside the HIP backward, the bf16 compress / decompress
kernels on the comm stream, the event hand-offs between the two streams.  Checks (same as tests/test_ddp_gloo.py, which uses kernel doubles on CPU):
synchronised gradients equal the bf16-compressed mean of the ranks' local gradients, all ranks hold identical values, no_sync keeps them local.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def grads(diff, golden, seed):
    torch.manual_seed(seed)
    batch = golden.batch()
    g = torch.Generator().manual_seed(1000 + seed)
    batch["txt_input_ids"] = torch.randint(0, golden.case["text_vocab_size"] - 1, batch["txt_input_ids"].shape, generator=g, dtype=torch.int32)
    diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, 1)
    out.loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().float().cpu().clone() for k, p in diff.backbone.named_parameters()}


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from golden_utils import Golden
    from product_utils import build_product
    from unidisc_amd import ddp

    fails = []
    for name, min_bucket in (("c_large", 1), ("c_large", 1 << 30), ("d_adaln_mm", 4096)):
        golden = Golden(name)
        diff = build_product(golden, "cuda")
        diff.rng_device = "cpu"
        ddp.broadcast_parameters(diff.backbone)
        local = grads(diff, golden, seed=rank)
        sync = ddp.wrap(diff.backbone, min_bucket_elems=min_bucket)
        synced = grads(diff, golden, seed=rank)
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        worst = 0.0
        for k in local:
            exp = sum((g[k].to(torch.bfloat16).float() / world).to(torch.bfloat16).float() for g in gathered).to(torch.bfloat16).float()
            scale = exp.abs().max().item() + 1e-12
            err = (synced[k] - exp).abs().max().item() / scale
            worst = max(worst, err)
            if err > 2e-2:
                fails.append((name, min_bucket, k, err))
        flat = torch.cat([synced[k].flatten() for k in sorted(synced)])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        if not torch.equal(flat, ref):
            fails.append((name, min_bucket, "ranks differ"))
        sync.enabled = False
        unsynced = grads(diff, golden, seed=rank)
        # the device backward is not bit-reproducible run to run (fp32 atomics in the embedding / column reductions), so: close to the local
        # gradients of the first run, and not the synchronised values
        dl = max(((unsynced[k] - local[k]).abs().max() / (local[k].abs().max() + 1e-12)).item() for k in local)
        ds = max(((unsynced[k] - synced[k]).abs().max() / (local[k].abs().max() + 1e-12)).item() for k in local)
        if dl > 1e-3 or ds < 10 * dl:
            fails.append((name, min_bucket, f"no_sync changed gradients (vs local {dl:.2e}, vs synced {ds:.2e})"))
        print(f"[rank {rank}] {name} min_bucket={min_bucket}: worst rel err {worst:.3e}, bytes on wire {sync.bytes_on_wire}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if fails:
        print(f"[rank {rank}] FAILED: {fails[:6]}", flush=True)
        sys.exit(1)
    print(f"[rank {rank}] ddp_gpu_check OK", flush=True)


if __name__ == "__main__":
    main()
