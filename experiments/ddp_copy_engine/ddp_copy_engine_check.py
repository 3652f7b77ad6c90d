#!/usr/bin/env python3
"""Two ranks on ONE MI355X: the copy-engine gradient exchange (unidisc_amd/ddp_copy.py, `UDM_DDP_MODE=copy_engine`) with the real HIP backward.

    python scripts/ddp_copy_engine_check.py          (spawns its two ranks; prints one JSON line; exit code 1 on a failed check)

What one GPU can prove: device buffers exported / imported across processes (CUDA IPC), the reduce-scatter + all-gather as `copy_()` into PEER buffers on a
copy stream, the helper thread's host fences (gloo control plane), the sum and cast kernels, the join at the end of the backward - and that the result is
the reference hook's value: every rank ends with sum_ranks bf16(bf16(g) / world) (fp32-accumulated, rounded once), identical on all ranks, for small and
large buckets and for the accumulate-then-sync path.  What it cannot show: xGMI bandwidth and peer access between different devices."""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grads(diff, golden, seed, zero=True):
    torch.manual_seed(seed)
    batch = golden.batch()
    g = torch.Generator().manual_seed(1000 + seed)
    batch["txt_input_ids"] = torch.randint(0, golden.case["text_vocab_size"] - 1, batch["txt_input_ids"].shape, generator=g, dtype=torch.int32)
    if zero:
        diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, 1)
    out.loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().float().cpu().clone() for k, p in diff.backbone.named_parameters()}


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from golden_utils import Golden
    from product_utils import build_product
    from unidisc_amd import ddp

    res, fails = {}, []
    try:
        for name, min_bucket in (("c_large", 1), ("c_large", 1 << 30), ("d_adaln_mm", 4096)):
            golden = Golden(name)
            diff = build_product(golden, "cuda")
            diff.rng_device = "cpu"
            ddp.broadcast_parameters(diff.backbone)
            local = grads(diff, golden, seed=rank)
            sync = ddp.wrap(diff.backbone, min_bucket_elems=min_bucket, mode="copy_engine")
            synced = grads(diff, golden, seed=rank)
            if sync._cx is None:
                fails.append((name, "the exchange was not set up (fell back to RCCL)"))
                break
            gathered = [None] * world
            dist.all_gather_object(gathered, local)
            worst = 0.0
            for k in local:
                exp = sum((g[k].to(torch.bfloat16).float() / world).to(torch.bfloat16).float() for g in gathered).to(torch.bfloat16).float()
                err = (synced[k] - exp).abs().max().item() / (exp.abs().max().item() + 1e-12)
                worst = max(worst, err)
            flat = torch.cat([synced[k].flatten() for k in sorted(synced)])
            ref = flat.clone()
            dist.broadcast(ref, src=0)
            same = bool(torch.equal(flat, ref))
            # accumulate-then-sync through the same exchange
            sync.enabled = False
            other = grads(diff, golden, seed=10 + rank)
            grads(diff, golden, seed=rank)
            sync.enabled = True
            acc = grads(diff, golden, seed=10 + rank, zero=False)
            summed = {k: local[k] + other[k] for k in local}
            dist.all_gather_object(gathered, summed)
            acc_worst = 0.0
            for k in local:
                exp = sum((g[k].to(torch.bfloat16).float() / world).to(torch.bfloat16).float() for g in gathered).to(torch.bfloat16).float()
                acc_worst = max(acc_worst, (acc[k] - exp).abs().max().item() / (exp.abs().max().item() + 1e-12))
            res[f"{name}_bucket{min_bucket}"] = dict(worst_rel_vs_bf16_mean=worst, ranks_identical=same, accumulate_worst_rel=acc_worst,
                                                     bytes_pushed_to_peers=sync._cx.bytes_copied, host_wait_ms=1e3 * sync._cx.host_wait_s)
            # the device backward is not bit-reproducible run to run (fp32 atomics): 2e-2 of the parameter's scale, as scripts/ddp_gpu_check.py
            if worst > 2e-2 or not same or acc_worst > 2e-2 or sync._cx.bytes_copied == 0:
                fails.append((name, min_bucket, worst, same, acc_worst))
            sync._cx.close()
            del diff
    except Exception as e:
        fails.append(f"{type(e).__name__}: {e}"[:400])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res, [str(f) for f in fails]))


def main():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(60)
    fails = [f for _, _, fl in out for f in fl]
    print(json.dumps(dict(ok=not fails, fails=fails, rank0=out[0][1], rank1=out[1][1], control_plane="gloo", ranks_on_one_gpu=2)), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
