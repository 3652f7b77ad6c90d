"""world_size-2 gloo test of the sharded data-parallel path (ZeRO-2-style owner buckets; CPU, kernels replaced by test doubles):
reduce-to-owner + owner-only AdamW + broadcast must leave every rank with the parameters the replicated path (bucketed all-reduce + AdamW on every
rank) produces, with Adam moments only for the owned parameters."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
# The two paths sum their bf16 buckets in different fp32 orders (1e-7 relative on a gradient); with Adam's default eps = 1e-8 a parameter whose gradient is
# rounding noise (a qk-norm bias: column sums with heavy cancellation) gets steps of +- lr whose SIGN that noise decides - a coin flip, not a property of the
# sharding.  eps well above the noise makes the comparison about the algebra.
ADAM_EPS = 1e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _step(diff, golden, seed):
    torch.manual_seed(seed)
    batch = golden.batch()
    g = torch.Generator().manual_seed(1000 + seed)
    batch["txt_input_ids"] = torch.randint(0, golden.case["text_vocab_size"] - 1, batch["txt_input_ids"].shape, generator=g, dtype=torch.int32)
    out = diff.training_step(batch, 1)
    out.loss.backward()


def _worker(rank, world, port, min_bucket, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fake_kernels
        import unidisc_amd.dit as dit_mod
        import unidisc_amd.diffusion as diff_mod
        import unidisc_amd.ddp as ddp_mod
        import unidisc_amd.optim as optim_mod
        import unidisc_amd.zero as zero_mod
        from golden_utils import Golden
        from product_utils import build_product

        dit_mod.K = diff_mod.K = ddp_mod.K = optim_mod.K = zero_mod.K = fake_kernels
        golden = Golden("c_large")
        # replicated path: all-reduce + AdamW everywhere
        ref = build_product(golden, "cpu")
        ddp_mod.broadcast_parameters(ref.backbone)
        ddp_mod.wrap(ref.backbone, min_bucket_elems=min_bucket)
        ref_opt = optim_mod.FusedAdamW(ref.backbone, lr=1e-2, eps=ADAM_EPS, max_grad_norm=0.5, maintain_shadows=False)
        # sharded path from the same start
        sh = build_product(golden, "cpu")
        sh.backbone.load_state_dict(ref.backbone.state_dict())
        sync = zero_mod.wrap_sharded(sh.backbone, min_bucket_elems=min_bucket)
        opt = zero_mod.ShardedAdamW(sh.backbone, sync, lr=1e-2, eps=ADAM_EPS, max_grad_norm=0.5, maintain_shadows=False)
        ok, worst = True, 0.0
        for it in range(3):
            for d, o in ((ref, ref_opt), (sh, opt)):
                d.backbone.zero_grad(set_to_none=True)
                _step(d, golden, seed=10 * it + rank)
                o.step()
            for (k, a), (_, b) in zip(ref.backbone.named_parameters(), sh.backbone.named_parameters()):
                err = (a - b).abs().max().item()
                worst = max(worst, err / (a.abs().max().item() + 1e-12))
                ok = ok and err <= 1e-6 + 1e-5 * a.abs().max().item()
        flat = torch.cat([p.detach().flatten() for p in sh.backbone.parameters()])
        other = flat.clone()
        dist.broadcast(other, src=0)
        same = torch.equal(flat, other)                       # every rank holds identical masters
        n_owned = sum(p.numel() for p in opt.params if id(p) in opt.state)
        n_all = sum(p.numel() for p in opt.params)
        owners = sorted(set(opt._owner.values()))
        norm_same = abs(float(opt.grad_norm) - float(ref_opt.grad_norm)) <= 1e-4 * float(ref_opt.grad_norm)
        sd = opt.state_dict()
        opt2 = zero_mod.ShardedAdamW(sh.backbone, sync, lr=1e-2, eps=ADAM_EPS, max_grad_norm=0.5, maintain_shadows=False)
        opt2.load_state_dict(sd)
        rt = all(torch.equal(opt2.state[id(p)][0], opt.state[id(p)][0]) for p in opt.params if id(p) in opt.state) and opt2.step_count == opt.step_count
        q.put((rank, ok, worst, same, n_owned, n_all, owners, norm_same, rt, len(sync.ranges)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("min_bucket", [1, 20000])
def test_sharded_optimizer_world2(min_bucket):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, min_bucket, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned_total = 0
    for rank, ok, worst, same, n_owned, n_all, owners, norm_same, rt, nb in res:
        assert ok, (rank, worst)          # same parameters as the replicated path after three steps
        assert same, rank                 # identical on every rank
        assert owners == [0, 1] and nb >= 2, (owners, nb)
        assert 0 < n_owned < n_all, (n_owned, n_all)     # moments for the owned parameters only
        assert norm_same and rt, rank
        owned_total += n_owned
    assert owned_total == res[0][5]       # the two shards partition the parameters


def _worker_acc_ema(rank, world, port, q):
    """gradient accumulation (two micro-steps per optimizer step, the first with sync.enabled = False) and the parameter EMA on the sharded path"""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fake_kernels
        import unidisc_amd.dit as dit_mod
        import unidisc_amd.diffusion as diff_mod
        import unidisc_amd.ddp as ddp_mod
        import unidisc_amd.optim as optim_mod
        import unidisc_amd.zero as zero_mod
        from golden_utils import Golden
        from product_utils import build_product

        dit_mod.K = diff_mod.K = ddp_mod.K = optim_mod.K = zero_mod.K = fake_kernels
        golden = Golden("c_large")
        ref = build_product(golden, "cpu")
        ddp_mod.broadcast_parameters(ref.backbone)
        rsync = ddp_mod.wrap(ref.backbone, min_bucket_elems=20000)
        ref_opt = optim_mod.FusedAdamW(ref.backbone, lr=1e-2, eps=ADAM_EPS, max_grad_norm=0.5, maintain_shadows=False, ema_decay=0.9)
        sh = build_product(golden, "cpu")
        sh.backbone.load_state_dict(ref.backbone.state_dict())
        sync = zero_mod.wrap_sharded(sh.backbone, min_bucket_elems=20000)
        opt = zero_mod.ShardedAdamW(sh.backbone, sync, lr=1e-2, eps=ADAM_EPS, max_grad_norm=0.5, maintain_shadows=False, ema_decay=0.9)
        ok, worst = True, 0.0
        for it in range(3):
            for d, o, sy in ((ref, ref_opt, rsync), (sh, opt, sync)):
                d.backbone.zero_grad(set_to_none=True)
                micro = 2 if it != 1 else 1          # (step 1 takes the in-backward path: ownership must be the same in both modes)
                for ms in range(micro):
                    sy.enabled = ms == micro - 1
                    _step(d, golden, seed=100 * it + 10 * ms + rank)
                o.step()
            for (k, a), (_, b) in zip(ref.backbone.named_parameters(), sh.backbone.named_parameters()):
                err = (a - b).abs().max().item()
                worst = max(worst, err / (a.abs().max().item() + 1e-12))
                ok = ok and err <= 1e-6 + 1e-5 * a.abs().max().item()
            # keep the two models in lockstep: a last-bit difference of the clipping norm (summation order) flips bf16 roundings of the weight shadows in the
            # NEXT forward, and Adam's normalised update turns such gradient noise into lr-sized parameter differences where a gradient is near zero
            sh.backbone.load_state_dict(ref.backbone.state_dict())
            sh.backbone.invalidate_shadows()
        # EMA: owned slices equal the replicated optimizer's EMA; store_and_copy hands every rank the full EMA weights; restore brings the masters back
        ema_ok = all(torch.allclose(opt.ema[id(p)], ref_opt.ema[id(rp)], rtol=1e-4, atol=1e-6)
                     for p, rp in zip(opt.params, ref_opt.params) if id(p) in opt.ema)
        n_ema = sum(p.numel() for p in opt.params if id(p) in opt.ema)
        before = [p.detach().clone() for p in sh.backbone.parameters()]
        opt.ema_store_and_copy()
        full_ema = all(torch.allclose(p, ref_opt.ema[id(rp)], rtol=1e-4, atol=1e-6) for p, rp in zip(opt.params, ref_opt.params))
        flat = torch.cat([p.detach().flatten() for p in sh.backbone.parameters()])
        other = flat.clone()
        dist.broadcast(other, src=0)
        same = torch.equal(flat, other)
        opt.ema_restore()
        restored = all(torch.equal(a, b) for a, b in zip(before, sh.backbone.parameters()))
        sd = opt.state_dict()
        q.put((rank, ok, worst, ema_ok, n_ema, full_ema, same, restored, "dropout_fwd_count" in sd and sd["ema"] is not None))
    finally:
        dist.destroy_process_group()


def test_sharded_accumulation_and_ema_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_acc_ema, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, worst, ema_ok, n_ema, full_ema, same, restored, sd_ok in res:
        assert ok, (rank, worst)                 # accumulated + sharded == accumulated + replicated
        assert ema_ok and n_ema > 0 and full_ema and same and restored and sd_ok, (rank, ema_ok, n_ema, full_ema, same, restored, sd_ok)
