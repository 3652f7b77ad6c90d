"""world_size-2 and world_size-8 gloo tests of the bucketed bf16 gradient all-reduce (CPU; kernels replaced by test doubles).  World 8 is BASELINE configs[3]'s
rank count (main.py:641-656): bf16(g) / 8 in bf16, bucket order, the collective mode agreement with ONE failing rank of eight, accumulate-then-sync."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads(diff, golden, seed, zero=True):
    torch.manual_seed(seed)
    batch = golden.batch()
    g = torch.Generator().manual_seed(1000 + seed)
    batch["txt_input_ids"] = torch.randint(0, golden.case["text_vocab_size"] - 1, batch["txt_input_ids"].shape, generator=g, dtype=torch.int32)
    if zero:
        diff.backbone.zero_grad(set_to_none=True)
    out = diff.training_step(batch, 1)
    out.loss.backward()
    return {k: p.grad.clone() for k, p in diff.backbone.named_parameters()}


def _worker(rank, world, port, min_bucket, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 2:
        torch.set_num_threads(1)      # eight ranks on an eight-core host
    rtol = 1e-2 * world               # gloo sums in bf16: one rounding (2^-9 relative) per add, world - 1 adds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fake_kernels
        import unidisc_amd.dit as dit_mod
        import unidisc_amd.diffusion as diff_mod
        import unidisc_amd.ddp as ddp_mod
        from golden_utils import Golden
        from product_utils import build_product

        dit_mod.K = diff_mod.K = ddp_mod.K = fake_kernels
        golden = Golden("c_large")
        diff = build_product(golden, "cpu")
        ddp_mod.broadcast_parameters(diff.backbone)
        local = _grads(diff, golden, seed=rank)            # unsynchronised local gradients
        sync = ddp_mod.wrap(diff.backbone, min_bucket_elems=min_bucket)
        synced = _grads(diff, golden, seed=rank)            # same step with the all-reduce hooked into backward
        # expected: every rank's local grads, bf16-compressed, divided by world, summed (reference BF16 hook)
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        ok, worst = True, 0.0
        for k in local:
            exp = sum((g[k].to(torch.bfloat16).float() / world).to(torch.bfloat16).float() for g in gathered)
            exp = exp.to(torch.bfloat16).float()
            err = (synced[k] - exp).abs().max().item()
            tol = rtol * exp.abs().max().item() + 1e-6
            worst = max(worst, err / (exp.abs().max().item() + 1e-12))
            ok = ok and err <= tol
        # all ranks hold identical synchronised gradients
        flat = torch.cat([synced[k].flatten() for k in sorted(synced)])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        same = torch.equal(flat, ref)
        sync.enabled = False                                   # no_sync micro-step: gradients stay local
        unsynced = _grads(diff, golden, seed=rank)
        local_ok = all(torch.equal(unsynced[k], local[k]) for k in local)
        # gradient accumulation (DDP no_sync recipe): micro-step 1 local, micro-step 2 synchronised -> every rank must end with
        # sum_ranks bf16(bf16(g1 + g2) / world), i.e. the ACCUMULATED gradients are what is reduced, and all ranks agree bit for bit
        other = _grads(diff, golden, seed=10 + rank)            # second micro-batch, gradients kept local (sync still disabled)
        sync.enabled = False
        _grads(diff, golden, seed=rank)                          # micro-step 1 (p.grad = local)
        sync.enabled = True
        acc = _grads(diff, golden, seed=10 + rank, zero=False)   # micro-step 2: p.grad = reduce(local + other)
        summed = {k: local[k] + other[k] for k in local}
        gathered = [None] * world
        dist.all_gather_object(gathered, summed)
        acc_ok, acc_worst = True, 0.0
        for k in local:
            exp = sum((g[k].to(torch.bfloat16).float() / world).to(torch.bfloat16).float() for g in gathered).to(torch.bfloat16).float()
            err = (acc[k] - exp).abs().max().item()
            acc_worst = max(acc_worst, err / (exp.abs().max().item() + 1e-12))
            acc_ok = acc_ok and err <= rtol * exp.abs().max().item() + 1e-6
        flat = torch.cat([acc[k].flatten() for k in sorted(acc)])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        acc_same = torch.equal(flat, ref)
        after = _grads(diff, golden, seed=rank)                  # and the next plain step takes the overlapped in-backward path again
        again_ok = all(torch.equal(after[k], synced[k]) for k in synced)
        # comm policy (round 4): every schedule ends with the SAME synchronised gradients; the planned one brackets the rest of the backward with a GEMM CU plan;
        # the serialized one sends nothing from inside the backward and coalesces ranges; autotune makes the same decision on every rank
        modes_ok, plan_calls = True, None
        for mode in ddp_mod.MODES:
            sync.set_mode(mode)
            fake_kernels.CUS_CALLS.clear()
            launched_inside = []
            orig_finish = sync.finish
            def spy_finish(orig=orig_finish, seen=launched_inside):
                seen.append(sync.bytes_on_wire)
                return orig()
            diff.backbone.grad_sync_finish = spy_finish
            w0 = sync.bytes_on_wire
            got = _grads(diff, golden, seed=rank)
            diff.backbone.grad_sync_finish = orig_finish
            # world 2: one add per element, bit-identical whatever the message boundaries; world 8: a ring's summation ORDER depends on where an element sits in
            # its message, and the serialized schedule coalesces ranges - equal to bf16 summation noise, and identical on every rank (checked below)
            modes_ok = modes_ok and all(torch.equal(got[k], synced[k]) if world == 2 else
                                        float((got[k] - synced[k]).abs().max()) <= rtol * float(synced[k].abs().max()) + 1e-6 for k in synced)
            flat_m = torch.cat([got[k].flatten() for k in sorted(got)])
            ref_m = flat_m.clone()
            dist.broadcast(ref_m, src=0)
            modes_ok = modes_ok and torch.equal(flat_m, ref_m)
            if mode == "overlap_planned":
                plan_calls = list(fake_kernels.CUS_CALLS)
            if mode == "serialized":
                modes_ok = modes_ok and launched_inside == [w0] and sync.bytes_on_wire > w0     # nothing on the wire before the end of the backward
            else:
                modes_ok = modes_ok and (min_bucket > 1 or launched_inside[0] > w0)                # small buckets leave from inside the backward
        sync.requested_mode = "auto"
        import time
        def slow_unless_serialized():
            if sync.mode != "serialized":
                time.sleep(0.6 if rank == 1 else 0.0)      # only ONE rank is slow: the MAX over ranks must decide
            _grads(diff, golden, seed=rank)
        table = sync.autotune(slow_unless_serialized, steps=1, settle=0)
        tables = [None] * world
        dist.all_gather_object(tables, (table, sync.mode))
        auto_ok = sync.mode == "serialized" and all(t == tables[0] for t in tables) and set(table) == set(ddp_mod.MODES)
        auto_ok = auto_ok and sync.autotune_report["decision"] == "serialized" and sync.autotune_report["gain_of_fastest_over_overlap"] > 0.03
        # hysteresis (round 5): a 1 % "win" of another schedule is noise - "overlap" stays, the report says which one was fastest and by how much
        sync.requested_mode = "auto"
        def near_tie():
            time.sleep({"overlap": 0.500, "overlap_planned": 0.490, "serialized": 0.520}[sync.mode])    # 2 % = 10 ms: above a loaded host's sleep jitter, below min_gain
        sync.autotune(near_tie, steps=2, settle=0)
        rep = sync.autotune_report
        hyst_ok = sync.mode == "overlap" and rep["decision"] == "overlap" and rep["fastest_measured"] == "overlap_planned" and 0.0 < rep["gain_of_fastest_over_overlap"] < 0.03
        # a rank that cannot switch leaves EVERY rank in "overlap" (divergent schedules = divergent collective sequences = a hang); the step still works
        orig_set = sync.set_mode
        def failing(mode, orig=orig_set):
            if rank == 1 and mode == "serialized":
                raise RuntimeError("this rank cannot run that schedule")
            return orig(mode)
        sync.set_mode = failing
        switched = sync._set_mode_everywhere("serialized")
        fail_ok = switched is False and sync.mode == "overlap"
        got = _grads(diff, golden, seed=rank)
        fail_ok = fail_ok and all(torch.equal(got[k], synced[k]) for k in synced)
        sync.autotune(near_tie, steps=1, settle=0)
        fail_ok = fail_ok and sync.autotune_report["modes_that_could_not_be_set"] == ["serialized"] and sync.mode in ("overlap", "overlap_planned")
        sync.set_mode = orig_set
        modes_now = [None] * world
        dist.all_gather_object(modes_now, sync.mode)
        fail_ok = fail_ok and len(set(modes_now)) == 1
        # bounded in wall time: a selection that would take longer than the budget is skipped, with the reason on record
        sync._set_mode_everywhere("overlap")
        sync.autotune(lambda: time.sleep(0.05), steps=4, settle=1, budget_s=0.2)
        bound_ok = sync.mode == "overlap" and "skipped" in sync.autotune_report
        policy_ok = hyst_ok and fail_ok and bound_ok
        q.put((rank, ok, same, local_ok, worst, sync.bytes_on_wire, acc_ok, acc_same, acc_worst, again_ok, modes_ok, plan_calls, auto_ok and policy_ok, sync.reserved_cus,
               (hyst_ok, fail_ok, bound_ok, rep)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,min_bucket", [(2, 1), (2, 1 << 30), (8, 1)], ids=["world2_small_buckets", "world2_one_bucket", "world8_small_buckets"])
def test_bucketed_allreduce(world, min_bucket):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, min_bucket, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    assert sorted(r[0] for r in res) == list(range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, same, local_ok, worst, nbytes, acc_ok, acc_same, acc_worst, again_ok, modes_ok, plan_calls, auto_ok, reserved, policy in res:
        assert all(policy[:3]), (rank, policy)    # hysteresis, a failing rank drags everybody back to "overlap", wall-time bound
        assert modes_ok, rank                 # overlap / overlap_planned / serialized: bit-identical synchronised gradients
        assert plan_calls == [256 - reserved, 0] and reserved == 32, (rank, plan_calls)   # plan on at the first bucket, off at the end of the backward
        assert auto_ok, rank                  # autotune: same table and same choice on every rank, decided by the slowest rank
        assert ok, (rank, worst)
        assert same, rank
        assert local_ok, rank
        assert nbytes > 0
        assert acc_ok, (rank, acc_worst)      # accumulate-then-sync reduces the accumulated gradients
        assert acc_same, rank                 # ... identically on every rank
        assert again_ok, rank


def test_rccl_channel_env_defaults_and_overrides():
    """NCCL_MAX_NCHANNELS bounds the CUs RCCL's channel kernels hold; bench.py sets it before the communicator exists (and in its children's environment)."""
    from unidisc_amd.ddp import DEFAULT_RCCL_CHANNELS, rccl_channel_env

    env = {}
    assert rccl_channel_env(env) == DEFAULT_RCCL_CHANNELS == 32 and env["NCCL_MAX_NCHANNELS"] == "32"
    env = {"NCCL_MAX_NCHANNELS": "8"}                       # the caller's choice wins
    assert rccl_channel_env(env) == 8 and env["NCCL_MAX_NCHANNELS"] == "8"
    env = {"UDM_RCCL_CHANNELS": "16", "NCCL_MIN_NCHANNELS": "64"}
    assert rccl_channel_env(env) == 16 and env["NCCL_MIN_NCHANNELS"] == "16"   # a MIN above the cap would defeat it
    env = {"UDM_RCCL_CHANNELS": "0"}                        # 0: leave RCCL alone
    assert rccl_channel_env(env) == 0 and "NCCL_MAX_NCHANNELS" not in env
