"""Sharded data parallelism for ``unidisc_amd.DIT``: optimizer state and reduced gradients partitioned over the ranks.

Reference behaviour replaced (main.py:593-639): FSDP with ``ShardingStrategy.SHARD_GRAD_OP`` (ZeRO-2) - parameters replicated for the forward and
backward, gradients reduce-scattered, optimizer state sharded - which the reference needs to fit 1.4 B on 48 GB parts.  Here the unit of ownership is
the engine's gradient BUCKET (a contiguous range of its flat gradient buffer: the head, each DiT block, the embeddings - the ranges the backward reports as
they become final), not an even split of a flat parameter:

  * ``ShardedGradSync`` (a ``BucketedGradSync``): every bucket has ONE owner rank (greedy least-loaded assignment in order of first appearance, identical
    on all ranks).  A finished bucket is bf16-compressed and REDUCED TO ITS OWNER on the comm stream while the backward continues (``dist.reduce``:
    half the wire traffic of an all-reduce); only the owner decompresses it.  Non-owned gradient ranges hold nothing useful afterwards.
  * ``ShardedAdamW`` (a ``FusedAdamW``): Adam moments exist only for the owned parameters (1 / world of 2 x 5.6 GB at 1.4 B), the global clipping norm
    is the all-reduced sum of the owners' partial sums of squares, the owner runs the fused AdamW kernels, then every bucket's updated fp32 masters are
    broadcast from their owner (the other half of the all-reduce's traffic) and the bf16 weight shadows are rebuilt by the next forward.

Every rank ends a step with bit-identical parameters, equal to what the replicated path (bucketed all-reduce + FusedAdamW on every rank) produces from the
same reduced gradients.  Gradient accumulation (``sync.enabled = False`` micro-steps, model.py:1412,1505-1506): local passes keep their gradients, the
closing pass reduces the accumulated ``p.grad`` to the same owners bucket by bucket at the end of its backward.  The parameter EMA (models/ema.py) is
sharded like the moments: the owner updates its slice inside the fused AdamW kernel; ``ema_store_and_copy`` broadcasts the EMA weights from their owners.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import kernels as K
from .ddp import BucketedGradSync
from .optim import FusedAdamW


class ShardedGradSync(BucketedGradSync):
    def __init__(self, module, **kw):
        kw.setdefault("mode", "overlap")    # reductions to owners always leave from inside the backward (ddp.py's comm policy covers the replicated path)
        super().__init__(module, **kw)
        self.rank = dist.get_rank(self.pg)
        self.owners: Dict[int, int] = {}                     # bucket start (element offset in the flat buffer) -> owner rank
        self.ranges: List[Tuple[int, int, int]] = []         # (lo, hi, owner) of the last backward, in completion order
        self._load = [0] * self.world
        self.ranges_open = False

    def _owner(self, lo: int, n: int) -> int:
        o = self.owners.get(lo)
        if o is None:   # same decision on every rank: buckets appear in the same order with the same sizes
            o = min(range(self.world), key=lambda r: (self._load[r], r))
            self.owners[lo] = o
            self._load[o] += n
        return o

    # The bucket walk runs in EVERY backward - also in local (enabled = False) micro-steps and in the pass that closes an accumulation - so that the
    # bucket ranges and their owners are the same whichever way the gradients get reduced; only `live` passes launch reductions from inside the backward.
    def _on_ready(self, flat, lo, hi):
        if not self.active:
            return
        if not self.ranges_open:
            self.ranges, self.ranges_open = [], True
        if self._pending is not None and self._pending[0] is flat and self._pending[2] == lo:
            lo = self._pending[1]
        elif self._pending is not None:
            self._close(*self._pending)
        self._pending = (flat, lo, hi)
        if hi - lo >= self.min_bucket:
            self._close(flat, lo, hi)
            self._pending = None

    def _close(self, flat, lo, hi):
        owner = self._owner(lo, hi - lo)
        self.ranges.append((lo, hi, owner))
        if self.enabled and not self._unsynced_passes:
            self._reduce_to(flat[lo:hi], ("flat", lo), owner)

    def finish(self):
        if not self.active:
            return
        if self._pending is not None:
            self._close(*self._pending)
            self._pending = None
        self.ranges_open = False
        if not self.enabled:      # gradient accumulation micro-step (DDP no_sync, model.py:1412,1505-1506): gradients stay local
            self._unsynced_passes += 1
            return
        if self._unsynced_passes:   # the accumulated p.grad exist once autograd has added this pass's gradients
            torch.autograd.Variable._execution_engine.queue_callback(self.reduce_accumulated)
            return
        self._join()

    def allreduce_accumulated(self):
        self.reduce_accumulated()

    def reduce_accumulated(self):
        """Reduce what is in ``p.grad`` now (the sum of this rank's micro-step gradients) to the bucket owners, bucket by bucket with the ranges and
        owners of the in-backward path."""
        pr = self.module._grad_ranges
        by_id = {id(p): p for p in self.module._ordered_params()}
        for lo, hi, owner in self.ranges:
            bucket = [by_id[pid] for pid, (o, _e) in pr.items() if lo <= o < hi and by_id[pid].grad is not None]
            bucket.sort(key=lambda q: pr[id(q)][0])
            if bucket:
                self._reduce_params_to(bucket, owner, ("acc", lo))
        self._unsynced_passes = 0
        self._join()

    def _reduce_params_to(self, bucket, owner, key):
        grads = [p.grad for p in bucket]
        first = grads[0]
        if all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):   # views of one flat buffer laid out back to back: reduce in place
            end, same = first.data_ptr(), True
            for g in grads:
                gap = g.data_ptr() - end
                same = same and 0 <= gap < 64 * 4 and g.untyped_storage().data_ptr() == first.untyped_storage().data_ptr()
                end = g.data_ptr() + g.numel() * 4
            if same:
                n = (end - first.data_ptr()) // 4
                self._reduce_to(torch.as_strided(first, (n,), (1,), first.storage_offset()), key, owner)
                return
        cat = torch.cat([g.reshape(-1).float() for g in grads])
        self._reduce_to(cat, key, owner)
        if self.rank == owner:
            if cat.is_cuda:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            off = 0
            for g in grads:
                g.copy_(cat[off:off + g.numel()].view_as(g))
                off += g.numel()

    def _reduce_to(self, seg: torch.Tensor, key, owner: int):
        n = seg.numel()
        self.bytes_on_wire += n * 2
        dst = dist.get_global_rank(self.pg, owner) if self.pg is not None else owner
        if seg.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=seg.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            wire = self._wire_buffer(key, n, seg.device)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                K.cast_f32_bf16(seg, wire, scale=1.0 / self.world)
                dist.reduce(wire, dst=dst, op=dist.ReduceOp.SUM, group=self.pg)
                if self.rank == owner:
                    K.cast_bf16_f32(wire, seg, scale=1.0)
            seg.record_stream(self.comm_stream)
        else:  # gloo / CPU tensors (tests)
            wire = (seg.to(self.wire_dtype).float() * (1.0 / self.world)).to(self.wire_dtype)
            dist.reduce(wire, dst=dst, op=dist.ReduceOp.SUM, group=self.pg)
            if self.rank == owner:
                seg.copy_(wire.float())

    def owner_of_params(self) -> Dict[int, int]:
        """id(parameter) -> owner rank, from the bucket ranges of the last backward and the engine's parameter ranges in the flat gradient buffer"""
        pr = getattr(self.module, "_grad_ranges", None)
        if not pr or not self.ranges:
            raise RuntimeError("ShardedGradSync: no backward has run yet")
        out = {}
        for pid, (o, _end) in pr.items():
            for lo, hi, owner in self.ranges:
                if lo <= o < hi:
                    out[pid] = owner
                    break
            else:
                raise RuntimeError("ShardedGradSync: a parameter's gradient range lies in no reduced bucket")
        return out


def wrap_sharded(module, **kw) -> ShardedGradSync:
    sync = getattr(module, "_grad_sync", None)
    if sync is None:
        sync = ShardedGradSync(module, **kw)
        module._grad_sync = sync
    if not isinstance(sync, ShardedGradSync):
        raise RuntimeError("wrap_sharded: the module already has a replicated gradient sync attached")
    return sync


class ShardedAdamW(FusedAdamW):
    """``sync = wrap_sharded(backbone); opt = ShardedAdamW(backbone, sync, lr=...)``; then ``loss.backward(); opt.step(); opt.zero_grad()`` as usual."""

    def __init__(self, backbone, sync: ShardedGradSync, **kw):
        params = kw.pop("params", None)
        ema_decay = kw.pop("ema_decay", None)
        super().__init__(backbone, params=[], **kw)   # (no moments yet: they are allocated for OWNED parameters only, once ownership is known)
        self.params = [p for p in (params if params is not None else backbone.parameters()) if p.requires_grad]
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("ShardedAdamW: parameters must be contiguous fp32 masters")
        self.sync = sync
        self._owner: Optional[Dict[int, int]] = None
        self._groups: Optional[List[Tuple[int, List[torch.nn.Parameter]]]] = None   # (owner, parameters of one bucket) in bucket order
        self._bcast_bufs: dict = {}   # {"ring": [staging buffers]}
        self.ema_decay = float(ema_decay) if ema_decay else None
        self.ema = {} if self.ema_decay else None     # EMA shadows of the OWNED parameters (models/ema.py; stepped with the optimizer, model.py:1541-1545)
        if self.ema is not None and not sync.active:
            self.ema = {id(p): p.detach().clone() for p in self.params}

    def owned(self, p) -> bool:
        return self._owner is not None and self._owner[id(p)] == self.sync.rank

    def _resolve_ownership(self):
        sync = self.sync
        self._owner = sync.owner_of_params()
        missing = [p for p in self.params if id(p) not in self._owner]
        if missing:
            raise RuntimeError("ShardedAdamW: parameters outside the engine's flat gradient buffer")
        pr = sync.module._grad_ranges
        groups = []
        for lo, hi, owner in sync.ranges:
            ps = sorted([p for p in self.params if lo <= pr[id(p)][0] < hi], key=lambda q: pr[id(q)][0])
            if ps:
                groups.append((owner, ps))
        self._groups = groups
        for p in self.params:   # (a loaded shard keeps its moments / EMA)
            if self._owner[id(p)] == sync.rank:
                if id(p) not in self.state:
                    self.state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
                if self.ema is not None and id(p) not in self.ema:
                    self.ema[id(p)] = p.detach().clone()

    _BCAST_RING = 3

    def _broadcast_from_owners(self, tensors_of):
        """One broadcast per bucket (a 1.4 B model has ~600 parameters but ~26 buckets): the owner packs `tensors_of(p)` of the bucket's parameters into a
        flat staging buffer, the others unpack into their parameters.  Staging is a RING of three buffers sized to the largest bucket (~0.6 GB at 1.4 B,
        not a second fp32 copy of the model): a slot is reused only after the broadcast that last used it has completed and been unpacked."""
        sync = self.sync
        sizes = [sum(p.numel() for p in ps) for _, ps in self._groups]
        if not sizes:
            return
        dev = self._groups[0][1][0].device
        ring = self._bcast_bufs.get("ring")
        if ring is None or ring[0].numel() < max(sizes) or ring[0].device != dev:
            ring = [torch.empty(max(sizes), dtype=torch.float32, device=dev) for _ in range(self._BCAST_RING)]
            self._bcast_bufs.clear()
            self._bcast_bufs["ring"] = ring
        pending = [None] * len(ring)

        def finish(slot):
            job = pending[slot]
            if job is None:
                return
            w, owner, ps, buf = job
            w.wait()
            if owner != sync.rank:
                torch._foreach_copy_([p.data.view(-1) for p in ps], list(buf.split([p.numel() for p in ps])))
            pending[slot] = None

        for gi, (owner, ps) in enumerate(self._groups):
            slot = gi % len(ring)
            finish(slot)
            buf = ring[slot][: sizes[gi]]
            if owner == sync.rank:
                torch.cat([tensors_of(p).reshape(-1) for p in ps], out=buf)
            src = dist.get_global_rank(sync.pg, owner) if sync.pg is not None else owner
            pending[slot] = (dist.broadcast(buf, src=src, group=sync.pg, async_op=True), owner, ps, buf)
        for slot in range(len(ring)):
            finish(slot)

    @torch.no_grad()
    def step(self):
        sync = self.sync
        if not sync.active:   # a single rank owns everything
            if not self.state:
                self.state = {id(p): (torch.zeros_like(p), torch.zeros_like(p)) for p in self.params}
            return super().step()
        todo = [p for p in self.params if p.grad is not None]
        if not todo:   # every rank must take part in the collectives below: an early return here would hang the others
            raise RuntimeError("ShardedAdamW.step(): no gradients on this rank (every rank has to run a backward before step)")
        if self._owner is None:
            self._resolve_ownership()
        mine = [p for p in todo if self._owner[id(p)] == sync.rank]
        self.step_count += 1
        dev = todo[0].device
        gsq = None
        if self.max_grad_norm is not None:   # global norm of the REDUCED gradients: owners' partial sums of squares, summed over the ranks
            gsq = torch.zeros(1, dtype=torch.float32, device=dev)
            for p in mine:
                part = torch.zeros(1, dtype=torch.float32, device=dev)
                K.sumsq(p.grad.reshape(-1), part)
                gsq += part
            dist.all_reduce(gsq, op=dist.ReduceOp.SUM, group=sync.pg)
            self.grad_norm = gsq.sqrt()
        b1, b2 = self.betas
        ed = self._ema_decay_now() if self.ema is not None else 0.0
        items = []
        for p in mine:
            m, v = self.state[id(p)]
            e = self.ema[id(p)] if self.ema is not None else None
            if p.is_cuda and p.is_contiguous() and p.grad.is_contiguous() and all(t.data_ptr() % 16 == 0 for t in (p, p.grad, m, v) + ((e,) if e is not None else ())):
                items.append((p, p.grad, m, v, e))      # every owned tensor in ONE launch below (optim.hip: udm_adamw_step_multi)
            else:
                K.adamw_step(p, p.grad, m, v, self.lr, b1, b2, self.eps, self.weight_decay, self.step_count, gsq, self.max_grad_norm, ema=e, ema_decay=ed)
        if items:
            key = tuple(t.data_ptr() if t is not None else 0 for it in items for t in it)
            if getattr(self, "_multi_key", None) != key:
                self._multi_jobs, self._multi_key = K.adamw_jobs(items, items[0][0].device), key
            K.adamw_step_multi(self._multi_jobs, self.lr, b1, b2, self.eps, self.weight_decay, self.step_count, gsq, self.max_grad_norm, ema_decay=ed)
        self._broadcast_from_owners(lambda p: p.data)   # updated masters: every bucket from its owner
        if hasattr(self.backbone, "invalidate_shadows"):   # the bf16 shadows of every rank follow on the next forward
            self.backbone.invalidate_shadows()
            self.backbone.recast_every_forward = True

    @torch.no_grad()
    def ema_store_and_copy(self):
        """`ema.store(params); ema.copy_to(params)`: every rank's parameters <- the EMA weights, broadcast from their owners; originals kept for `ema_restore`."""
        if self.ema is None:
            raise RuntimeError("ShardedAdamW: no EMA (ema_decay not set)")
        if not self.sync.active:
            return super().ema_store_and_copy()
        if self._owner is None:
            raise RuntimeError("ShardedAdamW: ema_store_and_copy before the first step")
        self._ema_backup = [p.detach().clone() for p in self.params]
        self._broadcast_from_owners(lambda p: self.ema[id(p)])
        for p in self.params:   # (the owner has packed its EMA slice but not yet taken it itself)
            if self._owner[id(p)] == self.sync.rank:
                p.copy_(self.ema[id(p)])
        if hasattr(self.backbone, "refresh_weight_shadows"):
            self.backbone.refresh_weight_shadows(force=True)

    def state_dict(self):
        own = [(i, p) for i, p in enumerate(self.params) if id(p) in self.state]
        sd = dict(step=self.step_count, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, max_grad_norm=self.max_grad_norm,
                  rank=self.sync.rank, world=self.sync.world,
                  owned=[i for i, _ in own],
                  exp_avg=[self.state[id(p)][0] for _, p in own],
                  exp_avg_sq=[self.state[id(p)][1] for _, p in own],
                  ema_decay=self.ema_decay, ema_num_updates=self.ema_num_updates,
                  ema=[self.ema[id(p)] for _, p in own] if self.ema is not None and all(id(p) in self.ema for _, p in own) else None,
                  dropout_fwd_count=int(getattr(self.backbone, "_fwd_count", 0)))   # position of the engine's dropout stream (a resume must not replay masks)
        return sd

    def load_state_dict(self, sd):
        if int(sd.get("world", self.sync.world)) != self.sync.world or int(sd.get("rank", self.sync.rank)) != self.sync.rank:
            raise ValueError("ShardedAdamW: this shard was written by another rank / world size")
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        self.state = {}
        for i, m, v in zip(sd["owned"], sd["exp_avg"], sd["exp_avg_sq"]):
            p = self.params[i]
            self.state[id(p)] = (m.to(p.device).clone(), v.to(p.device).clone())
        if self.ema is not None and sd.get("ema") is not None:
            self.ema_num_updates = sd.get("ema_num_updates", self.ema_num_updates)
            for i, e in zip(sd["owned"], sd["ema"]):
                self.ema[id(self.params[i])] = e.to(self.params[i].device).clone()
        if "dropout_fwd_count" in sd and hasattr(self.backbone, "_fwd_count"):
            self.backbone._fwd_count = int(sd["dropout_fwd_count"])
        if hasattr(self.backbone, "invalidate_shadows"):
            self.backbone.invalidate_shadows()
            self.backbone.recast_every_forward = True
