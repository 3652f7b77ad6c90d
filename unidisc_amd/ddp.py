"""Data-parallel gradient synchronisation for ``unidisc_amd.DIT``: bucketed bf16 all-reduce over RCCL/xGMI,
overlapped with the rest of the backward.

Reference behaviour replaced (main.py:641-656): torch DDP with ``comm_hook=DDPCommunicationHookType.BF16`` —
per bucket ``c = bucket_fp32.to(bf16).div_(world)``, async ``all_reduce(c, SUM)``, ``bucket_fp32.copy_(c)`` —
with ``gradient_as_bucket_view=True``.  Here the engine's backward writes every gradient into ONE flat fp32
buffer in completion order (head, blocks n-1..0, embeddings) and reports finished ranges; each report is a
bucket (~100 MB bf16 per DiT block at 1.4 B: large messages, because xGMI rings are per-link bound) that is
compressed, all-reduced and decompressed in place on a dedicated comm stream while the compute stream keeps
running the remaining layers.  The compute stream waits for the comm stream once, at the end of backward.

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm.  On CPU tensors (gloo, used by the
world_size-2 tests of this file) the same code path runs with torch casts instead of the HIP cast kernels.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import kernels as K


class BucketedGradSync:
    def __init__(self, module, process_group=None, min_bucket_elems: int = 32 * 1024 * 1024, wire_dtype=torch.bfloat16):
        if not dist.is_initialized():
            raise RuntimeError("BucketedGradSync needs an initialised torch.distributed process group")
        self.module, self.pg = module, process_group
        self.world = dist.get_world_size(process_group)
        self.min_bucket = int(min_bucket_elems)
        self.wire_dtype = wire_dtype
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self._pending: Optional[Tuple[torch.Tensor, int, int]] = None
        self._work: List = []
        self._bufs = {}
        self.enabled = True  # set False for gradient-accumulation micro-steps (DDP no_sync, model.py:1412)
        self.bytes_on_wire = 0
        module.grad_ready_callback = self._on_ready
        module.grad_sync_finish = self.finish

    # ---- called from inside backward, on the compute stream's thread
    def _on_ready(self, flat: torch.Tensor, lo: int, hi: int):
        if not self.enabled or self.world == 1:
            return
        if self._pending is not None and self._pending[0] is flat and self._pending[2] == lo:
            lo = self._pending[1]
        elif self._pending is not None:
            self._launch(*self._pending)
        self._pending = (flat, lo, hi)
        if hi - lo >= self.min_bucket:
            self._launch(flat, lo, hi)
            self._pending = None

    def _wire_buffer(self, n: int, device) -> torch.Tensor:
        key = (n, device)
        buf = self._bufs.get(key)
        if buf is None:
            buf = torch.empty(n, dtype=self.wire_dtype, device=device)
            self._bufs[key] = buf
        return buf

    def _launch(self, flat: torch.Tensor, lo: int, hi: int):
        seg = flat[lo:hi]
        n = hi - lo
        self.bytes_on_wire += n * 2
        if flat.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            wire = torch.empty(n, dtype=self.wire_dtype, device=flat.device)  # per-bucket: several buckets are in flight at once
            wire.record_stream(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                K.cast_f32_bf16(seg, wire, scale=1.0 / self.world)   # bf16 first, then divide in bf16 (reference hook order)
                dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.pg)
                K.cast_bf16_f32(wire, seg, scale=1.0)
        else:  # gloo / CPU tensors (tests)
            wire = (seg.to(self.wire_dtype).float() * (1.0 / self.world)).to(self.wire_dtype)
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.pg)
            seg.copy_(wire.float())

    def finish(self):
        """End of backward: flush the last partial bucket and make the compute stream wait for all reductions."""
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)


def wrap(module, **kw) -> BucketedGradSync:
    """Attach gradient synchronisation to a ``unidisc_amd.DIT`` (idempotent per module)."""
    sync = getattr(module, "_grad_sync", None)
    if sync is None:
        sync = BucketedGradSync(module, **kw)
        module._grad_sync = sync
    return sync


def broadcast_parameters(module, src: int = 0, process_group=None):
    """Make every rank start from rank `src`'s weights (what torch DDP does at construction)."""
    for p in module.parameters():
        dist.broadcast(p.data, src=src, group=process_group)
