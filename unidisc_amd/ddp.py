"""Data-parallel gradient synchronisation for ``unidisc_amd.DIT``: bucketed bf16 all-reduce over RCCL/xGMI,
overlapped with the rest of the backward.

Reference behaviour replaced (main.py:641-656): torch DDP with ``comm_hook=DDPCommunicationHookType.BF16`` —
per bucket ``c = bucket_fp32.to(bf16).div_(world)``, async ``all_reduce(c, SUM)``, ``bucket_fp32.copy_(c)`` —
with ``gradient_as_bucket_view=True``.  Here the engine's backward writes every gradient into ONE flat fp32
buffer in completion order (head, blocks n-1..0, embeddings) and reports finished ranges; each report is a
bucket (~100 MB bf16 per DiT block at 1.4 B: large messages, because xGMI rings are per-link bound) that is
compressed, all-reduced and decompressed in place on a dedicated comm stream while the compute stream keeps
running the remaining layers.  The compute stream waits for the comm stream once, at the end of backward.

Gradient accumulation (DDP ``no_sync``, model.py:1412): backward passes run with ``sync.enabled = False`` keep
their gradients local (autograd sums them into ``p.grad``).  The first backward with ``enabled = True`` after such
passes must reduce the ACCUMULATED gradients, which only exist once autograd has added this pass's gradients to
``p.grad`` — so that pass does not reduce inside the backward; it queues an end-of-backward callback that
compresses, all-reduces and decompresses ``p.grad`` bucket by bucket (``allreduce_accumulated``).  Without
accumulation every pass takes the overlapped in-backward path.

Communication policy (``mode`` / ``UDM_DDP_MODE``; round 4).  A collective's channel kernels HOLD CUs while they run, and every
backward GEMM of the 1.4 B model is exactly 256 one-workgroup tiles - with k >= 1 CUs held it runs two rounds (measured with a spinning
kernel: +40 % per step while CUs are held).  Which schedule is fastest therefore depends on the fabric, and is decided by
MEASUREMENT at start-up rather than assumed:
  * ``overlap``          buckets all-reduced on the comm stream while the backward continues (the reference DDP's schedule);
  * ``overlap_planned``  the same, and from the first bucket of a backward until its end the GEMMs plan for ``256 - reserved_cus`` CUs
                         (`kernels.gemm_set_cus`): single-round grids are cut to what fits beside the collective, leftovers split in K;
  * ``serialized``       nothing is launched inside the backward; at its end the finished ranges go out as a few large all-reduces (coalesced
                         up to ``serial_bucket_elems``) and the compute stream waits: no CU contention, communication fully exposed;
  * ``auto`` (default)   ``autotune(step_fn)`` times each of the three for a few steps on all ranks, takes all_reduce(MAX) of the times and keeps
                         ``overlap`` unless another schedule is at least 3 % faster for the slowest rank (a noise-level "win" must not select a schedule
                         that costs +20 % when nothing is held); until it has run, ``auto`` behaves as ``overlap``.  The selection is bounded in wall time
                         and every mode switch is agreed by all ranks (a rank that cannot switch leaves EVERY rank in ``overlap``: divergent schedules
                         would deadlock the collectives).
(A copy-engine exchange - peer-to-peer copies instead of a collective - was rehearsed in round 4 with two ranks on one GPU; it cannot be validated on this
pool and lives under experiments/ddp_copy_engine/, outside the product.)
RCCL's CU footprint is bounded by ``rccl_channel_env()`` (NCCL_MAX_NCHANNELS, default 32 = ``reserved_cus``; must be in the environment
before the communicator is created).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm.  On CPU tensors (gloo, used by the
world_size-2 tests of this file) the same code path runs with torch casts instead of the HIP cast kernels.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import kernels as K


MODES = ("overlap", "overlap_planned", "serialized")      # the schedules `auto` times
DEFAULT_RCCL_CHANNELS = 32     # one channel = one workgroup = one CU; 32 is what `overlap_planned` reserves (a whole multiple of the 8 XCDs x 4 shader engines)


def rccl_channel_env(env=None, channels=None):
    """Bound RCCL's CU footprint: NCCL_MAX_NCHANNELS (and a MIN that does not exceed it) in `env` (default os.environ) unless the caller already
    set them.  Must run BEFORE `init_process_group` / the first collective - RCCL reads them when it creates the communicator.
    `UDM_RCCL_CHANNELS` overrides the default of 32; 0 leaves RCCL's own choice alone.  Returns the channel cap in force (0 = RCCL's default)."""
    env = os.environ if env is None else env
    if channels is None:
        channels = int(env.get("UDM_RCCL_CHANNELS", DEFAULT_RCCL_CHANNELS) or 0)
    if channels > 0:
        env.setdefault("NCCL_MAX_NCHANNELS", str(channels))
        if "NCCL_MIN_NCHANNELS" in env and int(env["NCCL_MIN_NCHANNELS"]) > int(env["NCCL_MAX_NCHANNELS"]):
            env["NCCL_MIN_NCHANNELS"] = env["NCCL_MAX_NCHANNELS"]
    return int(env.get("NCCL_MAX_NCHANNELS", "0") or 0)


class BucketedGradSync:
    def __init__(self, module, process_group=None, min_bucket_elems: int = 32 * 1024 * 1024, wire_dtype=torch.bfloat16, force_single_rank=None,
                 mode=None, reserved_cus=None, serial_bucket_elems: int = 256 * 1024 * 1024):
        if not dist.is_initialized():
            raise RuntimeError("BucketedGradSync needs an initialised torch.distributed process group")
        self.module, self.pg = module, process_group
        self.world = dist.get_world_size(process_group)
        self.min_bucket = int(min_bucket_elems)
        self.wire_dtype = wire_dtype
        # world_size 1 has nothing to reduce and is skipped, unless forced (tests / the 1-GPU RCCL rehearsal run the whole path on one rank)
        self.force_single_rank = bool(int(os.environ.get("UDM_DDP_FORCE", "0"))) if force_single_rank is None else bool(force_single_rank)
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self._pending: Optional[Tuple[torch.Tensor, int, int]] = None
        self._bufs = {}
        self.enabled = True            # set False for gradient-accumulation micro-steps (DDP no_sync, model.py:1412)
        self._unsynced_passes = 0      # backward passes since the last reduction whose gradients stayed local
        self.bytes_on_wire = 0
        self.measure_exposed = False   # bench: time the compute stream spends waiting for the comm stream at the end of backward
        self._exposed: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []
        mode = os.environ.get("UDM_DDP_MODE", "auto") if mode is None else mode
        if mode != "auto" and mode not in MODES:
            raise ValueError(f"BucketedGradSync: unknown mode {mode!r} (one of {MODES + ('auto',)})")
        self.requested_mode = mode
        self.mode = "overlap" if mode == "auto" else mode     # what runs now; `autotune` replaces an "auto" request by the measured winner
        self.mode_timings_ms = None                            # {mode: ms per step, max over ranks} once autotune has run
        self.autotune_report = None                            # how the decision was taken (margin, steps, or why the selection was skipped)
        if reserved_cus is None:
            reserved_cus = int(os.environ.get("UDM_DDP_RESERVED_CUS", "0") or 0) or (int(os.environ.get("NCCL_MAX_NCHANNELS", "0") or 0) or DEFAULT_RCCL_CHANNELS)
        self.reserved_cus = (int(reserved_cus) + 31) // 32 * 32     # whole shader-engine multiples: an exact fit only works when the held CUs spread one per engine (DESIGN §5 v)
        self.serial_bucket = int(serial_bucket_elems)
        self._deferred: List[Tuple[torch.Tensor, int, int]] = []  # serialized mode: ranges finished inside the backward, reduced at its end
        self._planned = False                                      # overlap_planned: the GEMM plan is in force for the rest of this backward
        self._in_backward = False
        p0 = next(iter(module.parameters()), None)
        self._agree_device = p0.device if (p0 is not None and p0.is_cuda) else "cpu"    # where the ranks' yes / no and timing tensors live (RCCL wants device tensors)
        module.grad_ready_callback = self._on_ready
        module.grad_sync_finish = self.finish

    @property
    def active(self):
        return self.world > 1 or self.force_single_rank

    # ---- called from inside backward, on the compute stream's thread
    def _on_ready(self, flat: torch.Tensor, lo: int, hi: int):
        if not self.active or not self.enabled or self._unsynced_passes:
            return   # local pass, or accumulated gradients: reduced after autograd has summed them (finish)
        if not self._in_backward:       # first range of a backward: nothing of a previous one may linger (a backward that raised never reached finish())
            self._in_backward = True
            self._pending, self._deferred = None, []
            self._unplan()
        if self.mode == "serialized":   # nothing leaves inside the backward; adjacent ranges coalesce into few large messages
            if self._deferred and self._deferred[-1][0] is flat and self._deferred[-1][2] == lo and hi - self._deferred[-1][1] <= self.serial_bucket:
                self._deferred[-1] = (flat, self._deferred[-1][1], hi)
            else:
                self._deferred.append((flat, lo, hi))
            return
        if self._pending is not None and self._pending[0] is flat and self._pending[2] == lo:
            lo = self._pending[1]
        elif self._pending is not None:
            self._launch(*self._pending)
        self._pending = (flat, lo, hi)
        if hi - lo >= self.min_bucket:
            self._launch(flat, lo, hi)
            self._pending = None

    def _wire_buffer(self, key, n: int, device) -> torch.Tensor:
        """Persistent bf16 wire buffer of one bucket position (the flat layout is the same every step, so a bucket keeps its buffer; several
        buckets are in flight at once, so they cannot share one)."""
        key = (key, n, device)
        buf = self._bufs.get(key)
        if buf is None:
            buf = torch.empty(n, dtype=self.wire_dtype, device=device)
            self._bufs[key] = buf
        return buf

    def _reduce_segment(self, seg: torch.Tensor, key):
        """bf16-compress `seg` (fp32, contiguous), all-reduce, decompress in place.  GPU: on the comm stream, after what the current stream queued."""
        n = seg.numel()
        self.bytes_on_wire += n * 2
        if seg.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=seg.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            wire = self._wire_buffer(key, n, seg.device)   # reuse is stream-ordered: all users run on the comm stream
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                K.cast_f32_bf16(seg, wire, scale=1.0 / self.world)   # bf16 first, then divide in bf16 (reference hook order)
                dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.pg)
                K.cast_bf16_f32(wire, seg, scale=1.0)
            seg.record_stream(self.comm_stream)
        else:  # gloo / CPU tensors (tests)
            wire = (seg.to(self.wire_dtype).float() * (1.0 / self.world)).to(self.wire_dtype)
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.pg)
            seg.copy_(wire.float())

    def _launch(self, flat: torch.Tensor, lo: int, hi: int):
        self._reduce_segment(flat[lo:hi], ("flat", lo))
        if self.mode == "overlap_planned" and not self._planned:
            # from here to the end of this backward a collective may be in flight: the GEMMs launched from now on plan for the CUs it leaves
            K.gemm_set_cus(256 - self.reserved_cus)
            self._planned = True

    def _unplan(self):
        if self._planned:
            K.gemm_set_cus(0)
            self._planned = False

    def _join(self):
        self._in_backward = False
        if self.comm_stream is not None:
            cur = torch.cuda.current_stream()
            if self.measure_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.comm_stream)
                e1.record(cur)
                self._exposed.append((e0, e1))
            else:
                cur.wait_stream(self.comm_stream)

    def finish(self):
        """End of the engine's backward: flush the last partial bucket and make the compute stream wait for all reductions; after
        accumulation micro-steps, reduce the accumulated ``p.grad`` once autograd has finished this pass."""
        if not self.active:
            return
        if not self.enabled:
            self._unsynced_passes += 1
            return
        if self._unsynced_passes:
            torch.autograd.Variable._execution_engine.queue_callback(self.allreduce_accumulated)
            return
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        for job in self._deferred:
            self._launch(*job)
        self._deferred = []
        self._unplan()     # (the next forward's GEMMs start after the join below: nothing overlaps them)
        self._join()

    def allreduce_accumulated(self):
        """Reduce what is in ``p.grad`` now (sum of this rank's micro-step gradients) across ranks, bucket by bucket in backward-completion
        order.  Runs by itself at the end of the first ``enabled`` backward after ``enabled = False`` passes; callable directly as well."""
        params = [p for p in self.module._ordered_params() if p.grad is not None]
        bucket, size = [], 0
        for i, p in enumerate(params):
            bucket.append(p)
            size += p.grad.numel()
            if size >= self.min_bucket or i + 1 == len(params):
                self._reduce_params(bucket)
                bucket, size = [], 0
        self._unsynced_passes = 0
        self._join()

    def _reduce_params(self, bucket):
        grads = [p.grad for p in bucket]
        key = ("acc", id(bucket[0]))
        first = grads[0]
        contiguous = all(g.is_contiguous() and g.dtype == torch.float32 for g in grads)
        if contiguous and len(grads) > 1:   # views of one flat buffer laid out back to back (the engine's layout, 64-element aligned): reduce in place
            end = first.data_ptr()
            for g in grads:
                gap = g.data_ptr() - end
                contiguous = contiguous and 0 <= gap < 64 * 4 and g.untyped_storage().data_ptr() == first.untyped_storage().data_ptr()
                end = g.data_ptr() + g.numel() * 4
            if contiguous:
                n = (end - first.data_ptr()) // 4
                seg = torch.as_strided(first, (n,), (1,), first.storage_offset())
                self._reduce_segment(seg, key)
                return
        if len(grads) == 1 and first.is_contiguous() and first.dtype == torch.float32:
            self._reduce_segment(first.view(-1), key)
            return
        cat = torch.cat([g.reshape(-1).float() for g in grads])
        self._reduce_segment(cat, key)
        if cat.is_cuda and self.comm_stream is not None:   # the copy-back below reads what the comm stream writes
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        off = 0
        for g in grads:
            g.copy_(cat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def set_mode(self, mode: str):
        if mode not in MODES:
            raise ValueError(f"BucketedGradSync: unknown mode {mode!r} (one of {MODES})")
        if self._pending is not None or self._deferred:
            raise RuntimeError("BucketedGradSync.set_mode inside a backward")
        self._unplan()
        if mode != self.mode:
            # the other schedule's persistent wire buffers (serialized: up to 512 MB each) are not kept beside this one's.  They were allocated on the compute
            # stream and are used on the comm stream: the compute stream waits for whatever the comm stream still has queued on them before the caching
            # allocator may hand that memory to a compute-stream kernel (today every caller has joined already; this makes the drop safe by itself)
            if self._bufs and self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._bufs = {}
        self.mode = mode

    def _set_mode_everywhere(self, mode: str) -> bool:
        """Collective: every rank switches to `mode`, or - if ANY rank cannot - every rank ends in "overlap" and False is returned.  Ranks in
        different schedules issue different collective sequences: that is a deadlock, not a slowdown."""
        ok = 1
        try:
            self.set_mode(mode)
        except Exception:
            ok = 0
        if self.world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=self._agree_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            ok = int(flag.item())
        if not ok:
            self._pending, self._deferred = None, []
            self._unplan()
            self._bufs = {}
            self.mode = "overlap"
        return bool(ok)

    def autotune(self, step_fn, steps: int = 4, settle: int = 1, modes=MODES, sync_device=None, min_gain: float = 0.03, budget_s: float = 60.0):
        """Measure `settle + steps` calls of `step_fn()` (one full training step: zero_grad, forward, backward) in every mode, the same sequence on all
        ranks; the time of a mode is the MAX over ranks of its mean step.  "overlap" (the reference DDP's schedule) is kept unless another mode is at
        least `min_gain` faster - the selection must not flip on noise.  Bounded: one probe step is timed first and the selection is skipped (with the
        reason in `autotune_report`) when all modes together would take more than `budget_s`.  Returns the table {mode: ms}.  No-op unless the mode was
        requested as "auto" and gradient synchronisation is active."""
        import time

        if self.requested_mode != "auto" or not self.active:
            return self.mode_timings_ms
        if sync_device is not None:     # (None keeps the device inferred from the parameters in __init__: an RCCL group cannot reduce a CPU tensor)
            self._agree_device = sync_device

        def fence():
            if sync_device is not None:
                torch.cuda.synchronize(sync_device)
            if self.world > 1:
                dist.barrier(group=self.pg)
            if sync_device is not None:
                torch.cuda.synchronize(sync_device)

        def max_over_ranks(values):
            t = torch.tensor(list(values), dtype=torch.float64, device=self._agree_device)
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)     # every rank sees the same numbers -> the same decision
            return t.tolist()

        modes = tuple(modes)
        base = "overlap" if "overlap" in modes else modes[0]
        self._set_mode_everywhere(base)
        fence()
        t0 = time.perf_counter()
        step_fn()
        fence()
        probe_s = max_over_ranks([time.perf_counter() - t0])[0]
        need_s = probe_s * len(modes) * (settle + steps)
        if need_s > budget_s:
            self.autotune_report = dict(decision=base, skipped=f"timing {len(modes)} modes x {settle + steps} steps of {probe_s:.2f} s would take {need_s:.0f} s > {budget_s:.0f} s",
                                        probe_step_s=probe_s)
            return self.mode_timings_ms
        table, failed = {}, []
        for m in modes:
            if not self._set_mode_everywhere(m):
                failed.append(m)
                table[m] = float("inf")
                continue
            for _ in range(settle):
                step_fn()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn()
            fence()
            table[m] = 1e3 * (time.perf_counter() - t0) / max(steps, 1)
        ms = max_over_ranks([table[m] for m in modes])
        ms = [float("inf") if modes[i] in failed else ms[i] for i in range(len(modes))]
        ib = modes.index(base)
        best = min(range(len(modes)), key=lambda i: (ms[i], i))
        if best != ib and not (ms[best] <= (1.0 - min_gain) * ms[ib]):
            best = ib            # a win inside the noise is no win
        self.mode_timings_ms = {m: ms[i] for i, m in enumerate(modes)}
        fastest = min(range(len(modes)), key=lambda i: (ms[i], i))
        self.autotune_report = dict(decision=modes[best], fastest_measured=modes[fastest], gain_of_fastest_over_overlap=1.0 - ms[fastest] / ms[ib] if ms[ib] > 0 else 0.0,
                                    min_gain=min_gain, steps_per_mode=steps, settle_steps=settle, probe_step_s=probe_s, modes_that_could_not_be_set=failed)
        if not self._set_mode_everywhere(modes[best]):
            self.autotune_report["decision"] = "overlap"
        return self.mode_timings_ms

    def exposed_ms(self, reset=True) -> float:
        """Sum over recorded backward passes of the time the compute stream waited for the comm stream (needs ``measure_exposed`` and a device sync)."""
        total = sum(a.elapsed_time(b) for a, b in self._exposed)
        if reset:
            self._exposed = []
        return total


def wrap(module, **kw) -> BucketedGradSync:
    """Attach gradient synchronisation to a ``unidisc_amd.DIT`` (idempotent per module)."""
    sync = getattr(module, "_grad_sync", None)
    if sync is None:
        sync = BucketedGradSync(module, **kw)
        module._grad_sync = sync
    return sync


def broadcast_parameters(module, src: int = 0, process_group=None):
    """Make every rank start from rank `src`'s weights (what torch DDP does at construction).  The bf16 weight shadows of the engine are
    invalidated: a collective writes the parameter storage without bumping tensor versions."""
    with torch.no_grad():
        for p in module.parameters():
            dist.broadcast(p, src=src, group=process_group)
    if hasattr(module, "invalidate_shadows"):
        module.invalidate_shadows()
