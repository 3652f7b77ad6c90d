"""Gradient exchange on the COPY ENGINES: a bf16 all-reduce of one bucket as peer-to-peer copies + one small sum kernel, holding no CUs.

Why: RCCL's channel kernels hold CUs for as long as a collective runs, and every backward GEMM of the 1.4 B model is exactly 256
one-workgroup tiles - with one CU held it runs two rounds.  xGMI is point-to-point (7 links per GPU): a reduce-scatter + all-gather written as
`copy_()` between peer buffers runs on the SDMA engines over all 7 links at once and leaves every CU to the backward; what touches CUs is one cast, one
W-way sum and one decompress kernel per bucket (microseconds each).

    rank r, bucket of n elements, slice = ceil(n / W):
      A  send = bf16(grad / W)                               (cast kernel, this rank's copy stream)
         for every peer p:  peer_recv[p][r, :] <- send[p * slice : (p + 1) * slice]     (W copies: W - 1 of them over xGMI)
         -- host: copy stream drained, control-plane barrier (every rank's pieces have landed) --
      B  red = bf16(sum_q recv[q, :])  (fp32 accumulate)     (one small kernel)
         for every peer p:  peer_gather[p][r * slice : (r + 1) * slice] <- red          (W copies)
         -- host: drain + barrier --
      C  grad <- fp32(gather[:n])                            (decompress kernel)

Peer buffers are ordinary device tensors exported once through torch's CUDA-IPC reductions (`torch.multiprocessing.reductions.reduce_tensor`: on ROCm
hipIpcGetMemHandle / hipIpcOpenMemHandle; this pool needs HSA_ENABLE_IPC_MODE_LEGACY=0); on a multi-GPU node `copy_()` between devices is
hipMemcpyPeerAsync.  The host-side waits run on a HELPER THREAD, one job per bucket, so the thread that launches the backward never blocks; the compute
stream joins the copy stream at the end of the backward.  The control plane (two barriers per bucket) is a gloo group: no GPU kernels.

STATUS: opt-in (`UDM_DDP_MODE=copy_engine`), NOT among the schedules `auto` times.  It has run with two ranks on ONE GPU (scripts/ddp_copy_engine_check.py,
tests/test_gpu_ddp_copy_engine.py: same values as the RCCL path's definition, identical on both ranks) - never on a multi-GPU node: none was
available to this build.  `setup` failing on any rank (no IPC, no peer access) makes every rank fall back to the RCCL schedule.
"""
from __future__ import annotations

import queue
import threading
import time

import torch
import torch.distributed as dist

from . import kernels as K


class CopyEngineExchange:
    def __init__(self, rank: int, world: int, device: torch.device, ctrl_group):
        self.rank, self.world, self.device, self.ctrl = rank, world, device, ctrl_group
        self.capacity = 0          # elements per slice the staging buffers hold
        self.stream = torch.cuda.Stream(device=device)
        self.q: "queue.Queue" = queue.Queue()
        self.error = None
        self.host_wait_s = 0.0     # time the launching thread spent in drain() (exposed communication on the host side)
        self.bytes_copied = 0
        self._thread = threading.Thread(target=self._run, name="udm-copy-engine", daemon=True)
        self._thread.start()

    # ---- staging buffers (grown on demand; every rank sees the same bucket sizes in the same order, so growth is collective by construction)
    def _grow(self, slice_elems: int):
        from torch.multiprocessing.reductions import reduce_tensor

        W, dev = self.world, self.device
        cap = max(slice_elems, 1 << 20)
        self.send = torch.empty(W * cap, dtype=torch.bfloat16, device=dev)
        self.recv = torch.zeros((W, cap), dtype=torch.bfloat16, device=dev)
        self.gather = torch.zeros(W * cap, dtype=torch.bfloat16, device=dev)
        self.acc = torch.empty(cap, dtype=torch.float32, device=dev)
        self.red = torch.empty(cap, dtype=torch.bfloat16, device=dev)
        torch.cuda.synchronize(dev)
        mine = (reduce_tensor(self.recv), reduce_tensor(self.gather))
        handles = [None] * W
        dist.all_gather_object(handles, mine, group=self.ctrl)
        self.peer_recv, self.peer_gather = [], []
        for p, ((f0, a0), (f1, a1)) in enumerate(handles):
            self.peer_recv.append(self.recv if p == self.rank else f0(*a0))
            self.peer_gather.append(self.gather if p == self.rank else f1(*a1))
        self.capacity = cap
        dist.barrier(group=self.ctrl)

    # ---- producer side (the thread that runs the backward)
    def submit(self, seg: torch.Tensor):
        """Queue the all-reduce (mean) of `seg` (fp32, contiguous, on `device`), ordered after what the current stream has queued so far."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.q.put((seg, ev))

    def drain(self):
        """Wait until every queued bucket has been exchanged (host), then order the current stream behind the copy stream."""
        t0 = time.perf_counter()
        self.q.join()
        self.host_wait_s += time.perf_counter() - t0
        if self.error is not None:
            err, self.error = self.error, None
            raise RuntimeError(f"copy-engine gradient exchange failed: {err!r}")
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def close(self):
        self.q.put(None)

    # ---- helper thread
    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            job = self.q.get()
            try:
                if job is None:
                    return
                if self.error is None:
                    self._exchange(*job)
            except Exception as e:   # surfaced by drain(); later jobs are skipped (every rank fails the same way or the barrier below times out loudly)
                self.error = e
            finally:
                self.q.task_done()

    def _host_fence(self):
        self.stream.synchronize()
        dist.barrier(group=self.ctrl)

    def _exchange(self, seg: torch.Tensor, ready: torch.cuda.Event):
        W, r = self.world, self.rank
        n = seg.numel()
        sl = (-(-n // W) + 7) // 8 * 8
        if sl > self.capacity:
            self._grow(sl)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            send = self.send[: W * sl]
            K.cast_f32_bf16(seg, send[:n], scale=1.0 / W)          # bf16 first, then divide in bf16 (the reference hook's order)
            if W * sl > n:
                send[n:].zero_()
            for p in range(W):
                self.peer_recv[p][r, :sl].copy_(send[p * sl:(p + 1) * sl], non_blocking=True)
        self._host_fence()                                         # every rank's pieces of MY slice have landed in self.recv
        with torch.cuda.stream(self.stream):
            torch.sum(self.recv[:, :sl], dim=0, dtype=torch.float32, out=self.acc[:sl])
            self.red[:sl].copy_(self.acc[:sl])
            for p in range(W):
                self.peer_gather[p][r * sl:(r + 1) * sl].copy_(self.red[:sl], non_blocking=True)
        self._host_fence()                                         # every reduced slice has landed in self.gather
        with torch.cuda.stream(self.stream):
            K.cast_bf16_f32(self.gather[:n], seg, scale=1.0)
        seg.record_stream(self.stream)
        self.bytes_copied += 2 * (W - 1) * sl * 2                  # bytes this rank pushed to peers (both phases)


def setup(rank: int, world: int, device: torch.device, process_group=None):
    """Create the exchange on every rank, or return None on EVERY rank if any rank cannot (the caller keeps the RCCL schedule).  Collective."""
    ctrl, cx, ok = None, None, 1
    try:
        ranks = list(range(dist.get_world_size(process_group)))
        if process_group is not None:
            ranks = [dist.get_global_rank(process_group, i) for i in ranks]
        ctrl = dist.new_group(ranks=ranks, backend="gloo")          # control plane only: barriers and the one-time handle exchange
    except Exception:
        ok = 0
    if ok:
        try:
            cx = CopyEngineExchange(rank, world, device, ctrl)
            cx._grow(1 << 20)                                       # proves IPC export / import (and peer access) on this node
        except Exception:
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    if ctrl is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctrl)
    if int(flag.item()) == 0:
        if cx is not None:
            cx.close()
        return None
    return cx
