"""Probe: can two processes on ONE GPU share device memory through torch's CUDA-IPC reductions on this pool (dmabuf IPC), and copy into each other's buffers?
Prints one JSON line.  (Feasibility check for a copy-engine gradient exchange)"""
import json, os, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.multiprocessing.reductions import reduce_tensor


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    res = {}
    try:
        mine = torch.full((1 << 20,), float(rank + 1), device="cuda")          # 4 MB
        handle = reduce_tensor(mine)                                           # (rebuild_fn, args): picklable IPC handle
        handles = [None] * world
        dist.all_gather_object(handles, handle)
        peers = [fn(*args) if i != rank else mine for i, (fn, args) in enumerate(handles)]
        dist.barrier()
        # every rank writes its id into slot [rank] region of every peer buffer
        n = mine.numel() // world
        src = torch.full((n,), 10.0 * (rank + 1), device="cuda")
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for p in peers:
                p[rank * n:(rank + 1) * n].copy_(src, non_blocking=True)
        s.synchronize()
        dist.barrier()
        got = [float(mine[i * n]) for i in range(world)]
        res = dict(rank=rank, ok=got == [10.0 * (i + 1) for i in range(world)], got=got)
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            for _ in range(20):
                for p in peers:
                    p[rank * n:(rank + 1) * n].copy_(src, non_blocking=True)
        s.synchronize()
        res["copy_GBps"] = 20 * world * n * 4 / (time.perf_counter() - t0) / 1e9
    except Exception as e:
        res = dict(rank=rank, ok=False, error=f"{type(e).__name__}: {e}"[:300])
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    out = [q.get(timeout=120) for _ in ps]
    [p.join(30) for p in ps]
    print(json.dumps(out))
