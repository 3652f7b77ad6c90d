"""Where the cycles of the asm K loop go (library built with `make -C unidisc_amd/csrc UDM_QUADLOOP=timeline`): per wave the loop's total shader cycles and the cycles
spent in the tile boundary's two waits (this wave's LDS-DMA pieces: vmcnt; the other waves: barrier), against the matrix pipe's 32.6 cycles per MFMA."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

M = 10240
K.gemm_set_quad(2)
K.debug_set("gemm_quad_asm", int(os.environ.get("UDM_QUAD_ASM_MODE", "1")))      # 2: the 16x16x32 statement
g = torch.Generator(device="cuda").manual_seed(0)
for (N, Kd, FM) in [(2048, 8192, 5), (2048, 2048, 5), (8192, 2048, 5)]:
    a = torch.randn(M, Kd, device="cuda", generator=g).bfloat16()
    b = torch.randn(N, Kd, device="cuda", generator=g).bfloat16()
    grid = (M // (64 * FM)) * (N // 256)
    tl = torch.zeros(grid, 4, 4, dtype=torch.int32, device="cuda")
    K.debug_set("gemm_quad_timeline", tl.data_ptr())      # (the diagnostic build stores its stamps on every launch: the pointer goes in first)
    for _ in range(5):
        K.gemm_nt(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.gemm_nt(a, b); e1.record()
    torch.cuda.synchronize()
    t = tl.cpu().numpy().astype("int64") & 0xFFFFFFFF
    nk = Kd // 64
    mf = nk * 4 * FM * 4       # counted as 32x32x16 MFMAs (the 16x16x32 loop issues twice as many of half the size)
    tot, vm, bar = t[..., 0].mean(), t[..., 1].mean(), t[..., 2].mean()
    us = e0.elapsed_time(e1) * 1e3
    print(json.dumps(dict(N=N, K=Kd, launch_us=round(us, 1), loop_cycles=round(tot), cycles_per_mfma=round(tot / mf, 2), vmcnt_wait_per_tile=round(vm / (nk - 1), 1),
                          barrier_wait_per_tile=round(bar / (nk - 1), 1), mfma_floor_per_tile=round(4 * FM * 4 * 32.6), loop_cycles_min=int(t[..., 0].min()), loop_cycles_max=int(t[..., 0].max()),
                          implied_clock_ghz=round(tot / us / 1e3, 2) if grid <= 256 else None)))      # (one round of tiles: the loop IS the launch, up to prologue + epilogue)
