"""A/B of the two forward attention kernels at head dim 128 (no document mask): bit-identity of O / LSE and wall time per call."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K


MODE = int(os.environ.get("MODE", "1"))   # 1 = one wave per SIMD (attention_w64.hip, product option), 2 = wave-specialised (attention_ws64.hip, experiments library)
_EXP = None


def fwd_new(q, k, v, B, L, H, D):
    """the kernel under test: product entry point with the w64 switch on, or the experiments library's wave-specialised kernel"""
    if MODE == 1 or L % 128 != 0:
        K.set_attention_w64(True)
        return K.attention_fwd_generic(q, k, v, B, L, H, D)
    global _EXP
    import ctypes
    from unidisc_amd import _lib
    if _EXP is None:
        _EXP = _lib.load_experiments()
    d = H * D
    o = torch.empty((B * L, d), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    i64 = ctypes.c_int64
    rc = _EXP.udm_exp_attention_fwd_ws64(vp(q), vp(k), vp(v), vp(o), vp(lse), i64(B), i64(H), i64(L), i64(d), i64(d), i64(d), i64(d), ctypes.c_void_p(0),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(_EXP.udm_last_error().decode())
    return o, lse


def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(B, H, L, D=128, spike=False, bench=True):
    g = torch.Generator(device="cuda").manual_seed(L)
    q, k, v = ((torch.randn(B * L, H * D, device="cuda", generator=g)).to(torch.bfloat16) for _ in range(3))
    if spike and L > 200:   # force the lazy-rescale branch late in the sequence
        k[L - 100] *= 12
        q[7] = (k[L - 100].float() / 4).to(torch.bfloat16)
    K.set_attention_w64(False)
    o0, l0 = K.attention_fwd_generic(q, k, v, B, L, H, D)
    o1, l1 = fwd_new(q, k, v, B, L, H, D)
    torch.cuda.synchronize()
    rec = dict(B=B, H=H, L=L, spike=spike, o_equal=bool(torch.equal(o0, o1)), lse_equal=bool(torch.equal(l0, l1)),
               o_maxdiff=float((o0.float() - o1.float()).abs().max()), lse_maxdiff=float((l0 - l1).abs().max()), finite=bool(torch.isfinite(o1.float()).all()))
    if bench:
        fl = 4 * B * H * L * L * D
        K.set_attention_w64(False)
        t0 = timeit(lambda: K.attention_fwd_generic(q, k, v, B, L, H, D))
        t1 = timeit(lambda: fwd_new(q, k, v, B, L, H, D))
        rec.update(old_us=round(t0 * 1e3, 1), new_us=round(t1 * 1e3, 1), old_tf=round(fl / t0 / 1e9), new_tf=round(fl / t1 / 1e9))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if os.environ.get("UDM_ATTN_W64_ABL"):   # timing-only ablations (wrong results): just the headline shape
        run(8, 16, 1280)
        sys.exit(0)
    for (B, H, L) in [(1, 1, 64), (1, 1, 2), (2, 3, 100), (1, 2, 257), (3, 5, 640), (2, 2, 1000), (1, 1, 191), (2, 1, 129), (1, 1, 128), (2, 3, 256), (1, 2, 384), (3, 1, 1152)]:
        run(B, H, L, bench=False)
        run(B, H, L, spike=True, bench=False)
    for (B, H, L) in [(8, 16, 1280), (8, 16, 1024), (8, 16, 2048), (2, 16, 4608), (8, 16, 1536)]:
        run(B, H, L)
