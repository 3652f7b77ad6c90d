"""fp8 attention forward (block-scaled 32x32x64 e4m3 MFMA) against the bf16 forward on the headline and the packed config-E shapes, with the cost of the
operand preparation (V quantise + transpose pass; the q / k quantisation rides inside the qk-norm + rope kernel and is timed there).
    python scripts/bench_attn_fp8.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K
from unidisc_amd import _lib
from unidisc_amd.kernels import _p, _s


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


H, D = 16, 128
d = H * D
g = torch.Generator().manual_seed(0)
for name, B, L, docs in (("headline", 8, 1280, 1), ("separate", 8, 1152, 1), ("packed", 2, 4608, 4)):
    M = B * L
    # rotating buffer sets so that operands come from HBM / MALL as in the step, not from a warm L2
    sets = []
    for _ in range(3):
        q, k, v = (torch.randn(M, d, generator=g).bfloat16().cuda() for _ in range(3))
        qkr = torch.cat([q, k], 1).contiguous()
        qk8, qk_e8 = K.attention_quantize_qk_fp8(qkr, D)
        v8t, v_e8 = K.attention_quantize_v_fp8(v.data_ptr(), d, B, L, H, D, q.device)
        sets.append((q, k, v, qkr, qk8, qk_e8, v8t, v_e8))
    sid = (torch.arange(L) // (L // docs))[None].repeat(B, 1).cuda() if docs > 1 else None
    r = K.attention_doc_ranges(sid) if sid is not None else None
    o = torch.empty((M, d), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty((B, H, L), dtype=torch.float32, device="cuda")
    it = [0]

    def f16():
        q, k, v = sets[it[0] % 3][:3]
        it[0] += 1
        K.attention_fwd_generic(q, k, v, B, L, H, D, sid, r)

    def f8():
        s_ = sets[it[0] % 3]
        it[0] += 1
        _lib.call("udm_attention_fwd_fp8", _p(s_[4]), _p(s_[5]), _p(s_[6]), _p(s_[7]), _p(o), _p(lse), _p(sid), _p(r), B, H, L, D, d, _s())

    def fv():
        s_ = sets[it[0] % 3]
        it[0] += 1
        K.attention_quantize_v_fp8(s_[2].data_ptr(), d, B, L, H, D, "cuda")

    def fqk():
        s_ = sets[it[0] % 3]
        it[0] += 1
        K.attention_quantize_qk_fp8(s_[3], D)

    a, b, c, e = t(f16), t(f8), t(fv), t(fqk)
    fl = 4.0 * B * H * docs * (L // docs) ** 2 * D
    print(f"{name:9s} B={B} L={L}: bf16 fwd {a:6.1f} us ({fl / a / 1e6:5.0f} TF)   fp8 fwd {b:6.1f} us ({fl / b / 1e6:5.0f} TF)   v quantise {c:5.1f} us   "
          f"(generic q/k quantise pass {e:5.1f} us: fused into qk-norm + rope at d = 2048)")
# the fused qk-norm + rope kernel with and without the fp8 emission (d = 2048)
for B, L in ((8, 1280), (2, 4608)):
    M = B * L
    qkv = torch.randn(M, 3 * d, generator=g).bfloat16().cuda()
    ang = torch.randn(L, D // 2, generator=g)
    cos, sin = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda()
    w = [torch.randn(d, generator=g).cuda() for _ in range(4)]
    kw = dict(gq=w[0], bq=w[1], gk=w[2], bk=w[3])
    a = t(lambda: K.qknorm_rope_fwd(qkv, cos, sin, L, D, **kw))
    b = t(lambda: K.qknorm_rope_fwd(qkv, cos, sin, L, D, fp8=True, **kw))
    print(f"qk-norm + rope forward M={M}: plain {a:5.1f} us   with fp8 emission {b:5.1f} us")
