"""Timing-only ablations of the 64-queries-per-wave attention forward (UDM_ATTN_FWD64_ABL read once per process: this script re-runs itself per value).
Needs a library built with `make -C unidisc_amd/csrc UDM_FWD64_ABL="1 3 7"`."""
import json, os, subprocess, sys
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from unidisc_amd import kernels as K
    B, H, L, D = 8, 16, 1280, 128
    d, M = H * D, B * L
    g = torch.Generator(device="cuda").manual_seed(0)
    qkr = torch.randn(M, 2 * d, device="cuda", generator=g)
    qkr[:, :d] *= K.attention_q_scale(D)
    qkr = qkr.to(torch.bfloat16)
    qkv = torch.randn(M, 3 * d, device="cuda", generator=g).to(torch.bfloat16)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.attention_fwd(qkr, qkv, B, L, H, D, q_prescaled=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(json.dumps({"abl": int(sys.argv[1]), "median_us": round(ts[15], 1), "min_us": round(ts[0], 1)}))
else:
    for abl in (0, 1, 3, 7, 0):   # 1 = no softmax arithmetic, 3 = also no fragment reads, 7 = MFMAs only
        env = dict(os.environ, UDM_ATTN_FWD64_ABL=str(abl))
        r = subprocess.run([sys.executable, __file__, str(abl)], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
