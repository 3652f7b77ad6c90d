#!/usr/bin/env python3
"""Where does the qk-norm weight-gradient error come from?  (q_norm / k_norm gradients at 2.7-3x the oracle's emulated bf16 floor)

The oracle's bf16 emulation rounds at the reference's tensor boundaries but runs the attention backward itself in fp32, while every flash-attention
backward (the reference's flash_attn / SDPA kernels and this repo's) rounds P and dS to bf16 before the dQ / dK / dV matrix products.  This script
measures, on LayerNorm-ed + rotated-like q, k at the headline shape, against an fp64 attention backward:
  (a) this repo's udm_attention_bwd,  (b) torch SDPA in bf16 (whatever backend the build has),  (c) an explicit FA2-rounding emulation in torch
      (P -> bf16, dS -> bf16, fp32 accumulate),  (d) the oracle-style emulation (fp32 backward, dq / dk rounded to bf16 at the output only),
for dq, dk, dv and for the LayerNorm-weight-gradient proxy  sum_rows dq * q_hat  (a column sum over all rows with heavy cancellation).
Prints one JSON object.
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def main():
    dev = "cuda"
    B, H, L, D = 2, 16, 1280, 128
    d = H * D
    g = torch.Generator(device="cpu").manual_seed(3)
    bf = lambda t: t.to(torch.bfloat16)
    # q, k: LayerNorm over the hidden size with a learned-looking affine, then a rotation-like mixing (unit-variance entries); v, dO: plain Gaussians
    w = 1 + 0.1 * torch.randn(d, generator=g)
    q32 = F.layer_norm(torch.randn(B * L, d, generator=g), [d]) * w
    k32 = F.layer_norm(torch.randn(B * L, d, generator=g), [d]) * w
    v32 = 0.5 * torch.randn(B * L, d, generator=g)
    do32 = 0.02 * torch.randn(B * L, d, generator=g)
    q, k, v, do = (bf(t).to(dev) for t in (q32, k32, v32, do32))
    scale = D ** -0.5

    def heads(t):   # [M, d] -> [B, H, L, D]
        return t.reshape(B, L, H, D).permute(0, 2, 1, 3)

    def flat(t):
        return t.permute(0, 2, 1, 3).reshape(B * L, d)

    # fp64 truth on the bf16-valued inputs
    qh, kh, vh, doh = (heads(t.double()) for t in (q, k, v, do))
    S = qh @ kh.transpose(-1, -2) * scale
    P = torch.softmax(S, -1)
    O = P @ vh
    dV = P.transpose(-1, -2) @ doh
    dP = doh @ vh.transpose(-1, -2)
    delta = (doh * O).sum(-1, keepdim=True)
    dS = P * (dP - delta)
    dQ, dK = dS @ kh * scale, dS.transpose(-1, -2) @ qh * scale
    truth = dict(dq=flat(dQ), dk=flat(dK), dv=flat(dV))
    qhat = q.double()   # proxy for x_hat of the LayerNorm backward (the rotation is orthogonal; magnitudes are what matters here)
    khat = k.double()

    def report(dq, dk, dv):
        return dict(dq=rel(dq, truth["dq"]), dk=rel(dk, truth["dk"]), dv=rel(dv, truth["dv"]),
                    qnorm_w_proxy=rel((dq.double() * qhat).sum(0), (truth["dq"] * qhat).sum(0)),
                    knorm_w_proxy=rel((dk.double() * khat).sum(0), (truth["dk"] * khat).sum(0)))

    out = {}
    # (a) this repo
    o, lse = K.attention_fwd_generic(q, k, v, B, L, H, D)
    dq, dk, dv = K.attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D)
    out["unidisc_amd"] = report(dq, dk, dv)
    out["unidisc_amd"]["o"] = rel(o, flat(O))
    # (b) torch SDPA bf16
    try:
        qt, kt, vt = (heads(t).detach().clone().requires_grad_() for t in (q, k, v))
        ot = F.scaled_dot_product_attention(qt, kt, vt)
        ot.backward(heads(do))
        out["torch_sdpa_bf16"] = report(flat(qt.grad), flat(kt.grad), flat(vt.grad))
        out["torch_sdpa_bf16"]["o"] = rel(flat(ot.detach()), flat(O))
    except Exception as e:   # noqa: BLE001
        out["torch_sdpa_bf16"] = f"failed: {type(e).__name__}: {e}"
    # (c) FA2 rounding points, explicit (fp32 accumulate)
    qf, kf, vf, dof = (heads(t.float()) for t in (q, k, v, do))
    S32 = qf @ kf.transpose(-1, -2) * scale
    lse32 = torch.logsumexp(S32, -1, keepdim=True)
    P32 = torch.exp(S32 - lse32)
    O16 = bf(bf(P32).float() @ vf).float()
    dlt = (dof * O16).sum(-1, keepdim=True)
    P16 = bf(P32).float()
    dV_c = P16.transpose(-1, -2) @ dof
    dP_c = dof @ vf.transpose(-1, -2)
    dS16 = bf(P32 * (dP_c - dlt)).float()
    dQ_c, dK_c = bf(dS16 @ kf * scale), bf(dS16.transpose(-1, -2) @ qf * scale)
    out["fa2_rounding_emulation"] = report(flat(dQ_c), flat(dK_c), flat(bf(dV_c)))
    # (d) oracle-style: fp32 backward, only the outputs rounded
    dS32 = P32 * (dP_c - (dof * (P32 @ vf)).sum(-1, keepdim=True))
    out["oracle_style_fp32_backward"] = report(flat(bf(dS32 @ kf * scale)), flat(bf(dS32.transpose(-1, -2) @ qf * scale)), flat(bf(P32.transpose(-1, -2) @ dof)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
