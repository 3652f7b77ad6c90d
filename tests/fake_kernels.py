"""CPU stand-ins for ``unidisc_amd.kernels`` used ONLY by tests/test_engine_orchestration.py.

They let the hand-scheduled forward/backward engine in ``unidisc_amd/dit.py`` (buffer plumbing, gradient
bookkeeping, adaLN chunk indexing, callbacks) be exercised in the GPU-less container.  They are test doubles:
nothing in the product imports them, and the product has no CPU path.
"""
import math

import torch
import torch.nn.functional as F

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_DGELU = 0, 1, 2, 3
NORM_RMS, NORM_LN = 0, 1
BF16, F32 = torch.bfloat16, torch.float32


def require_gpu(t):
    return None


def norm_id(norm_type):
    return NORM_RMS if norm_type == "rms" else NORM_LN


@torch.enable_grad()
def gemm_nt(a, b, out=None, *, out_dtype=BF16, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, epilogue=EPI_NONE, bias=None, aux=None, ldaux=None,
            beta=0.0):
    M = a.shape[0] if M is None else M
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    acc = a[:M, :K].float() @ b[:N, :K].float().t()
    if epilogue in (EPI_BIAS, EPI_BIAS_GELU):
        acc = acc + bias[:N].float()
    if epilogue == EPI_BIAS_GELU:   # aux receives bf16(gelu'(u)), u = bf16(pre-activation)
        u = acc.bfloat16().float().requires_grad_()
        y = F.gelu(u, approximate="tanh")
        (gr,) = torch.autograd.grad(y.sum(), u)
        aux[:M, :N] = gr.bfloat16()
        acc = y.detach()
    if epilogue == EPI_DGELU:
        acc = acc * aux[:M, :N].float()
        if bias is not None:  # EPI_DGELU: `bias` is the bias-gradient OUTPUT (column sums of the result)
            bias[:N] += acc.detach().bfloat16().float().sum(0)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype)
    if beta != 0.0:
        acc = acc + beta * out[:M, :N].float()
    out[:M, :N] = acc.to(out.dtype)
    return out


def transpose(x, out=None, colsum=None, R=None, C=None):
    if colsum is not None:
        colsum += x.float().sum(0)
    t = x.t().contiguous()
    if out is not None:
        out.copy_(t)
        return out
    return t


def cast_transpose(w, out, out_t):
    R, C = w.shape
    if out is not None:
        out[:R, :C] = w.bfloat16()
    if out_t is not None:
        out_t[:C, :R] = w.t().bfloat16()


def cast_transpose_jobs(items, device):
    return list(items), len(items), 0


def cast_transpose_multi(jobs):
    for w, out, out_t in jobs[0]:
        cast_transpose(w, out, out_t)


def cast_f32_bf16(x, y, scale=1.0):
    y.copy_((x.bfloat16().float() * scale).bfloat16())
    return y


def cast_bf16_f32(x, y, scale=1.0):
    y.copy_(x.float() * scale)
    return y


def _rows_to_batch(M, L):
    return torch.arange(M) // L


def _modulated(n, mod, idx, modality, any_img, L):
    if mod is None:
        return n
    d = n.shape[1]
    b = _rows_to_batch(n.shape[0], L)
    sh, sc = mod[b, idx[0] * d:(idx[0] + 1) * d], mod[b, idx[1] * d:(idx[1] + 1) * d]
    out = n * (1 + sc) + sh
    if modality is not None and (any_img is None or int(any_img) != 0):
        out = torch.where((modality == 1)[:, None], out, n)
    return out


def _norm(x, w, nt):
    if nt == NORM_RMS:
        rstd = torch.rsqrt(x.pow(2).mean(-1) + 1e-6)
        return x * rstd[:, None] * w, rstd, None
    mean = x.mean(-1)
    rstd = torch.rsqrt(x.var(-1, unbiased=False) + 1e-5)
    return (x - mean[:, None]) * rstd[:, None] * w, rstd, mean


def norm_fwd(x, w, norm_type, L, *, mod=None, mod_idx=(0, 1), modality=None, any_img=None):
    n, rstd, mean = _norm(x, w, norm_type)
    y = _modulated(n, mod.float() if mod is not None else None, mod_idx, modality, any_img, L)
    return y.bfloat16(), rstd, mean


@torch.enable_grad()
def norm_bwd(dy, x, rstd, mean, w, norm_type, L, dx, dw, *, accumulate=True, mod=None, dmod=None, mod_idx=(0, 1), modality=None, any_img=None):
    x_, w_ = x.clone().requires_grad_(), w.clone().requires_grad_()
    mod_ = mod.float().clone().requires_grad_() if mod is not None else None
    n, _, _ = _norm(x_, w_, norm_type)
    y = _modulated(n, mod_, mod_idx, modality, any_img, L)
    y.backward(dy.float())
    dx.copy_(x_.grad + (dx if accumulate else 0))
    dw += w_.grad
    if mod is not None:
        dmod += mod_.grad


def _branch(x_in, br, L, w_b, norm_type, mod, gate_idx, modality):
    n = br
    rstd = mean = None
    if w_b is not None:
        if norm_type == NORM_RMS:
            rstd = torch.rsqrt(br.pow(2).mean(-1) + 1e-6)
            nh = br * rstd[:, None]
            n = (nh + (nh.detach().bfloat16().float() - nh.detach())) * w_b  # straight-through bf16 rounding
        else:
            n, rstd, mean = _norm(br, w_b, norm_type)
    out = n
    if mod is not None and gate_idx is not None:
        d = br.shape[1]
        g = mod[_rows_to_batch(br.shape[0], L), gate_idx * d:(gate_idx + 1) * d]
        out = g * n
        if modality is not None:
            out = torch.where((modality == 1)[:, None], out, n)
    return x_in + out, rstd, mean


def residual_fwd(x_in, branch, L, *, w_b=None, norm_type=NORM_RMS, mod=None, gate_idx=None, modality=None, p_drop=0.0, seed=0, next_w=None, next_mod=None,
                 next_mod_idx=(0, 1), next_modality=None, next_any_img=None):
    assert p_drop == 0.0, "fake kernels: dropout not emulated"
    out, rstd, mean = _branch(x_in, branch.float(), L, w_b, norm_type, mod.float() if mod is not None else None, gate_idx, modality)
    if next_w is None:
        return out, rstd, mean
    return out, rstd, mean, norm_fwd(out, next_w, norm_type, L, mod=next_mod, mod_idx=next_mod_idx, modality=next_modality, any_img=next_any_img)


@torch.enable_grad()
def norm_residual_bwd(dy, x, rstd, mean, w, norm_type, L, dx, dw, branch, *, accumulate=True, w_b=None, rstd_b=None, mean_b=None, dw_b=None, p_drop=0.0, seed=0,
                      dbias=None):
    norm_bwd(dy, x, rstd, mean, w, norm_type, L, dx, dw, accumulate=accumulate)
    out = residual_bwd(dx, branch, L, w_b=w_b, rstd=rstd_b, mean=mean_b, norm_type=norm_type, dw_b=dw_b, p_drop=p_drop, seed=seed)
    if dbias is not None:
        colsum(out, dbias)
    return out


def norm_residual_bwd_ada_ok(M, d, L):
    return L > 0 and M % L == 0


def norm_residual_bwd_ada(dy, x, rstd, mean, w, norm_type, L, dx, dw, branch, *, accumulate=True, w_b=None, rstd_b=None, mean_b=None, dw_b=None, p_drop=0.0, seed=0,
                          dbias=None, mod_n=None, dmod_n=None, mod_idx=(0, 1), modality=None, any_img=None, mod_r=None, dmod_r=None, gate_idx=None, modality_r=None):
    norm_bwd(dy, x, rstd, mean, w, norm_type, L, dx, dw, accumulate=accumulate, mod=mod_n, dmod=dmod_n, mod_idx=mod_idx, modality=modality, any_img=any_img)
    out = residual_bwd(dx, branch, L, w_b=w_b, rstd=rstd_b, mean=mean_b, norm_type=norm_type, mod=mod_r, dmod=dmod_r, gate_idx=gate_idx, modality=modality_r, dw_b=dw_b,
                       p_drop=p_drop, seed=seed)
    if dbias is not None:
        colsum(out, dbias)
    return out


@torch.enable_grad()
def residual_bwd(dx, branch, L, *, w_b=None, rstd=None, mean=None, norm_type=NORM_RMS, mod=None, dmod=None, gate_idx=None, modality=None, dw_b=None,
                 p_drop=0.0, seed=0):
    br = branch.float().clone().requires_grad_()
    w_ = w_b.clone().requires_grad_() if w_b is not None else None
    mod_ = mod.float().clone().requires_grad_() if (mod is not None and gate_idx is not None) else None
    out, _, _ = _branch(torch.zeros_like(dx), br, L, w_, norm_type, mod_, gate_idx, modality)
    out.backward(dx)
    if w_b is not None:
        dw_b += w_.grad
    if mod_ is not None:
        dmod += mod_.grad
    return br.grad.bfloat16()


def _qk(qkv, cos, sin, L, D, gq, bq, gk, bk):
    M, d3 = qkv.shape
    d = d3 // 3
    H = d // D
    q, k = qkv[:, :d], qkv[:, d:2 * d]
    if gq is not None:
        q = F.layer_norm(q, [d], gq, bq, 1e-5)
        k = F.layer_norm(k, [d], gk, bk, 1e-5)
        q = q + (q.detach().bfloat16().float() - q.detach())
        k = k + (k.detach().bfloat16().float() - k.detach())
    qk = torch.stack([q, k], 1).reshape(M, 2 * H, D)
    if cos.dim() == 3:
        c, s = cos.reshape(M, -1), sin.reshape(M, -1)
    else:
        idx = torch.arange(M) % L
        c, s = cos[idx], sin[idx]
    c, s = torch.cat([c, c], -1)[:, None], torch.cat([s, s], -1)[:, None]
    x1, x2 = qk.chunk(2, -1)
    out = qk * c + torch.cat([-x2, x1], -1) * s
    return out.reshape(M, 2 * d)


def attention_q_scale(D):
    return 1.4426950408889634 / math.sqrt(D)


def qknorm_rope_fwd(qkv, cos, sin, L, D, *, gq=None, bq=None, gk=None, bk=None, q_scale=1.0):
    out = _qk(qkv.float(), cos, sin, L, D, gq, bq, gk, bk)
    d = out.shape[1] // 2
    out = torch.cat([out[:, :d] * q_scale, out[:, d:]], 1).bfloat16()
    stats = torch.zeros(qkv.shape[0], 4) if gq is not None else None
    return out, stats


_saved_qk = {}


@torch.enable_grad()
def qknorm_rope_bwd(dqkr, qkv, dqkv, cos, sin, L, D, *, gq=None, gk=None, stats=None, dgq=None, dbq=None, dgk=None, dbk=None, bq=None, bk=None, q_scale=1.0):
    d = qkv.shape[1] // 3
    if q_scale != 1.0:   # the incoming dq is the gradient wrt q * q_scale
        dqkr = torch.cat([dqkr[:, :d].float() * q_scale, dqkr[:, d:].float()], 1)
    x = qkv.float().clone().requires_grad_()
    p = [t.clone().requires_grad_() if t is not None else None for t in (gq, gk)]
    zeros = torch.zeros(d)
    out = _qk(x, cos, sin, L, D, p[0], zeros if gq is not None else None, p[1], zeros if gq is not None else None)
    out.backward(dqkr.float())
    dqkv[:, :2 * d] = x.grad[:, :2 * d].bfloat16()
    if gq is not None:
        dgq += p[0].grad
        dgk += p[1].grad
        # d beta = column sums of the gradient w.r.t. the LayerNorm output = grad through the rotation
        g = dqkr.float().reshape(qkv.shape[0], -1, D)
        M = qkv.shape[0]
        if cos.dim() == 3:
            c, s = cos.reshape(M, -1), sin.reshape(M, -1)
        else:
            idx = torch.arange(M) % L
            c, s = cos[idx], sin[idx]
        c, s = c[:, None], s[:, None]
        g1, g2 = g.chunk(2, -1)
        gl = g1 * c + g2 * s
        gh = g2 * c - g1 * s
        gb = torch.cat([gl, gh], -1).reshape(M, 2 * d)
        dbq += gb[:, :d].sum(0)
        dbk += gb[:, d:].sum(0)


def _attn(q, k, v, B, L, H, D, sid):
    q, k, v = (t.reshape(B, L, H, D).transpose(1, 2) for t in (q, k, v))
    s = q @ k.transpose(-1, -2) / math.sqrt(D)
    if sid is not None:   # mask codes (csrc/attention_common.h: attn_pair_ok): low 32 bits sample id, bits 32-39 key class, 40-47 query mask
        ids = ((sid & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000   # sign-extended low half
        kb, qm = (sid >> 32) & 0xFF, (sid >> 40) & 0xFF
        kb, qm = torch.where(kb == 0, torch.full_like(kb, 0xFF), kb), torch.where(qm == 0, torch.full_like(qm, 0xFF), qm)
        allow = (ids[:, :, None] == ids[:, None, :]) & (ids[:, :, None] >= 0) & ((qm[:, :, None] & kb[:, None, :]) != 0)
        s = s.masked_fill(~allow[:, None], float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, -1), nan=0.0)
    return (p @ v).transpose(1, 2).reshape(B * L, H * D)


def modality_mask_codes(txt_drop, img_drop, txt_length, L):
    from unidisc_amd.kernels import modality_mask_codes as f
    return f(txt_drop, img_drop, txt_length, L)


def attention_doc_ranges(sample_ids):
    B, L = sample_ids.shape
    nT = (L + 63) // 64
    r = torch.zeros(B, nT, 8, dtype=torch.int32)
    for b in range(B):
        for t in range(nT):
            tile = sample_ids[b, t * 64:(t + 1) * 64]
            ids = tile[tile >= 0]
            r[b, t, 2], r[b, t, 3] = -1, -2
            if ids.numel():
                hit = ((sample_ids[b] >= ids.min()) & (sample_ids[b] <= ids.max())).nonzero().flatten()
                r[b, t, 0], r[b, t, 1] = int(hit[0]), int(hit[-1]) + 1
                r[b, t, 2], r[b, t, 3] = (int(ids.min()) if ids.numel() == tile.numel() else -1), int(ids.max())
                r[b, t, 4] = int(r[b, t, 2] >= 0 and r[b, t, 2] == r[b, t, 3] and hit.numel() == int(hit[-1]) + 1 - int(hit[0]))
    return r


def attention_fwd(qkr, qkv, B, L, H, D, sample_ids=None, doc_ranges=None, q_prescaled=False):
    d = H * D
    qs = attention_q_scale(D) if q_prescaled else 1.0
    o = _attn(qkr[:, :d].float() / qs, qkr[:, d:].float(), qkv[:, 2 * d:].float(), B, L, H, D, sample_ids)
    return o.bfloat16(), torch.zeros(B, H, L)


@torch.enable_grad()
def attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, sample_ids=None, doc_ranges=None, q_prescaled=False):
    d = H * D
    qs = attention_q_scale(D) if q_prescaled else 1.0
    q, k, v = (t.float().clone().requires_grad_() for t in (qkr[:, :d], qkr[:, d:], qkv[:, 2 * d:]))
    _attn(q / qs, k, v, B, L, H, D, sample_ids).backward(do.float())   # (q.grad is then the gradient wrt the stored, scaled q)
    dqkr[:, :d], dqkr[:, d:], dqkv[:, 2 * d:] = q.grad.bfloat16(), k.grad.bfloat16(), v.grad.bfloat16()


def embedding_fwd(ids, E, modality=None, Em=None):
    x = E[ids]
    if Em is not None:
        x = x + Em[(modality != 0).long()]
    return x


def embedding_bwd(ids, dx, dE, hot_id, modality=None, dEm=None):
    dE.index_add_(0, ids, dx)
    if dEm is not None:
        dEm.index_add_(0, (modality != 0).long(), dx)


def _valid(M, V, Vt, mask_id, modality, restrict):
    v = torch.ones(M, V, dtype=torch.bool)
    if restrict:
        img = (modality == 1)[:, None]
        ar = torch.arange(V)[None]
        v = torch.where(img, ar >= Vt, ar < Vt)
    v[:, mask_id] = False
    return v


def subs_ce_fwd(logits, x0, xt, modality, V, Vt, mask_id, restrict):
    M = logits.shape[0]
    z = logits[:, :V].float().masked_fill(~_valid(M, V, Vt, mask_id, modality, restrict), float("-inf"))
    lse = torch.logsumexp(z, -1)
    zx = z.gather(1, x0[:, None])[:, 0]
    zx = torch.where(torch.isinf(zx), torch.full_like(zx, -1e6), zx)
    masked = xt == mask_id
    log_p = torch.where(masked, zx - lse, torch.where(x0 == xt, torch.zeros_like(lse), torch.full_like(lse, -1e6)))
    return log_p, torch.where(masked, lse, torch.zeros_like(lse))


def subs_ce_bwd(logits, x0, xt, modality, lse, g, V, Vt, mask_id, restrict, narrow_txt_rows=-1):
    M = logits.shape[0]
    valid = _valid(M, V, Vt, mask_id, modality, restrict)
    z = logits[:, :V].float()
    p = torch.where(valid, torch.exp(z - lse[:, None]), torch.zeros_like(z))
    onehot = F.one_hot(x0, V).float() * valid
    dl = g[:, None] * (onehot - p)
    dl = torch.where((xt == mask_id)[:, None], dl, torch.zeros_like(dl))
    if narrow_txt_rows >= 0 and restrict:   # only the columns the row's GROUP of head GEMMs reads are defined: poison the rest, as the kernel leaves it unwritten
        ld = logits.shape[1]
        img_group = (torch.arange(M) >= narrow_txt_rows)[:, None]
        col = torch.arange(ld)[None]
        keep = torch.where(img_group, col >= Vt // 8 * 8, col < min((Vt + 63) // 64 * 64, ld))
        full = torch.zeros_like(logits, dtype=torch.float32)
        full[:, :V] = dl
        logits.copy_(torch.where(keep, full, torch.full_like(full, float("nan"))).bfloat16())
        return
    logits.zero_()
    logits[:, :V] = dl.bfloat16()


@torch.enable_grad()
def diffusion_loss(log_p, w_loss, w_std, attention_mask, modality_mask, *, weighted, full_mask=False, text_w=1.0, img_w=1.0, ratio=None):
    """CPU double of udm_diffusion_loss: the reference's own sequence of tensor statements (model.py:1010-1160), d loss / d log_p through autograd."""
    lp = log_p.detach().clone().requires_grad_()
    loss = -lp * w_loss[:, None]
    nlls = (-log_p * w_std[:, None]) * attention_mask
    sc = torch.zeros(8, dtype=torch.float32, device=log_p.device)
    if modality_mask is not None:
        txt_mask, img_mask = modality_mask[..., 0] & attention_mask, modality_mask[..., 1] & attention_mask
        txt_count, img_count = txt_mask.sum(), img_mask.sum()
        total = txt_count + img_count
        txt_frac, img_frac = txt_count / total, img_count / total
        sc[3], sc[4], sc[6], sc[7] = txt_frac, img_frac, txt_count, img_count
    sc[5] = attention_mask.sum() / attention_mask.numel()
    if weighted:
        loss = loss * attention_mask
        txt_loss = ((loss * txt_mask).sum() / txt_count) * txt_frac * text_w
        img_loss = ((loss * img_mask).sum() / img_count) * img_frac * img_w
        if ratio is not None:
            max_txt_loss = float(ratio) * img_loss.detach()
            scale = torch.minimum(torch.tensor(1.0, device=txt_loss.device), max_txt_loss / (txt_loss.detach() + 1e-8))
            ok = ~(torch.isnan(img_loss.detach()) | torch.isnan(txt_loss.detach()))
            txt_loss = txt_loss * torch.where(ok, scale, torch.ones_like(scale))
        txt_loss, img_loss = torch.nan_to_num(txt_loss, nan=0.0), torch.nan_to_num(img_loss, nan=0.0)
        total_loss = txt_loss + img_loss
        sc[1], sc[2] = txt_loss.detach(), img_loss.detach()
    else:
        am = torch.ones_like(attention_mask) if full_mask else attention_mask
        total_loss = torch.nan_to_num((loss * am).sum() / am.sum(), nan=0.0)
    sc[0] = total_loss.detach()
    (coef,) = torch.autograd.grad(total_loss, lp)
    return nlls.float(), coef.float(), sc


def timestep_embedding(sigma, out, B, dim=256):
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = sigma[:, None].float() * freqs[None]
    out[:B] = torch.cat([torch.cos(args), torch.sin(args)], -1).bfloat16()


def silu_fwd(x, n=None):
    return F.silu(x.float()).bfloat16()


@torch.enable_grad()
def silu_bwd(x, dy):
    xf = x.float().requires_grad_()
    F.silu(xf).backward(dy.float())
    return xf.grad.bfloat16()


def subs_logprobs(logits, xt, modality, V, Vt, mask_id, restrict, out_dtype=BF16):
    M = logits.shape[0]
    valid = _valid(M, V, Vt, mask_id, modality, restrict)
    z = logits[:, :V].float().masked_fill(~valid, float("-inf"))
    out = z - torch.logsumexp(z, -1, keepdim=True)
    out = torch.where(valid, out, torch.full_like(out, -1e6))
    if xt is not None:
        un = (xt != mask_id)[:, None]
        onehot = F.one_hot(xt.clamp(0, V - 1), V).bool()
        out = torch.where(un, torch.where(onehot, torch.zeros_like(out), torch.full_like(out, -1e6)), out)
    return out.to(out_dtype)


CUS_CALLS = []   # gemm_set_cus history (the DDP comm-policy tests read it)


def gemm_set_cus(cus):
    CUS_CALLS.append(int(cus))


def gemm_tn_wants_splitk(M, N, K=None):
    return ((M + 255) // 256) * ((N + 255) // 256) <= 128 and M * N >= 1 << 16


def gemm_tn_pair_ok(M0, M1, N, K):
    return False   # (CPU doubles: the two wgrads stay separate calls)


def gemm_nn_ok(M, N, K):
    return N % 256 == 0 and K % 64 == 0 and K >= 128 and any(M % t == 0 for t in (192, 256, 320))


def gemm_nn(a, b, out=None, *, N=None):
    N = b.shape[1] if N is None else N
    r = (a.float() @ b[:, :N].float()).to(torch.bfloat16)
    if out is None:
        return r
    out[:, :N] = r
    return out


def gemm_tn(a, b, out, *, M=None, N=None, beta=0.0):
    M = a.shape[1] if M is None else M
    N = b.shape[1] if N is None else N
    acc = a[:, :M].float().t() @ b[:, :N].float()
    out[:M, :N] = acc + (beta * out[:M, :N] if beta != 0.0 else 0)
    return out


gemm_tn_splitk = gemm_tn


def gemm_nt_splitk(a, b, out=None, *, N=None):
    return gemm_nt(a, b, out=out, N=N)


SMALL_BATCH_LINEAR_MAX_B, SMALL_BATCH_LINEAR_MAX_IN = 64, 128


def small_batch_linear_bwd_tiles(out):
    return 3


def small_batch_linear_bwd(dy, x, w16, dw, db, dx=None, dx_parts=None):
    d16 = dy.bfloat16().float()
    out = dy.shape[1]
    dw.copy_(d16.t() @ x.float())
    if db is not None:
        db += d16.sum(0)
    full = d16 @ w16[:out].float()
    if dx_parts is not None:
        dx_parts.zero_()
        dx_parts[0] = full
    else:
        dx += full


def colsum(x, out):
    out += x.float().sum(0)
    return out


# ------------------------------------------------------------------------------------------------ optimizer (csrc/optim.hip)
def sumsq(x, out):
    out[0] = (x.float() ** 2).sum()
    return out


def _adam(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm):
    clip = 1.0
    if grad_norm_sq is not None:
        clip = torch.clamp(max_grad_norm / (grad_norm_sq.sqrt() + 1e-6), max=1.0)
    g = g * clip
    p.mul_(1 - lr * weight_decay)
    m.add_((1 - beta1) * (g - m))
    v.mul_(beta2).add_((1 - beta2) * g * g)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p.sub_((lr / bc1) * (m / (v.sqrt() / (bc2 ** 0.5) + eps)))


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq=None, max_grad_norm=None, ema=None, ema_decay=0.0):
    _adam(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm)
    if ema is not None:
        ema.sub_((1.0 - ema_decay) * (ema - p))


def adamw_step_shadow(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm, w16, w16t, ema=None, ema_decay=0.0):
    _adam(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm)
    if ema is not None:
        ema.sub_((1.0 - ema_decay) * (ema - p))
    R, C = p.shape
    if w16 is not None:
        w16[:R, :C] = p.bfloat16()
    if w16t is not None:
        w16t[:C, :R] = p.t().bfloat16()


# ------------------------------------------------------------------------------------------------ sampler (csrc/ce.hip: udm_ddpm_sample_rows)
def ddpm_sample_rows(logits, V, Vt, mask_id, *, t=None, s=None, modality=None, restrict=False, u=None, seed=0, greedy=False, logits_u=None, w=None):
    M = logits.shape[0]
    valid = _valid(M, V, Vt, mask_id, modality, restrict)
    z = logits[:, :V].float()
    if logits_u is not None:
        z = (1 + w[:, None]) * z - w[:, None] * logits_u[:, :V].float()
    z = z.masked_fill(~valid, float("-inf"))
    logp = z - torch.logsumexp(z, -1, keepdim=True)
    if greedy:
        return torch.where(valid, logp, torch.full_like(logp, -1e6)).argmax(-1)
    q = torch.where(valid, logp.exp() * (t - s)[:, None], torch.zeros_like(logp))
    q[:, mask_id] = s
    if u is None:
        u = torch.rand(M, V, generator=torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF))
    return (q / (1e-10 - (u[:, :V] + 1e-10).log())).argmax(-1)


def categorical_sample_rows(logits, V, Vt, mask_id, *, modality=None, restrict=False, u=None, seed=0, given=None, logits_u=None, w=None):
    M = logits.shape[0]
    valid = _valid(M, V, Vt, mask_id, modality, restrict)
    z = logits[:, :V].float()
    if logits_u is not None:
        z = (1 + w[:, None]) * z - w[:, None] * logits_u[:, :V].float()
    z = z.masked_fill(~valid, float("-inf"))
    logp = z - torch.logsumexp(z, -1, keepdim=True)
    if given is not None:
        tok = given
    else:
        if u is None:
            u = torch.rand(M, V, generator=torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF))
        tok = (logp.exp() / (1e-10 - (u[:, :V] + 1e-10).log())).argmax(-1)
    return tok, logp.gather(-1, tok[:, None]).squeeze(-1)


# ------------------------------------------------------------------------------------------------ token data path (csrc/tokens.hip)
def sample_t_noise(u, *, antithetic, sampling_eps, noise_eps):
    """CPU double of udm_sample_t_noise: the reference's statements (model.py:589-619, models/noise_schedule.py:128-157)"""
    n = u.numel()
    e = u
    if antithetic:
        e = (e / n + torch.arange(n, device=u.device) / n) % 1
    t = ((1 - sampling_eps) * e + sampling_eps).to(torch.float32)
    sigma = -torch.log1p(-(1 - noise_eps) * t)
    dsigma = (1 - noise_eps) / (1 - (1 - noise_eps) * t)
    return t, sigma, dsigma, 1 - torch.exp(-sigma)


def qxt_absorbing(x, r_move, move_chance, mask_id, *, r_txt=None, r_img=None, p_txt=0.0, p_img=0.0, modality_mask=None):
    """CPU double of udm_qxt_absorbing: the reference's statements (model.py:424-587, multimodal non-interleaved branch)"""
    move = r_move < move_chance.reshape(-1, 1)
    txt = img = ign = None
    if r_txt is not None or r_img is not None:
        txt = (r_txt < p_txt) if r_txt is not None else torch.zeros((x.shape[0], 1), dtype=torch.bool, device=x.device)
        img = (r_img < p_img) if r_img is not None else torch.zeros((x.shape[0], 1), dtype=torch.bool, device=x.device)
        both = txt & img
        txt = torch.where(both, False, txt)
        img = torch.where(both, False, img)
        move = torch.where(txt, modality_mask[..., 0], move)
        move = torch.where(img, modality_mask[..., 1], move)
        ign = img | txt
    return torch.where(move, mask_id, x), move, txt, img, ign


def assemble_joint_tokens(txt, txt_mask, img, Vt, idx=None):
    if idx is not None:
        txt, img = txt[idx], img[idx]
        txt_mask = None if txt_mask is None else txt_mask[idx]
    ids = torch.cat([txt.to(torch.int64), img.to(torch.int64) + int(Vt)], -1)
    tm = torch.ones_like(txt, dtype=torch.bool) if txt_mask is None else txt_mask.to(torch.bool)
    mask = torch.cat([tm, torch.ones_like(img, dtype=torch.bool)], -1)
    modality = torch.cat([torch.zeros_like(txt, dtype=torch.int64), torch.ones_like(img, dtype=torch.int64)], -1)
    return ids, mask, modality
