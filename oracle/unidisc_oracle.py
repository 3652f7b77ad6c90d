"""CPU oracle: a plain-PyTorch restatement of UniDisc's denoising hot path.

TEST INFRASTRUCTURE — not product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker.  The product
(``unidisc_amd``) never routes through it and fails loudly when its HIP library is missing.

Parity status: PINNED against golden vectors generated from the imported reference
(``oracle/make_golden.py`` → ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks fp32
equality to ≤1e-5 and bit-exact integer/boolean quantities).  The one exception is the Lumina
2-D RoPE *table generator* (``oracle/cases.py::lumina_rope_2d``), which restates an un-vendored
third-party function (diffusers 0.32.2) and is "parity unpinned"; the tables are inputs here.

Every function cites the reference file:line (relative to /root/reference) it follows.
All tensors are fp32 unless ``bf16=True`` is requested, in which case the rounding points of the
reference's CUDA-autocast flow are emulated with explicit casts.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn.functional as F

NEG_INF = -1000000.0  # model_setup.py:269


# ----------------------------------------------------------------------------------------------
# configuration (only the keys the hot path reads)
# ----------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    hidden_size: int
    n_heads: int
    cond_dim: int
    n_blocks: int
    txt_length: int
    img_length: int
    vocab_size: int
    text_vocab_size: int
    norm_type: str = "rms"
    qk_norm: bool = True
    sandwich_normalization: bool = True
    modality_embed: bool = True
    rope_2d: bool = False
    linear_factor: float = 1.0
    time_conditioning: bool = False
    multimodal_batches: bool = True
    force_argmax_valid_indices: bool = True
    dropout: float = 0.0
    # trainer-level
    mask_entire_modality: Optional[float] = None
    mask_txt_only: bool = False
    softmin_snr: Optional[float] = None
    text_loss_weight: Optional[float] = None
    img_loss_weight: Optional[float] = None
    force_full_attention_mask_loss_only: bool = False
    force_full_attention_mask: bool = False
    set_max_txt_loss_ratio: Optional[float] = None
    antithetic_sampling: bool = True
    sampling_eps: float = 1e-3
    flex_attention_txt_masking_prob: Optional[float] = None   # model.flex_attention_{txt,img}_masking_prob: modality attention dropout (model.py:863-878)
    flex_attention_img_masking_prob: Optional[float] = None
    interleaved: bool = False          # trainer.interleaved + data.require_sample_ids + interleaved_training_flex_attention
    extra: dict = field(default_factory=dict)

    @property
    def length(self):
        return self.txt_length + self.img_length

    @property
    def mask_index(self):
        return self.text_vocab_size - 1  # asserted model.py:569

    @property
    def head_dim(self):
        return self.hidden_size // self.n_heads

    @classmethod
    def from_case(cls, case: dict):
        names = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in case.items() if k in names and v is not None}
        return cls(**kw)


def _r(x, bf16):
    """Rounding point: value is materialised as bf16 in the reference's autocast flow."""
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


# ----------------------------------------------------------------------------------------------
# small ops
# ----------------------------------------------------------------------------------------------
def rms_norm(x, w, eps=1e-6, bf16=False):
    """models/dit.py:95-100 — ``_norm(x.float()).type_as(x) * weight``."""
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return y * w


def rms_norm_lowp_input(x_bf16val, w, eps=1e-6, bf16=False):
    """RMSNorm applied to a bf16 tensor (sandwich norms): normalised value is cast back to bf16
    (``.type_as(x)``) *before* the fp32 weight multiply (dit.py:98-100; A5b items 6 and 8)."""
    y = x_bf16val * torch.rsqrt(x_bf16val.pow(2).mean(-1, keepdim=True) + eps)
    return _r(y, bf16) * w


def layer_norm_nobias(x, w):
    """models/dit.py:383-403 — F.layer_norm(x.float(), [dim]) * weight (eps 1e-5, no bias)."""
    return F.layer_norm(x, [x.shape[-1]]) * w


def get_norm(cfg: OracleConfig):
    if cfg.norm_type == "rms":
        return lambda x, w: rms_norm(x, w)
    return layer_norm_nobias


def linear(x, w, b=None, bf16=False):
    """nn.Linear under autocast: bf16 operands, fp32 accumulate, bf16 result (A5b)."""
    if bf16:
        y = F.linear(_r(x, True), _r(w, True), None)
        if b is not None:
            y = y + _r(b, True)
        return _r(y, True)
    return F.linear(x, w, b)


def gelu_tanh(x):
    """nn.GELU(approximate='tanh') — models/dit.py:918."""
    return F.gelu(x, approximate="tanh")


def rotary_table_1d(seq_len, dim, base=10000.0):
    """models/dit.py:307-330 ``Rotary`` + buffer slicing :1226-1239 → cos/sin fp32 [L, dim/2]."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(seq_len).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return freqs.cos(), freqs.sin()


def apply_rotary(x, cos, sin):
    """models/standalone_rotary.py:14-31 (non-interleaved / NeoX half rotation).

    x: [B, L, H', D]; cos/sin: [L, D/2] or [B, L, D/2].
    """
    cos = torch.cat([cos, cos], -1).unsqueeze(-2)
    sin = torch.cat([sin, sin], -1).unsqueeze(-2)
    x1, x2 = x.chunk(2, dim=-1)
    rot = torch.cat((-x2, x1), dim=-1)
    return x * cos + rot * sin


def timestep_embedding(t, dim=256, max_period=10000):
    """models/dit.py:428-444."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def modulate(x, shift, scale, modality):
    """models/dit.py:263-304 ``modulate_fused``: image-only when a modality map with any image token is given."""
    if modality is not None and bool(modality.any()):
        return torch.where(modality.unsqueeze(-1) == 1, x * (1 + scale) + shift, x)
    return x * (1 + scale) + shift


def bias_dropout_add_scale(x, scale, residual, modality):
    """models/dit.py:229-253 with bias=None, dropout p=0 (parity runs use p=0).

    With a modality map, text tokens receive the raw branch (no gate), image tokens the gated branch.
    """
    out = x
    if scale is not None:
        out = scale * out
    if modality is not None:
        out = torch.where((modality == 1).unsqueeze(-1), out, x)
    return residual + out


def modality_dropout_mask(txt_drop, img_drop, txt_length, L):
    """model_utils.py:721-731 `_attn_mask`: allowed[b, q, kv] for the modality attention dropout of model.py:863-878."""
    q = torch.arange(L)[None, :, None]
    kv = torch.arange(L)[None, None, :]
    td, idr = txt_drop.reshape(-1, 1, 1).bool(), img_drop.reshape(-1, 1, 1).bool()
    txt_case = ~td | (((q < txt_length) & (kv < txt_length)) | (q >= txt_length))
    img_case = ~idr | (((q >= txt_length) & (kv >= txt_length)) | (q < txt_length))
    return txt_case & img_case


_FLASH_ROUNDING = [False]   # set by compute_loss(bf16=True, flash_rounding=True) around its forward


class _FlashRoundingAttention(torch.autograd.Function):
    """The rounding points INSIDE the reference's attention kernel (flash_attn 2.x `flash_attn_qkvpacked_func`, dit.py:843; the library is a third-party wheel,
    absent here - restated from its published algorithm, FlashAttention-2, Dao 2023, Algorithms 1-2): with bf16 inputs the probabilities are cast to bf16
    before P V (forward) and before dV = P^T dO (backward), dS = P (dP - delta) is cast to bf16 before dQ = dS K and dK = dS^T Q, and
    delta = rowsum(dO * O) uses the STORED bf16 output; the softmax statistics and all accumulations are fp32.  Used only by the bf16 noise-floor comparator
    (`compute_loss(bf16=True, flash_rounding=True)`): tensor-boundary rounding alone (autograd through `_r`) runs this backward in fp32 and understates the
    noise the reference's own gradients carry in dq / dk - and therefore in the qk-norm vectors, whose gradients are column sums of them."""

    @staticmethod
    def forward(ctx, q, k, v, neg_mask):
        rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
        scale = 1.0 / math.sqrt(q.shape[-1])
        s = (q @ k.transpose(-1, -2)) * scale
        if neg_mask is not None:
            s = s.masked_fill(neg_mask, float("-inf"))
        lse = torch.logsumexp(s, -1, keepdim=True)
        lse = torch.where(torch.isfinite(lse), lse, torch.zeros_like(lse))
        p = torch.exp(s - lse)
        o = rb(p) @ v
        ctx.save_for_backward(q, k, v, lse, rb(o), neg_mask if neg_mask is not None else torch.zeros(0, dtype=torch.bool))
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
        q, k, v, lse, o16, neg_mask = ctx.saved_tensors
        do = rb(do)
        s = (q @ k.transpose(-1, -2)) * ctx.scale
        if neg_mask.numel():
            s = s.masked_fill(neg_mask, float("-inf"))
        p = torch.exp(s - lse)
        dv = rb(p).transpose(-1, -2) @ do
        dp = do @ v.transpose(-1, -2)
        delta = (do * o16).sum(-1, keepdim=True)
        ds = rb(p * (dp - delta))
        return ds @ k * ctx.scale, ds.transpose(-1, -2) @ q * ctx.scale, dv, None


def attention_core(q, k, v, sample_ids=None, allow_mask=None, flash_rounding=False):
    """softmax(q kᵀ/√D) v, bidirectional (dit.py:826-829 SDPA ≡ :843 FA2); optional document mask
    ``sid[q]==sid[kv] & sid[q]!=-1`` (model_utils.py:740-771) or a dense allowed[b, q, kv] mask (FlexAttention block_mask, dit.py:784-812).
    q,k,v: [B, L, H, D] → [B, L, H*D].  flash_rounding: see _FlashRoundingAttention (bf16 comparator only)."""
    B, L, H, D = q.shape
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    neg = None
    if allow_mask is not None:
        neg = ~allow_mask[:, None]
    if sample_ids is not None:
        sid = sample_ids.clone()
        allpad = (sid == -1).all(-1)
        sid[allpad, 0] = 0
        allow = (sid[:, :, None] == sid[:, None, :]) & (sid[:, :, None] != -1)
        neg = ~allow[:, None] if neg is None else (neg | ~allow[:, None])
    if flash_rounding:
        o = _FlashRoundingAttention.apply(q, k, v, neg.expand(B, H, L, L) if neg is not None else None)
        return o.transpose(1, 2).reshape(B, L, H * D)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(D)
    if neg is not None:
        s = s.masked_fill(neg, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    o = p @ v
    return o.transpose(1, 2).reshape(B, L, H * D)


# ----------------------------------------------------------------------------------------------
# backbone (models/dit.py:1324-1500 with production flags; A5 of SURVEY.md)
# ----------------------------------------------------------------------------------------------
def select_rotary(cfg: OracleConfig, buffers: Dict[str, torch.Tensor], modality, L):
    """models/dit.py:1413-1460 (non-interleaved branches)."""
    if cfg.modality_embed and cfg.rope_2d and cfg.multimodal_batches:
        ct, st = buffers["rotary_cos_emb_txt"], buffers["rotary_sin_emb_txt"]
        ci, si = buffers["rotary_cos_emb_img"], buffers["rotary_sin_emb_img"]
        if modality.shape[-1] != cfg.img_length:
            pad = max(modality.shape[-1] - cfg.img_length, 0)
            ci = torch.cat([torch.full((pad, ci.shape[-1]), float("nan")), ci], 0)
            si = torch.cat([torch.full((pad, si.shape[-1]), float("nan")), si], 0)
        sel = modality[:, :, None] == 0
        return torch.where(sel, ct[None, :L], ci[None, :L]), torch.where(sel, st[None, :L], si[None, :L])
    return buffers["rotary_cos_emb"][:L], buffers["rotary_sin_emb"][:L]


def interleaved_indices(modality_mask):
    """unidisc/utils/tensor_utils.py:4-22: maximal runs of image tokens per row -> (batch index, start, end) in row-major order."""
    out = []
    for b in range(modality_mask.shape[0]):
        row, start = modality_mask[b].tolist(), None
        for i, v in enumerate(row + [False]):
            if v and start is None:
                start = i
            elif not v and start is not None:
                out.append((b, start, i))
                start = None
    return out


def contiguous_blocks(ids, extra=None):
    """tensor_utils.py:24-44 (and :46-69 with `extra` = modality): maximal runs of equal sample id (and equal modality) with id >= 0."""
    out = []
    for b in range(ids.shape[0]):
        row = ids[b].tolist()
        ex = extra[b].tolist() if extra is not None else [0] * len(row)
        start = 0
        for i in range(1, len(row) + 1):
            if i == len(row) or row[i] != row[start] or ex[i] != ex[start]:
                if row[start] >= 0:
                    out.append((b, start, i))
                start = i
    return out


def interleaved_rotary(cfg: OracleConfig, P, buffers, x, modality, sample_ids):
    """models/dit.py:1421-1444 with :122-191.  cos = sin = 0 [B, L, D/2]; every image run whose length is one of the supported block sizes gets
    that size's 2-D table and `img_count_embedding[j]` added to its embeddings (j = number of earlier image runs of the same row that start in
    the same packed sample); then inside every run of one sample id the TEXT positions get the 1-D table indexed from the start of the run.
    Image runs of any other length keep cos = sin = 0.  Returns (x, cos, sin)."""
    B, L = modality.shape
    D2 = cfg.head_dim // 2
    cos, sin = torch.zeros(B, L, D2), torch.zeros(B, L, D2)
    is_img = modality.bool()
    runs = interleaved_indices(is_img)
    x = x.clone()
    for i, (b, s, e) in enumerate(runs):
        n = e - s
        j = sum(1 for (b2, s2, _) in runs[:i] if b2 == b and int(sample_ids[b2, s2]) == int(sample_ids[b, s]))
        if n in (256, 1024, 2304, 4096):
            x[b, s:e] = x[b, s:e] + P["img_count_embedding"][j]
            cos[b, s:e] = buffers[f"rotary_cos_emb_img_{n}"][:n]
            sin[b, s:e] = buffers[f"rotary_sin_emb_img_{n}"][:n]
    ct, st = buffers["rotary_cos_emb_txt"], buffers["rotary_sin_emb_txt"]
    for (b, s, e) in contiguous_blocks(sample_ids):
        txt = ~is_img[b, s:e, None]
        cos[b, s:e] = torch.where(txt, ct[: e - s], cos[b, s:e])
        sin[b, s:e] = torch.where(txt, st[: e - s], sin[b, s:e])
    return x, cos, sin


def make_buffers(cfg: OracleConfig, lumina_fn=None):
    """Non-persistent rotary buffers registered by DIT.__init__ (models/dit.py:1203-1239)."""
    D = cfg.head_dim
    out = {}
    if cfg.rope_2d:
        if cfg.interleaved:  # dit.py:1209-1213: one 2-D table per supported image block size (tokens, linear factor)
            for n_img, lf in ((256, 1), (1024, 2), (2304, 3), (4096, 4)):
                side = int(math.sqrt(n_img))
                emb = lumina_fn(D, side, side, linear_factor=lf, ntk_factor=1.0)
                out[f"rotary_cos_emb_img_{n_img}"] = emb.flatten(0, 1).real.contiguous()
                out[f"rotary_sin_emb_img_{n_img}"] = emb.flatten(0, 1).imag.contiguous()
        else:
            side = int(math.sqrt(cfg.img_length))
            emb = lumina_fn(D, side, side, linear_factor=cfg.linear_factor, ntk_factor=1.0)
            out["rotary_cos_emb_img"] = emb.flatten(0, 1).real.contiguous()
            out["rotary_sin_emb_img"] = emb.flatten(0, 1).imag.contiguous()
        n = cfg.length if cfg.multimodal_batches else cfg.txt_length
        c, s = rotary_table_1d(n, D)
        out["rotary_cos_emb_txt"], out["rotary_sin_emb_txt"] = c, s
    else:
        c, s = rotary_table_1d(cfg.length, D)
        out["rotary_cos_emb"], out["rotary_sin_emb"] = c, s
    return out


def dit_block(cfg, P, pre, x, cos, sin, c, modality, sample_ids, bf16, allow_mask=None):
    """models/dit.py:948-1033 ``DDiTBlock.forward`` + :616-887 ``Attention.forward`` (SDPA branch)."""
    B, L, d = x.shape
    H, D = cfg.n_heads, cfg.head_dim
    norm = get_norm(cfg)
    tc = cfg.time_conditioning
    mod_map = modality if tc else None  # dit.py:884,1012
    if tc:
        ada = linear(c, P[pre + "adaLN_modulation.weight"], P[pre + "adaLN_modulation.bias"], bf16)[:, None, :]
        sh1, sc1, g1, sh2, sc2, g2 = ada.chunk(6, dim=2)
    x_skip = x
    h = norm(x, P[pre + "norm1.weight"])
    if tc:
        h = modulate(h, sh1, sc1, modality)
    qkv = linear(h, P[pre + "attention.attn_qkv.weight"], None, bf16)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    if cfg.qk_norm:  # dit.py:680-682: LayerNorm over the full hidden dim, written back in place (bf16)
        q = _r(F.layer_norm(q, [d], P[pre + "attention.q_norm.weight"], P[pre + "attention.q_norm.bias"], 1e-5), bf16)
        k = _r(F.layer_norm(k, [d], P[pre + "attention.k_norm.weight"], P[pre + "attention.k_norm.bias"], 1e-5), bf16)
    qk = torch.stack([q, k], 2).reshape(B, L, 2 * H, D)  # "b s (three h d)" → q heads then k heads
    qk = _r(apply_rotary(qk, cos, sin), bf16)  # dit.py:723-726
    q, k = qk[:, :, :H], qk[:, :, H:]
    a = _r(attention_core(q, k, v.reshape(B, L, H, D), sample_ids, allow_mask, flash_rounding=bool(bf16 and _FLASH_ROUNDING[0])), bf16)
    a = linear(a, P[pre + "attention.attn_out.weight"], None, bf16)
    if cfg.sandwich_normalization:  # dit.py:993-994 (gate_msa unused, :983)
        x = x_skip + (rms_norm_lowp_input(a, P[pre + "pre_residual_norm.weight"], bf16=bf16) if cfg.norm_type == "rms"
                      else norm(a, P[pre + "pre_residual_norm.weight"]))
    else:  # dit.py:877-885.  Attention.time_conditioning is never set by DDiTBlock (ctor default False, dit.py:533;
        # DIT passes the flag to the block only, :1267-1290), so the attention branch is gated on ALL tokens.
        x = bias_dropout_add_scale(a, g1 if tc else None, x_skip, None)
    h = norm(x, P[pre + "norm2.weight"])
    if tc:
        h = modulate(h, sh2, sc2, modality)
    u = linear(h, P[pre + "mlp.0.weight"], P[pre + "mlp.0.bias"], bf16)
    u = _r(gelu_tanh(u), bf16)
    u = linear(u, P[pre + "mlp.2.weight"], P[pre + "mlp.2.bias"], bf16)
    if cfg.sandwich_normalization:
        u = (rms_norm_lowp_input(u, P[pre + "post_ff_norm.weight"], bf16=bf16) if cfg.norm_type == "rms"
             else norm(u, P[pre + "post_ff_norm.weight"]))
    return bias_dropout_add_scale(u, g2 if tc else None, x, mod_map)  # dit.py:1015-1031


def dit_forward(cfg: OracleConfig, P: Dict[str, torch.Tensor], buffers, indices, sigma=None, modality=None,
                sample_ids=None, bf16=False, return_hidden=False, allow_mask=None):
    """models/dit.py:1324-1500 ``DIT.forward`` → logits [B, L, V]."""
    B, L = indices.shape
    x = P["vocab_embed.embedding"][indices]  # :1375
    c = None
    if cfg.time_conditioning:  # :1377-1379
        te = timestep_embedding(sigma, 256)
        hdn = F.silu(linear(te, P["sigma_map.mlp.0.weight"], P["sigma_map.mlp.0.bias"], bf16))
        c = F.silu(linear(hdn, P["sigma_map.mlp.2.weight"], P["sigma_map.mlp.2.bias"], bf16))
    if cfg.modality_embed:  # :1402-1411
        Em = P["modality_embed.embedding"]
        if cfg.multimodal_batches:
            x = x + torch.where((modality == 0).unsqueeze(-1), Em[0][None, None], Em[1][None, None])
        else:
            x = torch.cat([x[:, :cfg.txt_length] + Em[0], x[:, cfg.txt_length:] + Em[1]], 1)
    if cfg.interleaved:
        x, cos, sin = interleaved_rotary(cfg, P, buffers, x, modality, sample_ids)
    else:
        cos, sin = select_rotary(cfg, buffers, modality, L)
    for i in range(cfg.n_blocks):
        x = dit_block(cfg, P, f"blocks.{i}.", x, cos, sin, c, modality, sample_ids, bf16, allow_mask)
    norm = get_norm(cfg)
    h = norm(x, P["output_layer.norm_final.weight"])  # :1083-1092
    if cfg.time_conditioning:
        ada = linear(c, P["output_layer.adaLN_modulation.weight"], P["output_layer.adaLN_modulation.bias"], bf16)[:, None, :]
        sh, sc = ada.chunk(2, dim=2)
        h = modulate(h, sh, sc, modality)
    logits = linear(h, P["output_layer.linear.weight"], P["output_layer.linear.bias"], bf16)
    return (logits, x) if return_hidden else logits


# ----------------------------------------------------------------------------------------------
# trainer-level pieces (model.py)
# ----------------------------------------------------------------------------------------------
def update_batch(cfg: OracleConfig, batch: dict):
    """model.py:183-212, 296-348 — token-dataset branch."""
    b = dict(batch)
    if "img_input_ids" in b:
        img = b.pop("img_input_ids").to(torch.int64)
        ids, am = img, torch.ones_like(img).to(torch.bool)
        if "txt_input_ids" in b:
            txt = b["txt_input_ids"].to(torch.int64)
            ids = torch.cat([txt, ids + cfg.text_vocab_size], -1)
            am = torch.cat([b["txt_attention_mask"], am], -1)
        b["input_ids"] = ids.to(torch.int64)
        b["attention_mask"] = am
        if "modality" not in b:
            mod = torch.zeros_like(ids)
            mod[:, -img.shape[-1]:] = 1
            b["modality"] = mod
    if "modality" in b:
        b["modality"] = b["modality"].to(torch.int64)
        b["modality"][b["modality"] == -1] = 0
        b["modality_mask"] = F.one_hot(b["modality"], num_classes=2).to(torch.bool)
        b["batch_contains_img"] = (b["modality"] == 1).any(-1)
        b["txt_sl"] = b["modality_mask"][..., 0]
        b["img_sl"] = b["modality_mask"][..., 1]
    if cfg.force_full_attention_mask:
        b["attention_mask"] = torch.ones_like(b["attention_mask"], dtype=torch.bool)
    if getattr(cfg, "interleaved", False):  # model.py:350-353 (+ :375-376 default ids)
        b["attention_mask"] = b["attention_mask"].to(torch.bool).clone()
        if "sample_ids" not in b:
            b["sample_ids"] = torch.zeros_like(b["modality"])
        b["sample_ids"] = b["sample_ids"].to(torch.int64).clone()
        b["sample_ids"][~b["attention_mask"]] = -1
        b["attention_mask"][b["sample_ids"] == -1] = False
    b["attention_mask"] = b["attention_mask"].to(torch.bool)
    return b


def sample_t(cfg: OracleConfig, n, generator=None):
    """model.py:589-619."""
    u = torch.rand(n, generator=generator)
    if cfg.antithetic_sampling:
        u = (u / n + torch.arange(n) / n) % 1
    return ((1 - cfg.sampling_eps) * u + cfg.sampling_eps).to(torch.float32)


def loglinear_noise(t, eps=1e-3):
    """models/noise_schedule.py:142-150 → (total σ, rate σ')."""
    return -torch.log1p(-(1 - eps) * t), (1 - eps) / (1 - (1 - eps) * t)


def q_xt(cfg: OracleConfig, x0, move_chance, batch, training=True, generator=None):
    """model.py:439, 468-539 (non-interleaved), 579."""
    B, L = x0.shape
    move = torch.rand(B, L, generator=generator) < move_chance
    ignore = smt = smi = None
    pm = cfg.mask_entire_modality
    if pm is not None and training:
        if cfg.mask_txt_only:
            smt = torch.rand(B, 1, generator=generator) < pm
            smi = torch.zeros_like(smt)
        else:
            smt = torch.rand(B, 1, generator=generator) < pm / 2
            smi = torch.rand(B, 1, generator=generator) < pm / 2
        if not (cfg.multimodal_batches and cfg.interleaved):
            both = smt & smi
            smt, smi = smt & ~both, smi & ~both
        if cfg.multimodal_batches and cfg.interleaved:
            # model.py:483-522: per (modality, sample) block of more than 4 tokens, masked as a whole with probability
            # 2 pm (k + 1) / n, k = index of the block inside its packed sample, n = number of such blocks in that sample
            blocks = [blk for blk in contiguous_blocks(batch["sample_ids"], batch["modality"]) if blk[2] - blk[1] > 4]
            sid_of = [int(batch["sample_ids"][b, s]) for (b, s, _) in blocks]
            r = torch.rand(len(blocks), 1, generator=generator)
            ignore = torch.zeros(B, dtype=torch.bool)
            for i, (b, s, e) in enumerate(blocks):
                k = sum(1 for i2 in range(i) if blocks[i2][0] == b and sid_of[i2] == sid_of[i])
                n = sum(1 for i2 in range(len(blocks)) if blocks[i2][0] == b and sid_of[i2] == sid_of[i])
                if bool(r[i, 0] < pm * ((k + 1) / n) * 2):
                    move[b, s:e] = True
                    ignore[b] = True
        elif cfg.multimodal_batches:
            move = torch.where(smt, batch["modality_mask"][..., 0], move)
            move = torch.where(smi, batch["modality_mask"][..., 1], move)
        else:
            smi = smi & ~batch["txt_sl"].all(-1, keepdim=True)
            move[:, :cfg.txt_length] |= smt
            move[:, L - cfg.img_length:] |= smi
        if not (cfg.multimodal_batches and cfg.interleaved):
            ignore = smi | smt
    xt = torch.where(move, cfg.mask_index, x0)
    return xt, ignore, smt, smi, move


def subs_parameterization(cfg: OracleConfig, logits, xt, modality=None, batch=None, bf16=False):
    """model.py:621-658 (training: not allow_slicing)."""
    z = logits.clone()
    m, Vt = cfg.mask_index, cfg.text_vocab_size
    z[..., m] = _r(z[..., m] + NEG_INF, bf16)
    if cfg.force_argmax_valid_indices:
        if cfg.multimodal_batches:
            txt = batch["txt_sl"] if modality is None else modality == 0
            img = batch["img_sl"] if modality is None else modality == 1
            z[..., Vt:] = torch.where(txt[..., None], NEG_INF, z[..., Vt:])
            z[..., :Vt] = torch.where(img[..., None], NEG_INF, z[..., :Vt])
        else:
            z[:, :cfg.txt_length, Vt:] = NEG_INF
            z[:, z.shape[1] - cfg.img_length:, :Vt] = NEG_INF
    z = _r(z, bf16)
    z = _r(z - _r(torch.logsumexp(z, -1, keepdim=True), bf16), bf16)
    if xt is None:  # model.py:645: the carry-over of unmasked tokens is skipped (CFG branch of `_ddpm_forward`)
        return z
    unmasked = xt != m
    z = torch.where(unmasked[..., None], torch.full_like(z, NEG_INF), z)
    onehot = torch.arange(z.shape[-1]) == xt[..., None]
    z = torch.where(unmasked[..., None] & onehot, torch.zeros_like(z), z)
    return _r(z, bf16)


@dataclass
class OracleLoss:
    loss: torch.Tensor
    img_loss: object = 0
    txt_loss: object = 0
    nlls: torch.Tensor = None
    token_mask: torch.Tensor = None
    txt_nlls: object = 0
    img_nlls: object = 0
    extra_losses: dict = None
    modality_mask: torch.Tensor = None
    aux: dict = None


def reduce_loss(cfg: OracleConfig, log_p, sigma, dsigma, attention_mask, modality_mask, ignore_batch):
    """model.py:967-1161 — weighting, reduction, Loss record."""
    std_w = (dsigma / torch.expm1(sigma))[:, None]
    loss = -log_p * std_w
    if cfg.softmin_snr is not None:
        loss = -log_p * (dsigma / (torch.expm1(sigma) + 1 / cfg.softmin_snr))[:, None]
    std_loss = (-log_p * std_w).detach()
    rec = dict(extra_losses={})
    if modality_mask is not None:
        rec["txt_nlls"] = std_loss * modality_mask[..., 0] * attention_mask
        rec["img_nlls"] = std_loss * modality_mask[..., 1] * attention_mask
    weighted = cfg.text_loss_weight is not None and cfg.img_loss_weight is not None
    if cfg.multimodal_batches or weighted:
        tm = modality_mask[..., 0] & attention_mask
        im = modality_mask[..., 1] & attention_mask
        tc, ic = tm.sum(), im.sum()
        tot = tc + ic
        tf, imf = tc / tot, ic / tot
        rec["extra_losses"]["trainer/img_frac"] = imf
        rec["extra_losses"]["trainer/txt_frac"] = tf
        rec["extra_losses"]["trainer/attention_mask_valid_frac"] = attention_mask.sum() / attention_mask.numel()
        if ignore_batch is not None:
            ib = ignore_batch.squeeze(-1)
            rec["extra_losses"]["trainer/ignore_batch_metrics_frac"] = ib.sum() / ib.numel()
    if weighted:
        loss = loss * attention_mask
        txt_loss = (loss[tm].sum() / tc) * tf * cfg.text_loss_weight
        img_loss = (loss[im].sum() / ic) * imf * cfg.img_loss_weight
        r = cfg.set_max_txt_loss_ratio
        if r is not None and not (torch.isnan(img_loss).any() or torch.isnan(txt_loss).any()):
            scale = torch.minimum(torch.tensor(1.0), (r * img_loss.detach()) / (txt_loss.detach() + 1e-8))
            txt_loss = txt_loss * scale
        txt_loss = torch.nan_to_num(txt_loss, nan=0.0)
        img_loss = torch.nan_to_num(img_loss, nan=0.0)
        total = txt_loss + img_loss
        rec.update(txt_loss=txt_loss.detach().clone(), img_loss=img_loss.detach().clone())
    else:
        am = torch.ones_like(attention_mask) if cfg.force_full_attention_mask_loss_only else attention_mask
        total = torch.nan_to_num((loss * am).sum() / am.sum(), nan=0.0)
    nlls = std_loss * attention_mask
    token_mask = attention_mask
    if ignore_batch is not None:
        token_mask = torch.where(ignore_batch.squeeze(-1)[:, None].repeat(1, attention_mask.shape[-1]),
                                 torch.full_like(attention_mask, False), attention_mask)
    return OracleLoss(loss=total, img_loss=rec.get("img_loss", 0), txt_loss=rec.get("txt_loss", 0), nlls=nlls,
                      token_mask=token_mask, txt_nlls=rec.get("txt_nlls", 0), img_nlls=rec.get("img_nlls", 0),
                      extra_losses=rec["extra_losses"], modality_mask=modality_mask)


def compute_loss(cfg: OracleConfig, P, buffers, batch, generator=None, bf16=False, training=True, flash_rounding=False):
    """model.py:797-1173, SUBS / continuous-time / absorbing branch.  ``batch`` must come from update_batch."""
    x0, am = batch["input_ids"], batch["attention_mask"]
    modality_mask = batch.get("modality_mask")
    t = sample_t(cfg, x0.shape[0], generator)
    sigma, dsigma = loglinear_noise(t)
    move_chance = 1 - torch.exp(-sigma[:, None])
    xt, ignore, smt, smi, move = q_xt(cfg, x0, move_chance, batch, training, generator)
    modality = batch["modality"] if cfg.multimodal_batches else None
    allow_mask = None
    if (cfg.flex_attention_txt_masking_prob is not None or cfg.flex_attention_img_masking_prob is not None) and training:   # model.py:863-875
        B = x0.shape[0]
        txt_drop = torch.rand(B, generator=generator) < cfg.flex_attention_txt_masking_prob
        img_drop = torch.rand(B, generator=generator) < cfg.flex_attention_img_masking_prob
        if smt is not None:   # a modality that is masked out entirely must not be left seeing only itself
            txt_drop = txt_drop & ~smt.squeeze(-1)
            img_drop = img_drop & ~smi.squeeze(-1)
        allow_mask = modality_dropout_mask(txt_drop, img_drop, cfg.txt_length, x0.shape[1])
        ignore = (txt_drop | img_drop).unsqueeze(-1) if ignore is None else (ignore | (txt_drop | img_drop).unsqueeze(-1))
    _FLASH_ROUNDING[0] = bool(flash_rounding)
    try:
        logits = dit_forward(cfg, P, buffers, xt, sigma, modality, batch["sample_ids"] if cfg.interleaved else None, bf16, allow_mask=allow_mask)
    finally:
        _FLASH_ROUNDING[0] = False
    lp = subs_parameterization(cfg, logits, xt, modality, batch, bf16).float()
    log_p = torch.gather(lp, -1, x0[:, :, None]).squeeze(-1)
    out = reduce_loss(cfg, log_p, sigma, dsigma, am, modality_mask, ignore)
    out.aux = dict(t=t, sigma=sigma, dsigma=dsigma, xt=xt, move_indices=move, logits=logits, log_probs=lp, allow_mask=allow_mask,
                   ignore_batch_mask=ignore, should_mask_txt=smt, should_mask_img=smi, log_p=log_p)
    return out


# ------------------------------------------------------------------------------------------------
# sampler inner loop: `ddpm_cache` predictor, no CFG, no attention caching
# ------------------------------------------------------------------------------------------------
def sample_categorical(q, u):
    """model_utils.py:95-97 with the uniforms passed in: argmax(q / (1e-10 - log(u + 1e-10)))."""
    gumbel_norm = 1e-10 - (u + 1e-10).log()
    return (q / gumbel_norm).argmax(dim=-1)


def cfg_weight(cfg_scale, t, cfg_min_timestep=None, cfg_max_timestep=None, force_cfg_value=False):
    """model_eval.py:1737-1758 `get_cfg_weight`: w = cfg (1 - t) per sample ([B, 1]), or the window-normalised ramp when both bounds are set;
    zero outside (cfg_min_timestep, cfg_max_timestep); `force_cfg_value` uses the scalar as is; cfg = -1 sweeps linspace(0, 10, B)."""
    c = cfg_scale
    if not force_cfg_value:
        if c == -1:
            c = torch.linspace(0, 10, t.shape[0])
        if cfg_min_timestep is not None and cfg_max_timestep is not None:
            w = (c * ((t - cfg_max_timestep) / (cfg_min_timestep - cfg_max_timestep)))[:, None]
        else:
            w = (c * (1 - t))[:, None]
    else:
        w = c
    if cfg_min_timestep is not None:
        w = torch.where(t > cfg_min_timestep, w, torch.tensor(0.0))
    if cfg_max_timestep is not None:
        w = torch.where(t < cfg_max_timestep, w, torch.tensor(0.0))
    return w if isinstance(w, torch.Tensor) else torch.tensor(w)


def ddpm_forward(cfg: OracleConfig, P, buffers, x, sigma_t, modality=None, batch=None, bf16=False, x0_unmask=None, w=None, allow_mask=None):
    """model_eval.py:1761-1834: p_x0 = exp(SUBS log-probs).  With a guidance weight `w` (> 0 somewhere) and conditioning positions
    `x0_unmask`: logits = (1 + w) logits(x) - w logits(x with the conditioning masked), SUBS WITHOUT the carry-over (xt=None, :1813)."""
    logits = dit_forward(cfg, P, buffers, x, sigma_t, modality, None, bf16, allow_mask=allow_mask)
    if w is not None and x0_unmask is not None and x0_unmask.sum() > 0 and (w > 0).any():
        x_uncond = x.clone()
        x_uncond[x0_unmask] = cfg.mask_index
        logits_u = dit_forward(cfg, P, buffers, x_uncond, sigma_t, modality, None, bf16)
        ww = w.unsqueeze(-1) if (w.ndim == 2 and logits.ndim == 3) else w
        mixed = (1 + ww) * logits - ww * logits_u
        return subs_parameterization(cfg, mixed, None, modality, batch, bf16).float().exp(), (logits, logits_u)
    return subs_parameterization(cfg, logits, x, modality, batch, bf16).float().exp(), logits


def ddpm_caching_update(cfg: OracleConfig, P, buffers, x, t, dt, u, p_x0=None, modality=None, batch=None, bf16=False, x0_unmask=None, cfg_scale=None,
                        allow_mask=None):
    """model_eval.py:2073-2106.  t: [B] or [B,1]; u: uniforms [B, L, V] (what torch.rand_like drew).  Returns (p_x0, x_next, nfe)."""
    if t.ndim > 1:
        t = t.squeeze(-1)
    sigma_t, _ = loglinear_noise(t)
    move_t, move_s = t[:, None, None], (t - dt)[:, None, None]
    nfe = 0
    if p_x0 is None:
        w = cfg_weight(cfg_scale, t) if (cfg_scale is not None and x0_unmask is not None and x0_unmask.sum() > 0) else None
        p_x0, _ = ddpm_forward(cfg, P, buffers, x, sigma_t, modality, batch, bf16, x0_unmask=x0_unmask, w=w, allow_mask=allow_mask)
        nfe = 1
    q_xs = p_x0 * (move_t - move_s)
    q_xs[:, :, cfg.mask_index] = move_s[:, :, 0]
    _x = sample_categorical(q_xs, u)
    copy_flag = (x != cfg.mask_index).to(x.dtype)
    return p_x0, copy_flag * x + (1 - copy_flag) * _x, nfe


def sample_ddpm_cache(cfg: OracleConfig, P, buffers, x_init, timesteps, dt, us, x0=None, x0_unmask=None, modality=None, batch=None,
                      noise_removal=True, bf16=False, cfg_scale=None):
    """The `ddpm_cache` path of model_eval.py:2307-2444 (`_sample`): loop over timesteps[:-1], p_x0 reused while x does not change (and
    there is no time conditioning), x0 / x0_unmask conditioning re-imposed after every step, final arg-max of the log-probs."""
    x = x_init.clone()
    B = x.shape[0]
    p_cache, nfe, xs = None, 0, []
    for i in range(len(timesteps) - 1):
        t = timesteps[i] * torch.ones(B, 1)
        p_cache, x_next, n = ddpm_caching_update(cfg, P, buffers, x, t, dt, us[i], p_x0=p_cache, modality=modality, batch=batch, bf16=bf16,
                                                 x0_unmask=x0_unmask, cfg_scale=cfg_scale)
        nfe += n
        if not torch.allclose(x_next, x) or cfg.time_conditioning:
            p_cache = None
        x = x_next
        if x0 is not None:
            x = torch.where(x0_unmask, x0, x)
        xs.append(x.clone())
    x_last = x
    if noise_removal:
        t = timesteps[-1] * torch.ones(B)
        logits = dit_forward(cfg, P, buffers, x, loglinear_noise(t)[0], modality, None, bf16)
        x = subs_parameterization(cfg, logits, x, modality, batch, bf16).float().argmax(dim=-1)
        if x0 is not None:
            x = torch.where(x0_unmask, x0, x)
    return x, xs, x_last, nfe


def sample_ddpm_cache_attention_caching(cfg: OracleConfig, P, buffers, x_init, timesteps, dt, us, ratio, modality=None, batch=None, noise_removal=True, bf16=False):
    """`eval.attention_caching` on the `ddpm_cache` path (model_eval.py:2296-2366, :2425-2440), three kinds of step by i mod ratio:
    0: a full joint update (text slice written back first when the state is sliced);  1: a full update under the block mask of
    get_block_mask(img_batch_attn_dropout = all) - image queries see image keys only (the step that writes the reference's per-layer cache;
    nothing ever reads that cache back, models/dit.py:797-803 vs :812);  else: the update runs on the TEXT slice alone (x, modality and the
    p_x0 cache sliced to static_txt_sl; text queries see text keys only because nothing else is in the sequence).
    us[i]: the uniforms of step i, [B, L, V] or [B, Lt, V].  Returns (x_final, xs (x_next of every step, in that step's view), x_last, nfe, modes)."""
    x = x_init.clone()
    B, Lt = x.shape[0], cfg.txt_length
    sl = slice(0, Lt)
    p_cache, nfe, xs, modes = None, 0, [], []
    sliced, full = False, {}
    cur_mod = modality
    for i in range(len(timesteps) - 1):
        t = timesteps[i] * torch.ones(B, 1)
        allow = None
        if i % ratio == 0:
            if sliced:   # model_eval.py:2312-2323: the saved full tensors take the current text slice back
                def back(key, new):
                    if full[key] is not None:
                        full[key][:, sl] = new
                    return full[key]
                x, p_cache, cur_mod = back("x", x), back("p", p_cache), back("m", cur_mod)
                full, sliced = {}, False
            mode = "full"
        elif (i - 1) % ratio == 0:
            allow = modality_dropout_mask(torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool), Lt, x.shape[1])
            mode = "build"
        else:
            if not sliced:   # :2345-2362
                cl = lambda v: None if v is None else v.clone()
                full = dict(x=cl(x), m=cl(cur_mod), p=cl(p_cache))
                x, cur_mod = x[:, sl], (None if cur_mod is None else cur_mod[:, sl])
                p_cache = None if p_cache is None else p_cache[:, sl]
                sliced = True
            mode = "text"
        modes.append(mode)
        p_cache, x_next, n = ddpm_caching_update(cfg, P, buffers, x, t, dt, us[i], p_x0=p_cache, modality=cur_mod, batch=batch, bf16=bf16, allow_mask=allow)
        nfe += n
        if not torch.allclose(x_next, x) or cfg.time_conditioning:
            p_cache = None
        x = x_next
        xs.append(x.clone())
    if sliced:   # :2425-2440
        full["x"][:, sl] = x
        x = full["x"]
        if full["m"] is not None:
            full["m"][:, sl] = cur_mod
            cur_mod = full["m"]
    x_last = x
    if noise_removal:
        t = timesteps[-1] * torch.ones(B)
        logits = dit_forward(cfg, P, buffers, x, loglinear_noise(t)[0], cur_mod, None, bf16)
        x = subs_parameterization(cfg, logits, x, cur_mod, batch, bf16).float().argmax(dim=-1)
    return x, xs, x_last, nfe, modes


# ------------------------------------------------------------------------------------------------
# sampler inner loop: `maskgit` predictor
# ------------------------------------------------------------------------------------------------
def adap_sche(x, step, mask_index, mode="arccos"):
    """model_eval.py:2964-3001: per-sample unmasking schedule [B, step] (how many tokens each step reveals), rounded, zeros lifted to 1,
    the last step absorbs the remainder (never negative)."""
    num_masked = (x == mask_index).sum(dim=-1)
    r = torch.linspace(1, 0, step)
    val = {"root": lambda: 1 - r ** 0.5, "linear": lambda: 1 - r, "square": lambda: 1 - r ** 2, "cosine": lambda: torch.cos(r * math.pi * 0.5),
           "arccos": lambda: torch.arccos(r) / (math.pi * 0.5)}[mode]()
    out = []
    for n in num_masked:
        s = ((val / val.sum()) * n).round()
        s[s == 0] = 1
        s[-1] += n - s.sum()
        s[-1] = max(s[-1], 0)
        out.append(s.int())
    return torch.stack(out, 0)


def maskgit_update(cfg: OracleConfig, P, buffers, x, t, schedule, step, pred, gumbel, r_temp, modality=None, batch=None, bf16=False):
    """model_eval.py:3046-3114 with the two random draws passed in: `pred` [B, L] = what torch.multinomial(p_x0) returned per position,
    `gumbel` [B, L] = np.random.gumbel.  conf = log p(pred) + r_temp * gumbel * t on [MASK] positions (-inf elsewhere); every sample keeps
    its num_unmask = min(schedule[:, step], #masked) most confident predictions (threshold = k-th largest, ties included)."""
    if t.ndim > 1:
        t_col = t
        t1 = t.squeeze(-1)
    else:
        t_col, t1 = t[:, None], t
    copy_flag = x != cfg.mask_index
    num_unmask = torch.minimum(schedule[:, step].to(torch.int64), (~copy_flag).sum(-1))
    if torch.all(num_unmask <= 0):
        return x, 0
    sigma_t, _ = loglinear_noise(t1)
    p_x0, _ = ddpm_forward(cfg, P, buffers, x, sigma_t, modality, batch, bf16)
    conf = torch.gather(p_x0, -1, pred.unsqueeze(-1)).squeeze(-1).log() + r_temp * gumbel * t_col
    conf = torch.where(copy_flag, torch.full_like(conf, float("-inf")), conf)
    k = int(num_unmask.max())
    top, _ = torch.topk(conf, k=k, dim=-1)
    thr = top.gather(-1, torch.clamp(num_unmask - 1, min=0)[:, None])
    thr = torch.where((num_unmask <= 0)[:, None], torch.full_like(thr, float("inf")), thr)
    return torch.where(conf >= thr, pred, x), 1


def nucleus_filter(p, top_p=0.9, temperature=1.0):
    """model_eval.py:2642-2685 `nucleus_sampling_batch` up to its multinomial draw: the distribution it samples from, in the ORIGINAL id order.
    Reference quirks kept: the "temperature" divides PROBABILITIES (so the kept set is the largest ids whose cumulative probability stays
    <= top_p * temperature), the most likely id is always kept, the kept probabilities are renormalised."""
    sp, si = torch.sort(p / temperature, descending=True, dim=-1)
    keep = sp.cumsum(-1) <= top_p
    keep[..., 0] = True
    fp = sp * keep.float()
    fp = fp / fp.sum(-1, keepdim=True)
    return torch.zeros_like(p).scatter_(-1, si, fp)


def sample_maskgit(cfg: OracleConfig, P, buffers, x_init, timesteps, dt, preds, gumbels, r_temp, x0=None, x0_unmask=None, modality=None, batch=None,
                   noise_removal=True, bf16=False):
    """The `maskgit` path of model_eval.py:2274-2447: arccos schedule from the initial x, one update per step, final arg-max of the log-probs."""
    x = x_init.clone()
    B = x.shape[0]
    schedule = adap_sche(x, len(timesteps) - 1, cfg.mask_index, "arccos")
    nfe, xs = 0, []
    for i in range(len(timesteps) - 1):
        t = timesteps[i] * torch.ones(B, 1)
        if preds[i] is None:
            xs.append(x.clone())
            continue
        x, n = maskgit_update(cfg, P, buffers, x, t, schedule, i, preds[i], gumbels[i], r_temp, modality, batch, bf16)
        nfe += n
        xs.append(x.clone())
    x_last = x
    if noise_removal:
        t = timesteps[-1] * torch.ones(B)
        logits = dit_forward(cfg, P, buffers, x, loglinear_noise(t)[0], modality, None, bf16)
        x = subs_parameterization(cfg, logits, x, modality, batch, bf16).float().argmax(dim=-1)
    if x0 is not None:
        x = torch.where(x0_unmask, x0, x)
    return x, xs, x_last, nfe, schedule


# ------------------------------------------------------------------------------------------------
# sampler inner loop: `first_hitting` predictor
# ------------------------------------------------------------------------------------------------
def first_hitting_update(cfg: OracleConfig, P, buffers, x, t, schedule, step, u, pos_u, modality=None, batch=None, bf16=False):
    """model_eval.py:3005-3043 with the two torch.rand_like draws passed in: u [B, L, V] (token race of `_sample_categorical(p_x0)`),
    pos_u [B, L] (which [MASK] positions are revealed: the num_unmask largest values among the masked positions)."""
    t1 = t.squeeze(-1) if t.ndim > 1 else t
    sigma_t, _ = loglinear_noise(t1)
    p_x0, _ = ddpm_forward(cfg, P, buffers, x, sigma_t, modality, batch, bf16)
    copy_flag = x != cfg.mask_index
    _x = sample_categorical(p_x0, u)
    num_unmask = torch.minimum(schedule[:, step].to(torch.int64), (~copy_flag).sum(-1))
    if torch.all(num_unmask <= 0):
        return x, 1
    rv = torch.where(~copy_flag, pos_u, torch.full_like(pos_u, -1.0))
    _, indices = torch.sort(rv, dim=-1, descending=True)
    final = torch.arange(x.shape[-1]).expand(x.shape) < num_unmask[:, None]
    result = torch.zeros_like(copy_flag)
    result.scatter_(-1, indices, final)
    return torch.where(result, _x, x), 1


def sample_first_hitting(cfg: OracleConfig, P, buffers, x_init, timesteps, dt, us, pos_us, x0=None, x0_unmask=None, modality=None, batch=None,
                         noise_removal=True, bf16=False):
    x = x_init.clone()
    B = x.shape[0]
    schedule = adap_sche(x, len(timesteps) - 1, cfg.mask_index, "linear")
    nfe, xs = 0, []
    for i in range(len(timesteps) - 1):
        t = timesteps[i] * torch.ones(B, 1)
        x, n = first_hitting_update(cfg, P, buffers, x, t, schedule, i, us[i], pos_us[i], modality, batch, bf16)
        nfe += n
        xs.append(x.clone())
    x_last = x
    if noise_removal:
        t = timesteps[-1] * torch.ones(B)
        logits = dit_forward(cfg, P, buffers, x, loglinear_noise(t)[0], modality, None, bf16)
        x = subs_parameterization(cfg, logits, x, modality, batch, bf16).float().argmax(dim=-1)
    if x0 is not None:
        x = torch.where(x0_unmask, x0, x)
    return x, xs, x_last, nfe, schedule
