"""MI355X-native drop-in for the reference's ``models/dit.py::DIT``.

Same constructor signature, ``forward`` signature and ``state_dict`` schema as the reference
(models/dit.py:1095-1500), so ``config.backbone == "dit"`` can select this class in
``model_setup.init`` (model_setup.py:134-161) and HF checkpoints load unchanged.  The sub-modules below
exist to own parameters under the reference's names; compute does not go through ``nn.Module.forward``
of the children.  Instead ``DIT`` runs a hand-scheduled engine: every op of the block is a HIP kernel
from ``libunidisc_hip.so`` enqueued on the current stream, activations needed by the backward are
kept in HBM (288 GB: no recompute needed at the BASELINE shapes), and the backward is an explicit
reverse schedule that hands finished gradient ranges to an optional callback (used by
``unidisc_amd.ddp`` to overlap the bf16 gradient all-reduce with the rest of the backward).

Numerics follow the reference's bf16-autocast flow: fp32 residual stream and master
weights, bf16 GEMM operands with fp32 accumulation, bf16 activations between GEMMs, fp32 statistics.
Only bf16 compute is implemented (``trainer.precision=bf16``, the reference's training precision).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import weakref

import torch
import torch.nn as nn

from . import kernels as K
from .rope import lumina_rope_2d, rotary_table_1d

try:  # optional, for from_pretrained / push_to_hub parity with the reference class
    from huggingface_hub import PyTorchModelHubMixin as _HubMixin
except Exception:  # pragma: no cover
    class _HubMixin:  # type: ignore
        pass

BF16, F32 = torch.bfloat16, torch.float32


def cfg_get(node, key, default=None):
    """getattr/getitem with default over OmegaConf nodes, attribute bags and dicts (the reference leans on getattr(cfg, k, default))."""
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(key, default)
    try:
        v = getattr(node, key)
    except (AttributeError, KeyError):
        return default
    return v


def _ceil(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------------
# parameter containers (names == reference state_dict keys)
# ------------------------------------------------------------------------------------------------
class EmbeddingLayer(nn.Module):  # models/dit.py:1036-1043
    def __init__(self, dim, vocab_dim):
        super().__init__()
        self.embedding = nn.Parameter(torch.empty((vocab_dim, dim)))
        torch.nn.init.kaiming_uniform_(self.embedding, a=math.sqrt(5))


class RMSNorm(nn.Module):  # models/dit.py:77-100
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class LayerNorm(nn.Module):  # models/dit.py:383-403 (weight only)
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones([dim]))
        self.dim = dim


def get_norm(dim, norm_type="layernorm"):
    if norm_type == "layernorm":
        return LayerNorm(dim)
    if norm_type == "rms":
        return RMSNorm(dim)
    raise ValueError(f"Unknown norm type: {norm_type}")


class TimestepEmbedder(nn.Module):  # models/dit.py:415-449
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(), nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size


class Attention(nn.Module):  # models/dit.py:515-575
    def __init__(self, dim, n_heads, qk_norm=False):
        super().__init__()
        self.n_heads, self.head_dim, self.qk_norm = n_heads, dim // n_heads, qk_norm
        self.attn_qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.attn_out = nn.Linear(dim, dim, bias=False)
        if qk_norm:
            self.q_norm = nn.LayerNorm(dim)
            self.k_norm = nn.LayerNorm(dim)


class DDiTBlock(nn.Module):  # models/dit.py:890-934
    def __init__(self, dim, n_heads, cond_dim, mlp_ratio=4, dropout=0.1, time_conditioning=True, norm_type="layernorm", sandwich_normalization=False,
                 qk_norm=False):
        super().__init__()
        self.time_conditioning, self.dropout, self.sandwich_normalization = time_conditioning, dropout, sandwich_normalization
        self.attention = Attention(dim, n_heads, qk_norm=qk_norm)
        self.norm1 = get_norm(dim, norm_type)
        self.norm2 = get_norm(dim, norm_type)
        self.mlp = nn.Sequential(nn.Linear(dim, mlp_ratio * dim, bias=True), nn.GELU(approximate="tanh"), nn.Linear(mlp_ratio * dim, dim, bias=True))
        if time_conditioning:
            self.adaLN_modulation = nn.Linear(cond_dim, 6 * dim, bias=True)
            self.adaLN_modulation.weight.data.zero_()
            self.adaLN_modulation.bias.data.zero_()
        if sandwich_normalization:
            self.post_ff_norm = get_norm(dim, norm_type)
            self.pre_residual_norm = get_norm(dim, norm_type)


class DDitFinalLayer(nn.Module):  # models/dit.py:1063-1092
    def __init__(self, hidden_size, out_channels, cond_dim, time_conditioning=True, norm_type="layernorm", zero_linear_init=True):
        super().__init__()
        self.time_conditioning = time_conditioning
        self.norm_final = get_norm(hidden_size, norm_type)
        self.linear = nn.Linear(hidden_size, out_channels)
        if zero_linear_init:
            self.linear.weight.data.zero_()
        self.linear.bias.data.zero_()
        if time_conditioning:
            self.adaLN_modulation = nn.Linear(cond_dim, 2 * hidden_size, bias=True)
            self.adaLN_modulation.weight.data.zero_()
            self.adaLN_modulation.bias.data.zero_()


class ModalityMask:
    """What the reference's `get_block_mask(txt_batch_attn_dropout, img_batch_attn_dropout, txt_length, ...)` (model_utils.py:721-737) describes: in
    samples with txt_drop[b] text queries attend to text keys only, with img_drop[b] image queries to image keys only (positions < txt_length are
    text).  Passed as `block_mask=` where the reference passes its FlexAttention BlockMask."""

    def __init__(self, txt_drop, img_drop, txt_length):
        self.txt_drop, self.img_drop, self.txt_length = txt_drop.reshape(-1).bool(), img_drop.reshape(-1).bool(), int(txt_length)


class _Lin:
    """bf16 shadows of one nn.Linear weight: w16 [out(p), in] for forward, w16t [in, out(p)] for dgrad."""

    def __init__(self, weight: nn.Parameter, bias: Optional[nn.Parameter], out_pad: int = 8):
        self.weight, self.bias = weight, bias
        self.out, self.inp = weight.shape
        self.outp = _ceil(self.out, out_pad)
        self.w16 = self.w16t = None
        self.sticky_t = False
        self.need_t = True   # False: every dgrad of this layer reads W itself (K.gemm_nn), the per-step cast writes no transposed shadow

    def alloc(self):
        dev = self.weight.device
        if self.w16 is None or self.w16.device != dev:
            self.w16 = torch.zeros((self.outp, self.inp), dtype=BF16, device=dev)
            self.w16t = None
        if self.need_t and self.w16t is None:
            self.w16t = torch.zeros((self.inp, self.outp), dtype=BF16, device=dev)
        if not self.need_t:
            self.w16t = None

    def refresh(self):
        self.alloc()
        K.cast_transpose(self.weight.detach(), self.w16, self.w16t)

    def dgrad_form(self):
        """Which operand the dgrad of a forward that runs NOW will read, decided at FORWARD time and carried to the backward in the forward's saved state:
        "nt" = the transposed shadow (it exists for this forward and, being sticky, for every later one), "nn" = W's forward shadow (no transposed one is kept)."""
        return "nt" if self.w16t is not None else "nn"

    def dgrad(self, dY, M_rows, form=None):
        """dX [M, in] = dY [M, out] W in the form the forward recorded (`dgrad_form`); a transposed shadow that appeared since (another forward with a row
        count outside gemm_nn made it sticky) is current and takes over an "nn" record whose shape no longer fits - never the other way round."""
        form = self.dgrad_form() if form is None else form
        if form == "nt" or (self.w16t is not None and not K.gemm_nn_ok(M_rows, self.inp, self.out)):
            if self.w16t is None:
                raise RuntimeError(f"unidisc_amd: the forward of this {self.out}x{self.inp} Linear recorded an NT dgrad but its transposed weight shadow is gone")
            return K.gemm_nt(dY, self.w16t, N=self.inp)
        if not K.gemm_nn_ok(M_rows, self.inp, self.out):
            raise RuntimeError(f"unidisc_amd: dgrad of a {self.out}x{self.inp} Linear over {M_rows} rows has no transposed weight shadow and the shape is outside gemm_nn")
        return K.gemm_nn(dY, self.w16, N=self.inp)


# ------------------------------------------------------------------------------------------------
# the module
# ------------------------------------------------------------------------------------------------
class DIT(nn.Module, _HubMixin):
    def __init__(self, config, vocab_size: int, text_vocab_size: int, mask_index: int, dtype=None, device=None, static_img_sl=None, static_txt_sl=None,
                 **kwargs):
        super().__init__()
        self.config = config
        m, tr, data = cfg_get(config, "model"), cfg_get(config, "trainer"), cfg_get(config, "data")
        self.autocast_dtype = dtype
        self.vocab_size, self.text_vocab_size, self.mask_index = vocab_size, text_vocab_size, mask_index
        self.time_conditioning = bool(cfg_get(config, "time_conditioning", False) or cfg_get(m, "force_time_conditioning", False))
        self.use_gradient_checkpointing = cfg_get(tr, "use_gradient_checkpointing", False)
        # trainer.use_gradient_checkpointing (models/dit.py:1486-1490: every block is recomputed in the backward; the reference needs it on 48 GB parts): the
        # engine then keeps only each block's input (and the head's activations) and re-runs the block's forward right before its backward - same kernels,
        # same dropout seeds, bit-identical gradients (tests/test_engine_orchestration.py, tests/test_gpu_e2e.py).  Off, a block's activations stay resident
        # (0.92 GB per block at 1.4 B, B = 8: 22 GB of 288 GB), which is the faster choice whenever they fit.
        self.use_gradient_checkpointing = bool(self.use_gradient_checkpointing)
        self.sandwich_normalization = cfg_get(m, "sandwich_normalization", False)
        self.static_img_sl, self.static_txt_sl = static_img_sl, static_txt_sl
        for flag, why in (("img_cond", "cross-attention image conditioning"), ("cond_label", "class-label conditioning"),
                          ("use_pretrained_img_emb", "pretrained VQ embedding table"), ("use_kv_cache", "inference KV cache"),
                          ("use_flex_attention_cache", "inference modality KV cache")):
            if cfg_get(m, flag, False):
                raise NotImplementedError(f"unidisc_amd.DIT: model.{flag} ({why}) is outside the denoising hot path (SURVEY.md §8)")
        if not cfg_get(m, "full_attention", True):
            raise NotImplementedError("unidisc_amd.DIT: causal attention (model.full_attention=false) is not on the denoising hot path")
        if cfg_get(tr, "image_mode", "discrete") == "continuous":
            raise NotImplementedError("unidisc_amd.DIT: continuous image mode is not on the denoising hot path")
        if cfg_get(m, "attn_dropout", None):
            raise NotImplementedError("unidisc_amd.DIT: attention dropout is not implemented (no shipped config enables it)")

        d, H = cfg_get(m, "hidden_size"), cfg_get(m, "n_heads")
        self.hidden_size, self.n_heads, self.head_dim = d, H, d // H
        self.cond_dim, self.n_blocks = cfg_get(m, "cond_dim"), cfg_get(m, "n_blocks")
        self.norm_type, self.qk_norm = cfg_get(m, "norm_type", "layernorm"), cfg_get(m, "qk_norm", False)
        self.dropout = float(cfg_get(m, "dropout", 0.0) or 0.0)
        if self.head_dim not in (32, 64, 128):
            raise NotImplementedError(f"unidisc_amd.DIT: head_dim {self.head_dim} unsupported (32/64/128)")

        self.vocab_embed = EmbeddingLayer(d, vocab_size)
        self.sigma_map = TimestepEmbedder(self.cond_dim) if self.time_conditioning else None
        self.modality_embed = EmbeddingLayer(d, 2) if cfg_get(m, "modality_embed", False) else None

        self.txt_length, self.img_length, self.total_length = cfg_get(m, "txt_length"), cfg_get(m, "img_length"), cfg_get(m, "length")
        self.multimodal_batches = bool(cfg_get(tr, "multimodal_batches", False))
        self.rope_2d = bool(cfg_get(m, "rope_2d", False))
        # The softmax scale lives in the stored q: the qk-norm + rope kernel writes bf16(q log2(e) / sqrt(D)) - one rounding, where the reference rounds q and
        # flash-attn scales the fp32 scores - so the attention kernels' scores are base-2 exponents as they leave the matrix pipe (no multiply per score in
        # the forward or in either backward kernel); the rope backward multiplies the incoming dq by the same factor.
        self.attn_q_scale = K.attention_q_scale(self.head_dim)
        # model.head_chunk_rows (extension key, 0 = off): fused vocabulary head + SUBS cross-entropy that never materialises [rows, V] logits - the head
        # runs on chunks of this many rows (forward: logits chunk -> log p, dropped; backward: the chunk's logits are recomputed, d logits formed in
        # place and consumed by the dgrad / wgrad).  Trades one extra head GEMM per step for rows * V * 2 bytes of peak memory (SURVEY K11 + K12).
        self.head_chunk_rows = int(cfg_get(m, "head_chunk_rows", 0) or 0)
        self.require_sample_ids = bool(cfg_get(data, "require_sample_ids", False))
        assert (self.txt_length + self.img_length == self.total_length) or self.multimodal_batches
        D = self.head_dim
        if self.rope_2d:  # models/dit.py:1203-1232
            if not (self.multimodal_batches and self.modality_embed is not None):
                raise NotImplementedError("unidisc_amd.DIT: rope_2d needs multimodal_batches and modality_embed (as in every shipped config)")
            if self.require_sample_ids:  # interleaved / packed batches (:1209-1216): one table per supported image block, image-count embedding
                for n_img, lf in self.IMG_BLOCKS:
                    side = int(math.sqrt(n_img))
                    emb = self._lumina(D, side, side, lf)
                    self.register_buffer(f"rotary_cos_emb_img_{n_img}", emb.flatten(0, 1).real.contiguous(), persistent=False)
                    self.register_buffer(f"rotary_sin_emb_img_{n_img}", emb.flatten(0, 1).imag.contiguous(), persistent=False)
                self.img_count_embedding = nn.Parameter(torch.zeros((16, d)))
            else:
                side = int(math.sqrt(self.img_length))
                assert side * side == self.img_length, f"seq_len_2d must be a square number, got {self.img_length}"
                emb = self._lumina(D, side, side, cfg_get(m, "linear_factor", 1.0))
                self.register_buffer("rotary_cos_emb_img", emb.flatten(0, 1).real.contiguous(), persistent=False)
                self.register_buffer("rotary_sin_emb_img", emb.flatten(0, 1).imag.contiguous(), persistent=False)
            c, s = rotary_table_1d(self.total_length, D)
            self.register_buffer("rotary_cos_emb_txt", c, persistent=False)
            self.register_buffer("rotary_sin_emb_txt", s, persistent=False)
        else:  # :1234-1239
            c, s = rotary_table_1d(self.total_length, D)
            self.register_buffer("rotary_cos_emb", c, persistent=False)
            self.register_buffer("rotary_sin_emb", s, persistent=False)

        self.blocks = nn.ModuleList([
            DDiTBlock(d, H, self.cond_dim, dropout=self.dropout, time_conditioning=self.time_conditioning, norm_type=self.norm_type,
                      sandwich_normalization=self.sandwich_normalization, qk_norm=self.qk_norm) for _ in range(self.n_blocks)
        ])
        self.output_layer = DDitFinalLayer(d, vocab_size, self.cond_dim, time_conditioning=self.time_conditioning, norm_type=self.norm_type,
                                           zero_linear_init=cfg_get(m, "zero_linear_init", True))
        self.allow_compiled_embed = False
        if device is not None:
            self.to(device)
        # engine state
        self._lins: Optional[Dict[str, _Lin]] = None
        self._fwd_count = 0
        self.compact_head = True          # "logp" mode: run the vocabulary head on the [MASK] rows only (exact: other rows have log p = 0)
        # ... and with it everything of the LAST block behind its attention (out-proj, residual adds, MLP, final norm): only the head reads that
        # block's output, so its unmasked rows feed nothing and receive a zero gradient (exact; no adaLN: the row kernels would need the row -> sample map)
        self.compact_last_block = os.environ.get("UDM_COMPACT_LAST", "1") != "0"
        # SUBS with force_argmax_valid_indices: a text row's logits matter on the text ids only, an image row's on the image ids only (everything else is -inf in the
        # forward and has a zero gradient) - the head (forward, dgrad, wgrad) then runs as (text rows x text ids) + (image rows x image ids) instead of rows x all ids
        self.split_head = os.environ.get("UDM_SPLIT_HEAD", "1") != "0"
        self.pair_wgrads = os.environ.get("UDM_PAIR_WGRADS", "1") != "0"   # qkv + out-proj weight gradients in one 256-tile launch (K.gemm_tn_pair)
        # the four weight gradients of a block in ONE split-K launch + ONE reduce where each of them is a few tiles over the long row contraction (UniDisc-S)
        self.multi_wgrads = os.environ.get("UDM_MULTI_WGRADS", "1") != "0"
        self.dgrad_from_w = os.environ.get("UDM_DGRAD_NN", "1") != "0"   # dgrads from the forward's W shadow where the shape allows (see refresh_weight_shadows)
        self.grad_ready_callback = None   # fn(flat_grads, lo, hi): elements [lo, hi) of this backward's flat fp32 gradient buffer are final
        self.grad_sync_finish = None      # fn(): called at the end of backward (e.g. make the compute stream wait for the all-reduces)
        self.recast_every_forward = True  # mirror autocast: fp32 master -> bf16 shadow on every training forward
        self.overlap_weight_cast = os.environ.get("UDM_OVERLAP_CAST", "0") != "0"   # opt-in: the casts on a side stream under the blocks before (measured: no gain, 96.1 vs 96.2 ms)
        self._shadow_versions = None

    @staticmethod
    def _lumina(D, h, w, linear_factor):
        try:
            from diffusers.models.embeddings import get_2d_rotary_pos_embed_lumina as fn  # the reference's source, when installed
            return fn(D, h, w, linear_factor=linear_factor, ntk_factor=1.0)
        except Exception:
            return lumina_rope_2d(D, h, w, linear_factor=linear_factor, ntk_factor=1.0)

    # Reference API surface that callers touch.  `eval.attention_caching` (model_eval.py:2296-2366): the reference allocates a per-layer
    # K / V buffer here and WRITES it in the cache-building / text-only steps (models/dit.py:797-803), but no forward ever reads it back (:812 attends
    # to the keys of the current input only) - so the sampler's three kinds of step need no state in the backbone: `Diffusion.sample` passes the
    # image-queries-see-image-keys mask / the text slice itself.  The two calls are accepted and remembered for callers that probe them.
    def reset_kv_cache(self, *a, **k):
        self.use_flex_attention_cache = False

    def set_flex_attention_cache(self, batch_size=None, seq_len=None, device=None, dtype=None):
        self.use_flex_attention_cache = True

    # -------------------------------------------------------------------------------------------- parameters / shadows
    IMG_BLOCKS = ((256, 1), (1024, 2), (2304, 3), (4096, 4))   # (tokens of an image block, Lumina linear factor), models/dit.py:1210

    def _ordered_params(self) -> List[nn.Parameter]:
        """Parameters in the order their gradients become final during backward (head first, embeddings last)."""
        out = list(self.output_layer.parameters())
        for blk in reversed(self.blocks):
            out += list(blk.parameters())
        out += list(self.vocab_embed.parameters())
        if self.modality_embed is not None:
            out += list(self.modality_embed.parameters())
        if getattr(self, "img_count_embedding", None) is not None:
            out.append(self.img_count_embedding)
        if self.sigma_map is not None:
            out += list(self.sigma_map.parameters())
        seen = {id(p) for p in out}
        assert len(seen) == len(out) == sum(1 for _ in self.parameters())
        return out

    def _build_lins(self):
        L: Dict[str, _Lin] = {}
        for i, blk in enumerate(self.blocks):
            L[f"{i}.qkv"] = _Lin(blk.attention.attn_qkv.weight, None)
            L[f"{i}.out"] = _Lin(blk.attention.attn_out.weight, None)
            L[f"{i}.fc1"] = _Lin(blk.mlp[0].weight, blk.mlp[0].bias)
            L[f"{i}.fc2"] = _Lin(blk.mlp[2].weight, blk.mlp[2].bias)
            if self.time_conditioning:
                L[f"{i}.ada"] = _Lin(blk.adaLN_modulation.weight, blk.adaLN_modulation.bias)
        L["head"] = _Lin(self.output_layer.linear.weight, self.output_layer.linear.bias, out_pad=128)
        if self.time_conditioning:
            L["head.ada"] = _Lin(self.output_layer.adaLN_modulation.weight, self.output_layer.adaLN_modulation.bias)
            L["sig0"] = _Lin(self.sigma_map.mlp[0].weight, self.sigma_map.mlp[0].bias)
            L["sig2"] = _Lin(self.sigma_map.mlp[2].weight, self.sigma_map.mlp[2].bias)
        self._lins = L

    def refresh_weight_shadows(self, force=False, rows=None):
        if self._lins is None or next(iter(self._lins.values())).weight.device != self.vocab_embed.embedding.device:
            self._build_lins()
            force = True
        if rows is not None and self.dgrad_from_w and next(iter(self._lins.values())).weight.is_cuda:
            # qkv / out-proj / mlp.0 dgrads read W itself (K.gemm_nn) where `rows` x in x out are whole tiles of that kernel: their transposed
            # shadows - a quarter of the per-step cast traffic - are then not produced.  (mlp.2's dgrad carries the GELU' epilogue and the last
            # block may run on a compacted row list: they keep the transposed shadow.)
            last = len(self.blocks) - 1
            for name, lin in self._lins.items():
                blk, _, kind = name.partition(".")
                want_t = not (kind in ("qkv", "out", "fc1") and not (self.compact_last_block and int(blk) == last) and K.gemm_nn_ok(rows, lin.inp, lin.out))
                # sticky: once some forward needed the transposed shadow it stays (a second forward with another row count - micro-batches, an eval pass inside
                # a training step - must not drop the shadow a still-pending backward dispatches on, nor force a full recast at every alternation)
                lin.sticky_t = lin.sticky_t or want_t
                want_t = lin.sticky_t
                if want_t != lin.need_t:
                    lin.need_t, force = want_t, True
        versions = [l.weight._version for l in self._lins.values()]
        self._cast_events = {}
        if force or self.recast_every_forward and self.training or versions != self._shadow_versions:
            groups: Dict[str, list] = {}
            for name, lin in self._lins.items():   # "pre" (timestep MLP), one group per block in forward order, "head"
                key = "pre" if name.startswith("sig") else ("head" if name.startswith("head") else name.split(".")[0])
                groups.setdefault(key, []).append(lin)
            order = [k for k in ["pre"] + [str(i) for i in range(len(self.blocks))] + ["head"] if k in groups]
            dev = self.vocab_embed.embedding.device
            if self.overlap_weight_cast and dev.type == "cuda" and not force:
                # Opt-in (UDM_OVERLAP_CAST=1): the casts are HBM-bound and the GEMMs they feed are not, so all but the first blocks' shadows are
                # refreshed on a side stream under the forward of the blocks before them: a few multi-matrix launches (first two blocks /
                # next quarter / the rest + head); the compute stream waits for a group right before its first block (_await_cast).
                if getattr(self, "_cast_stream", None) is None:
                    self._cast_stream = torch.cuda.Stream(device=dev)
                side, main = self._cast_stream, torch.cuda.current_stream(dev)
                nb = len(self.blocks)
                cuts = [0, min(2, nb), min(max(nb // 3, 2), nb), nb]
                bounds = sorted(set(cuts))
                parts = []
                for lo_b, hi_b in zip(bounds[:-1], bounds[1:]):
                    keys = [str(i) for i in range(lo_b, hi_b)]
                    if lo_b == 0:
                        keys = ["pre"] + keys
                    if hi_b == nb:
                        keys = keys + ["head"]
                    parts.append([k for k in keys if k in groups])
                for lins_k in parts:
                    for k in lins_k:
                        for lin in groups[k]:
                            lin.alloc()
                key = tuple((l.weight.data_ptr(), l.w16.data_ptr(), l.w16t.data_ptr() if l.w16t is not None else 0) for k in order for l in groups[k])
                if getattr(self, "_cast_parts_key", None) != key:
                    self._cast_parts = [K.cast_transpose_jobs([(l.weight.detach(), l.w16, l.w16t) for k in ks for l in groups[k]], dev) for ks in parts]
                    self._cast_parts_key = key
                side.wait_stream(main)   # everything enqueued so far (the previous backward reads the Wᵀ shadows) comes first
                with torch.cuda.stream(side):
                    for ks, jobs in zip(parts, self._cast_parts):
                        K.cast_transpose_multi(jobs)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        self._cast_events[ks[0]] = ev
            else:
                # every weight of the forward in ONE launch (97 at 1.4 B): the job table lives on the device and is rebuilt only when a
                # master weight or a shadow moved
                lins = [lin for k in order for lin in groups[k]]
                for lin in lins:
                    lin.alloc()
                key = tuple((l.weight.data_ptr(), l.w16.data_ptr(), l.w16t.data_ptr() if l.w16t is not None else 0) for l in lins)
                if getattr(self, "_cast_jobs_key", None) != key:
                    self._cast_jobs = K.cast_transpose_jobs([(l.weight.detach(), l.w16, l.w16t) for l in lins], dev)
                    self._cast_jobs_key = key
                K.cast_transpose_multi(self._cast_jobs)
            self._shadow_versions = versions

    def invalidate_shadows(self):
        """Force the next forward to rebuild the bf16 weight shadows.  Needed after any write to the master weights that does not bump tensor
        versions (collectives, ``p.data`` writes, kernels called through ctypes, optimizer state reloads)."""
        self._shadow_versions = None

    @staticmethod
    def _dropout_rank():
        """Data-parallel ranks seeded identically must still draw different dropout masks (torch's CUDA generator differs per process in the
        reference because every rank seeds with seed + rank, main.py:1058-1068; here the rank is mixed in explicitly)."""
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def _await_cast(self, key):
        ev = self._cast_events.pop(str(key), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # -------------------------------------------------------------------------------------------- public forward
    def forward(self, indices, sigma=None, label=None, x_cond=None, attention_mask=None, continuous_mode=False, x_img_emb=None, modality=None,
                start_pos=None, block_mask=None, update_cache_slice=None, sample_ids=None):
        """→ logits [B, L, V] (bf16).  Signature of the reference's DIT.forward (models/dit.py:1324-1338)."""
        self._check_unsupported(label, x_cond, None, continuous_mode, x_img_emb, start_pos, block_mask, update_cache_slice, sample_ids)
        params = self._ordered_params()
        inputs = dict(indices=indices, sigma=sigma, modality=modality, sample_ids=sample_ids, x0=None, save=self._needs_grad(params),
                      block_mask=block_mask if isinstance(block_mask, ModalityMask) else None, key_mask=self._key_mask(attention_mask, indices))
        return _DitFn.apply(self, "logits", inputs, *params)

    @staticmethod
    def _key_mask(attention_mask, indices):
        """`model.use_attention_mask` (model.py:405-406 -> `sdpa(..., attn_mask=attention_mask)`, models/dit.py:829): the batch's [B, L] bool mask reaches SDPA
        unchanged, where a 2-D mask aligns with the (query, key) axes - i.e. it broadcasts (B = 1; for B > 1 the reference call does not broadcast and
        fails) as a KEY mask: padded keys are hidden from every query of their sample, padded queries still attend to the valid keys."""
        if attention_mask is None:
            return None
        if attention_mask.dtype != torch.bool or tuple(attention_mask.shape) != tuple(indices.shape):
            raise NotImplementedError("unidisc_amd.DIT.forward: attention_mask must be the batch's bool [B, L] padding mask (dense [B, 1, L, L] masks - the "
                                      "transfusion path - are outside the denoising hot path)")
        return attention_mask

    def forward_logp(self, xt, x0, sigma=None, modality=None, sample_ids=None, restrict_modality=False, block_mask=None, attention_mask=None):
        """Fused training path: log p_theta(x0 | xt) per token [B, L] fp32 under the SUBS parameterisation
        (== gather(_subs_parameterization(logits, xt), x0), model.py:621-658 + :967) without materialising log-probs.
        block_mask: a `ModalityMask` (modality attention dropout) or None.  attention_mask: the key-padding mask of `model.use_attention_mask` or None."""
        params = self._ordered_params()
        inputs = dict(indices=xt, sigma=sigma, modality=modality, sample_ids=sample_ids, x0=x0, restrict=restrict_modality, save=self._needs_grad(params),
                      block_mask=block_mask, key_mask=self._key_mask(attention_mask, xt))
        return _DitFn.apply(self, "logp", inputs, *params)

    @torch.no_grad()
    def forward_masked_logits(self, xt, sigma=None, modality=None, sample_ids=None, plan_ids=None, block_mask=None):
        """Sampler path: (logits [R, Vp] bf16 of the [MASK] positions of `xt` first, then padding rows; their flat row indices [R];
        the number of [MASK] rows).  Unmasked positions keep their token under SUBS (model.py:646-656), so they need no logits.
        `plan_ids` (same shape as xt): take the row selection from the [MASK] positions of this tensor instead of xt's (guided sampling runs
        [x ; x_uncond] as one batch and needs the SAME positions from both halves)."""
        inputs = dict(indices=xt, sigma=sigma, modality=modality, sample_ids=sample_ids, x0=None, save=False, plan_ids=plan_ids,
                      block_mask=block_mask if isinstance(block_mask, ModalityMask) else None)
        out, _ = self._engine_forward(inputs, "rows", save=False)
        return out

    @staticmethod
    def _needs_grad(params):
        return torch.is_grad_enabled() and any(p.requires_grad for p in params)

    def _check_unsupported(self, label, x_cond, attention_mask, continuous_mode, x_img_emb, start_pos, block_mask, update_cache_slice, sample_ids):
        for name, v in (("label", label), ("x_cond", x_cond), ("attention_mask", attention_mask), ("x_img_emb", x_img_emb), ("start_pos", start_pos),
                        ("update_cache_slice", update_cache_slice)):
            if v is not None:
                raise NotImplementedError(f"unidisc_amd.DIT.forward: argument `{name}` is outside the denoising hot path")
        if continuous_mode:
            raise NotImplementedError("unidisc_amd.DIT.forward: continuous_mode is outside the denoising hot path")
        if block_mask is not None and block_mask is not True and sample_ids is None and not isinstance(block_mask, ModalityMask):
            raise NotImplementedError("unidisc_amd.DIT.forward: pass a unidisc_amd.ModalityMask (modality attention dropout) as block_mask; "
                                      "FlexAttention BlockMask objects are not used here (document masks are derived from sample_ids)")

    # -------------------------------------------------------------------------------------------- engine: forward
    def _rotary_interleaved(self, modality, sample_ids):
        """Rotary rows and image-count indices of packed rows: ONE launch on the GPU (`udm_interleaved_rope`, tokens.hip), the tensor-statement form below elsewhere
        (CPU tests) - bit-identical (tests/test_gpu_kernels.py)."""
        if modality.is_cuda and os.environ.get("UDM_INTERLEAVED_KERNELS", "1") != "0":
            dev = modality.device
            if getattr(self, "_img_tab_cos", None) is None or self._img_tab_cos.device != dev:
                self._img_tab_cos = torch.cat([getattr(self, f"rotary_cos_emb_img_{n}") for n, _ in self.IMG_BLOCKS], 0).to(dev).contiguous()
                self._img_tab_sin = torch.cat([getattr(self, f"rotary_sin_emb_img_{n}") for n, _ in self.IMG_BLOCKS], 0).to(dev).contiguous()
            return K.interleaved_rope(modality, sample_ids, self._img_tab_cos, self._img_tab_sin, [n for n, _ in self.IMG_BLOCKS],
                                      self.rotary_cos_emb_txt, self.rotary_sin_emb_txt)
        return self._rotary_interleaved_torch(modality, sample_ids)

    def _rotary_interleaved_torch(self, modality, sample_ids):
        """models/dit.py:1421-1444 with `add_img_data_to_blocks` / `add_txt_data_to_blocks` (:122-191), as tensor operations on the device (the
        reference loops over blocks on the host): cos, sin fp32 [B, L, D/2] and, per position, the row of `img_count_embedding` to add (-1: none).
        Image runs whose length is a supported block size get that size's 2-D table and count embedding j = number of earlier image runs of the row
        that start in the same packed sample; text positions inside a run of one sample id >= 0 get the 1-D table from the start of that run;
        everything else (padding, image runs of other lengths) keeps cos = sin = 0."""
        B, L = modality.shape
        dev = modality.device
        ar = torch.arange(L, device=dev)[None].expand(B, L)
        is_img = modality.bool()
        prev = torch.nn.functional.pad(is_img[:, :-1], (1, 0))
        start = is_img & ~prev
        run_start = torch.cummax(torch.where(start, ar, torch.full_like(ar, -1)), dim=1).values
        rows = torch.arange(B, device=dev)[:, None].expand(B, L)
        key = (rows * L + run_start.clamp(min=0))
        # (static shapes only - no boolean-mask indexing, no host reads: the host must be able to run ahead of the device across this preamble)
        counts = torch.bincount(torch.where(is_img, key, torch.full_like(key, B * L)).reshape(-1), minlength=B * L + 1)[: B * L]
        run_len = torch.where(is_img, counts[key], torch.zeros_like(key))
        pos_in_run = ar - run_start
        if getattr(self, "_img_tab_cos", None) is None or self._img_tab_cos.device != dev:
            self._img_tab_cos = torch.cat([getattr(self, f"rotary_cos_emb_img_{n}") for n, _ in self.IMG_BLOCKS], 0).to(dev)
            self._img_tab_sin = torch.cat([getattr(self, f"rotary_sin_emb_img_{n}") for n, _ in self.IMG_BLOCKS], 0).to(dev)
        off = torch.full_like(run_len, -1)
        base = 0
        for n, _ in self.IMG_BLOCKS:
            off = torch.where(run_len == n, torch.full_like(off, base), off)
            base += n
        valid_img = is_img & (off >= 0)
        idx = (off + pos_in_run).clamp(min=0)
        cos_img, sin_img = self._img_tab_cos[idx], self._img_tab_sin[idx]
        # image index inside its packed sample: running count of image-run starts per (row, sample id)
        sid = sample_ids
        # rank of every image-run start among the starts of its (row, sample id), in position order: stable sort of the flattened positions by group key (non-starts
        # sort behind every group), rank = index in the sorted order - index of the group's first element.  No one-hot over the number of sample ids (which would need it
        # on the host), any layout of ids inside a row.
        N = B * L
        flat_start = start.reshape(-1)
        gkey = torch.where(flat_start, (rows * (L + 2) + sid.clamp(min=-1) + 1).reshape(-1), torch.full((N,), B * (L + 2) + 1, dtype=torch.int64, device=dev))
        order = torch.argsort(gkey, stable=True)
        skey = gkey[order]
        arn = torch.arange(N, device=dev)
        newg = torch.ones(N, dtype=torch.bool, device=dev)
        newg[1:] = skey[1:] != skey[:-1]
        g_first = torch.cummax(torch.where(newg, arn, torch.zeros_like(arn)), 0).values
        j_at = torch.empty(N, dtype=torch.int64, device=dev).scatter_(0, order, arn - g_first).reshape(B, L)   # valid at run starts (read only there)
        j_run = j_at.gather(1, run_start.clamp(min=0))
        count_idx = torch.where(valid_img, j_run, torch.full_like(j_run, -1))
        # text: positions restart at every run of one sample id
        sprev = torch.nn.functional.pad(sid[:, :-1], (1, 0), value=-2)
        sstart = sid != sprev
        s_run_start = torch.cummax(torch.where(sstart, ar, torch.full_like(ar, -1)), dim=1).values
        tpos = (ar - s_run_start).clamp(min=0, max=self.rotary_cos_emb_txt.shape[0] - 1)
        is_txt = (~is_img) & (sid >= 0)
        zero = torch.zeros((), dtype=cos_img.dtype, device=dev)
        cos = torch.where(valid_img[:, :, None], cos_img, torch.where(is_txt[:, :, None], self.rotary_cos_emb_txt[tpos], zero))
        sin = torch.where(valid_img[:, :, None], sin_img, torch.where(is_txt[:, :, None], self.rotary_sin_emb_txt[tpos], zero))
        return cos.contiguous(), sin.contiguous(), count_idx

    def _rotary(self, modality, L):
        """models/dit.py:1413-1460 (non-interleaved branches) → fp32 [L, D/2] or per-sample [B, L, D/2]."""
        if self.modality_embed is not None and self.rope_2d and self.multimodal_batches:
            ci, si = self.rotary_cos_emb_img, self.rotary_sin_emb_img
            if modality.shape[-1] != self.img_length:
                pad = max(modality.shape[-1] - self.img_length, 0)
                nanpad = torch.full((pad, ci.shape[-1]), float("nan"), device=ci.device, dtype=ci.dtype)
                ci, si = torch.cat([nanpad, ci], 0), torch.cat([nanpad, si], 0)
            sel = modality[:, :, None] == 0
            cos = torch.where(sel, self.rotary_cos_emb_txt[None, :L], ci[None, :L]).contiguous()
            sin = torch.where(sel, self.rotary_sin_emb_txt[None, :L], si[None, :L]).contiguous()
            return cos, sin
        return self.rotary_cos_emb[:L].contiguous(), self.rotary_sin_emb[:L].contiguous()

    def _engine_forward(self, inp, mode, save):
        ids = inp["indices"]
        K.require_gpu(ids)
        dev = ids.device
        B, L = ids.shape
        M, d, H, D = B * L, self.hidden_size, self.n_heads, self.head_dim
        if M % 8 != 0:
            raise ValueError(f"unidisc_amd.DIT: batch*length = {M} must be a multiple of 8")
        nt = K.norm_id(self.norm_type)
        tc, sw = self.time_conditioning, self.sandwich_normalization
        train = self.training
        self.refresh_weight_shadows(rows=M)
        lin = self._lins
        p_drop = self.dropout if train else 0.0
        if train and save:   # only training forwards advance the dropout stream (eval / sampler passes draw no dropout masks)
            self._fwd_count += 1
        seed0 = ((torch.initial_seed() * 1000003 + self._fwd_count * 4096) ^ (self._dropout_rank() * 0x9E3779B97F4A7C15)) & ((1 << 62) - 1)

        ids = ids.contiguous().view(-1).to(torch.int64)
        modality = inp["modality"]
        mod_flat = modality.contiguous().view(-1).to(torch.int64) if modality is not None else None
        sample_ids = inp["sample_ids"]
        sid = sample_ids.contiguous().to(torch.int64) if sample_ids is not None else None
        if self.modality_embed is not None and mod_flat is None:
            if self.multimodal_batches:
                raise ValueError("unidisc_amd.DIT: modality_embed with multimodal_batches needs the `modality` argument")
            pos = torch.arange(L, device=dev)  # static slices (models/dit.py:1409-1411)
            is_img = torch.zeros(L, dtype=torch.bool, device=dev)
            is_img[self.static_img_sl] = True
            emb_mod = is_img.to(torch.int64)[None].expand(B, L).contiguous().view(-1)
            del pos
        else:
            emb_mod = mod_flat
        doc_ranges = K.attention_doc_ranges(sid) if sid is not None else None   # once per step, shared by every block's forward and backward
        bm = inp.get("block_mask")
        if isinstance(bm, ModalityMask) and sid is None:
            # modality attention dropout (model.py:863-878): an asymmetric per-sample mask, carried to the attention kernels as mask codes in
            # the sample-id slot (class bits: no tile skipping, every tile takes the per-element test)
            sid = K.modality_mask_codes(bm.txt_drop.to(dev), bm.img_drop.to(dev), bm.txt_length, L)
        raw_sid = sid if sample_ids is not None else None   # (document ids as given: the per-sample rotary positions below need them without class bits)
        km = inp.get("key_mask")
        if km is not None:
            # key-padding mask (model.use_attention_mask): padded keys get a key class (4) no query mask contains; positions without other codes become
            # sample 0 / key class 1 / query mask "classes 1 | 2".  Class bits switch tile skipping off (doc_ranges = None): every tile takes the element test.
            km = km.to(dev).reshape(B, L)
            base = sid if sid is not None else torch.zeros((B, L), dtype=torch.int64, device=dev)
            kbits = (base >> 32) & 0xff
            qbits = (base >> 40) & 0xff
            kbits = torch.where(km, torch.where(kbits == 0, torch.ones_like(kbits), kbits), torch.full_like(kbits, 4))
            qbits = torch.where(qbits == 0, torch.full_like(qbits, 3), qbits)
            sid = ((base & 0xffffffff) | (kbits << 32) | (qbits << 40)).contiguous()
            doc_ranges = None
        S = dict(B=B, L=L, ids=ids, modality=mod_flat, emb_mod=emb_mod, sid=sid, p_drop=p_drop, seed0=seed0, blocks=[])
        S["doc_ranges"] = doc_ranges
        S["dgrad_form"] = {name: lin.dgrad_form() for name, lin in self._lins.items()}   # NN vs NT per Linear, fixed by THIS forward's shadows (not re-derived at backward time)
        # SUBS: only [MASK] rows have a non-zero log-probability (model.py:621-658), so in "logp" mode the vocabulary head (GEMM fwd,
        # dgrad, wgrad and the cross-entropy) runs on the masked rows only.  Their number is data dependent: it is counted on a side
        # stream NOW and only read back right before the head, when the host has already queued every block of this forward -- the
        # device never waits for the host.
        plan_ids = inp.get("plan_ids")
        split_ok = (mode == "logp" and self.compact_head and self.split_head and bool(inp.get("restrict", False)) and mod_flat is not None and plan_ids is None
                    and 0 < self.text_vocab_size < self.vocab_size and self.head_chunk_rows <= 0)
        head_plan = (self._plan_masked_rows(ids if plan_ids is None else plan_ids.reshape(ids.shape), mod_flat if split_ok else None)
                     if ((mode == "logp" and self.compact_head) or mode == "rows") else None)

        x = K.embedding_fwd(ids, self.vocab_embed.embedding.detach(), emb_mod if self.modality_embed is not None else None,
                            self.modality_embed.embedding.detach() if self.modality_embed is not None else None)
        if self.rope_2d and self.require_sample_ids:
            if raw_sid is None:
                raise ValueError("unidisc_amd.DIT: data.require_sample_ids needs sample_ids")
            cos, sin, count_idx = self._rotary_interleaved(modality.to(torch.int64), raw_sid)
            # x[b, l] += img_count_embedding[j] on the positions of supported image blocks.  Static shapes (no nonzero(): that is a host read in the step's preamble):
            # every position gathers a row of the table extended by one zero row, positions without an image index take that row
            n_cnt = self.img_count_embedding.shape[0]
            cnt_j = torch.where(count_idx.view(-1) >= 0, count_idx.view(-1), torch.full_like(count_idx.view(-1), n_cnt))
            tab = torch.cat([self.img_count_embedding.detach().to(x.dtype), torch.zeros((1, x.shape[1]), dtype=x.dtype, device=x.device)], 0)
            x.add_(tab.index_select(0, cnt_j))
            S["cnt_j"] = cnt_j
        else:
            cos, sin = self._rotary(modality, L)
        S["cos"], S["sin"] = cos, sin

        any_img = None
        if tc:
            sigma = inp["sigma"]
            if sigma is None:
                raise ValueError("unidisc_amd.DIT: time_conditioning needs sigma")
            sigma = sigma.reshape(-1).to(F32).contiguous()
            Bp = _ceil(B, 8)
            te = torch.zeros((Bp, 256), dtype=BF16, device=dev)
            K.timestep_embedding(sigma, te, B, 256)
            self._await_cast("pre")
            l1 = K.gemm_nt(te, lin["sig0"].w16, N=lin["sig0"].out, epilogue=K.EPI_BIAS, bias=lin["sig0"].bias.detach())
            s1 = K.silu_fwd(l1)
            l2 = K.gemm_nt(s1, lin["sig2"].w16, N=lin["sig2"].out, epilogue=K.EPI_BIAS, bias=lin["sig2"].bias.detach())
            c = K.silu_fwd(l2)
            if mod_flat is not None:
                any_img = (mod_flat != 0).any().to(torch.int32).reshape(1)
            S.update(Bp=Bp, te=te, l1=l1, s1=s1, l2=l2, c=c, any_img=any_img)

        ckpt = bool(self.use_gradient_checkpointing) and save
        ada_cache = {}

        def ada_mod(i):
            """adaLN_modulation(c) of block i (i == n_blocks: the final layer's), computed once: the residual add in front of a modulated norm needs it one block early"""
            key = str(i) if i < self.n_blocks else "head"
            if key not in ada_cache:
                self._await_cast(i if i < self.n_blocks else "head")
                a = lin[f"{key}.ada"]
                ada_cache[key] = K.gemm_nt(S["c"], a.w16, N=a.out, epilogue=K.EPI_BIAS, bias=a.bias.detach())   # [Bp, 6d] (final layer: [Bp, 2d]) bf16
            return ada_cache[key]

        def block_fwd(i, x, pre, last_rows=None, recompute=False):
            """One DiT block: x [M, d] fp32 -> (x_out, pre_out, R).  R holds what the block's backward reads.  `recompute` (activation checkpointing,
            models/dit.py:1486-1490): the backward re-runs the block from its saved input - same kernels, same dropout seeds, the fused next-block pre-norm
            replaced by the block's own norm kernel (bit-identical) - so only x_in (and the compacted row list of the last block) is kept per block."""
            blk = self.blocks[i]
            R = {}
            mod = None
            if not recompute:
                self._await_cast(i)
            if tc:
                mod = ada_mod(i)   # [Bp, 6d] bf16
                R["mod"] = mod
            if pre is not None:  # norm1 came fused out of the previous block's MLP residual add
                h1, rstd1, mean1 = pre
            else:
                h1, rstd1, mean1 = K.norm_fwd(x, blk.norm1.weight.detach(), nt, L, mod=mod, mod_idx=(0, 1), modality=mod_flat, any_img=any_img)
            qkv = K.gemm_nt(h1, lin[f"{i}.qkv"].w16, N=3 * d)
            at = blk.attention
            qn_kw = dict(gq=at.q_norm.weight.detach() if self.qk_norm else None, bq=at.q_norm.bias.detach() if self.qk_norm else None,
                         gk=at.k_norm.weight.detach() if self.qk_norm else None, bk=at.k_norm.bias.detach() if self.qk_norm else None)
            qkr, qstats = K.qknorm_rope_fwd(qkv, cos, sin, L, D, q_scale=self.attn_q_scale, **qn_kw)
            o, lse = K.attention_fwd(qkr, qkv, B, L, H, D, sid, S["doc_ranges"], q_prescaled=True)
            rows_c = last_rows
            if not recompute and i + 1 == self.n_blocks and head_plan is not None and mode == "logp" and self.compact_last_block and not tc:
                head_rows_c = self._masked_rows(head_plan, M)   # (the count was queued at the top of this forward: the host does not wait for the device here)
                if head_rows_c is not None:
                    rows_c = head_rows_c[0]
            x_full, o_full = x, o
            if rows_c is not None:   # from here on this block works on the [MASK] rows (+ padding) only
                o = o.index_select(0, rows_c)
                x = x.index_select(0, rows_c)
                Mb = rows_c.numel()
            else:
                Mb = M
            a_out = K.gemm_nt(o, lin[f"{i}.out"].w16, N=d)
            # The next pre-norm is fused into the residual add (x_out is normalised while in registers) - with adaLN in its modulated form (round 5)
            fuse_w2 = blk.norm2.weight.detach()
            nkw2 = dict(next_mod=mod, next_mod_idx=(3, 4), next_modality=mod_flat, next_any_img=any_img) if tc else {}
            if sw:  # x = x_skip + pre_residual_norm(attn)   (dit.py:993-994; no gate, no dropout)
                res = K.residual_fwd(x, a_out, L, w_b=blk.pre_residual_norm.weight.detach(), norm_type=nt, next_w=fuse_w2, **nkw2)
            else:   # bias_dropout_add_scale with gate_msa on every token (Attention.time_conditioning is never set: dit.py:533,884)
                res = K.residual_fwd(x, a_out, L, norm_type=nt, mod=mod, gate_idx=2 if tc else None, p_drop=p_drop, seed=seed0 + 4 * i + 1, next_w=fuse_w2, **nkw2)
            x_mid, rstd_a, mean_a = res[:3]
            h2, rstd2, mean2 = res[3]
            f1, f2 = lin[f"{i}.fc1"], lin[f"{i}.fc2"]
            u1 = torch.empty((Mb, 4 * d), dtype=BF16, device=dev)
            g = K.gemm_nt(h2, f1.w16, N=4 * d, epilogue=K.EPI_BIAS_GELU, bias=f1.bias.detach(), aux=u1)
            u2 = K.gemm_nt(g, f2.w16, N=d, epilogue=K.EPI_BIAS, bias=f2.bias.detach())
            # the consumer of x_out: norm1 of the next block, or norm_final (with adaLN: modulated by the NEXT block's / the final layer's adaLN output)
            nxt_w = (self.blocks[i + 1].norm1.weight if i + 1 < self.n_blocks else self.output_layer.norm_final.weight).detach()
            nkw = {}
            if tc and not recompute:
                nkw = dict(next_mod=ada_mod(i + 1), next_mod_idx=(0, 1), next_modality=mod_flat, next_any_img=any_img)
            elif tc:      # (recomputation of ONE block in the backward: its output's norm is not needed)
                nxt_w = None
            res = K.residual_fwd(x_mid, u2, L, w_b=blk.post_ff_norm.weight.detach() if sw else None, norm_type=nt, mod=mod,
                                 gate_idx=5 if tc else None, modality=mod_flat if tc else None, p_drop=p_drop, seed=seed0 + 4 * i + 2, next_w=nxt_w, **nkw)
            x_out, rstd_m, mean_m = res[:3]
            pre = res[3] if nxt_w is not None else None
            if save:
                if ckpt and not recompute:
                    R = dict(x_in=x_full, rows_c=rows_c, ckpt=True)
                else:
                    R.update(x_in=x_full, h1=h1, rstd1=rstd1, mean1=mean1, qkv=qkv, qkr=qkr, qstats=qstats, o=o_full, lse=lse, a_out=a_out, rstd_a=rstd_a,
                             mean_a=mean_a, x_mid=x_mid, h2=h2, rstd2=rstd2, mean2=mean2, u1=u1, g=g, u2=u2, rstd_m=rstd_m, mean_m=mean_m, rows_c=rows_c,
                             o_c=o if rows_c is not None else None)
            return x_out, pre, R

        pre = None
        for i in range(self.n_blocks):
            x, pre, R = block_fwd(i, x, pre)
            if save:
                S["blocks"].append(R)
        if ckpt:
            S["block_fwd"] = block_fwd

        fl = self.output_layer
        fmod = None
        self._await_cast("head")
        if tc:
            fmod = ada_mod(self.n_blocks)  # [Bp, 2d]
        if pre is not None:
            hf, rstdf, meanf = pre
        else:
            hf, rstdf, meanf = K.norm_fwd(x, fl.norm_final.weight.detach(), nt, L, mod=fmod, mod_idx=(0, 1), modality=mod_flat, any_img=any_img)
        head = lin["head"]
        V, Vp = self.vocab_size, head.outp
        if mode == "rows":  # sampler: logits of the [MASK] rows only (no autograd); rows = their flat indices, padded to a multiple of 64
            rows_p, n_masked = self._masked_rows(head_plan, M, always=True)
            hf_h = hf.index_select(0, rows_p)
            logits = torch.empty((hf_h.shape[0], Vp), dtype=BF16, device=dev)
            K.gemm_nt(hf_h, head.w16, out=logits, N=V, epilogue=K.EPI_BIAS, bias=head.bias.detach())
            return (logits, rows_p, n_masked), S
        if mode == "logits":
            logits = torch.empty((M, Vp), dtype=BF16, device=dev)
            K.gemm_nt(hf, head.w16, out=logits, N=V, epilogue=K.EPI_BIAS, bias=head.bias.detach())
            if save:
                S.update(x_final=x, hf=hf, rstdf=rstdf, meanf=meanf, fmod=fmod, logits=logits, head_rows=None)
            return logits[:, :V].view(B, L, V), S
        x0 = inp["x0"].contiguous().view(-1).to(torch.int64)
        restrict = bool(inp.get("restrict", False))
        if restrict and mod_flat is None:  # static slices (model.py:634-635)
            cm = torch.zeros(L, dtype=torch.int64, device=dev)
            cm[self.static_img_sl] = 1
            ce_mod = cm[None].expand(B, L).contiguous().view(-1)
        else:
            ce_mod = mod_flat
        ids_h = ids
        head_rows = self._masked_rows(head_plan, M) if head_plan is not None else None
        stream_compact = head_rows is not None and hf.shape[0] != M   # the last block already runs on the compacted rows (same list, same order)
        if head_rows is not None:  # compact operands: masked rows first, padded (with an unmasked row: zero loss, zero gradient) to a multiple of 64
            rows_p, n_masked = head_rows
            # fixed-capacity buffers (views of M-row allocations): a different size every step would make the caching allocator go back
            # to hipMalloc until its pool covers every size seen (measured: occasional 120+ ms steps)
            hf_h = hf if stream_compact else torch.index_select(hf, 0, rows_p, out=torch.empty_like(hf)[: rows_p.numel()])
            x0, ids_h = x0.index_select(0, rows_p), ids.index_select(0, rows_p)
            ce_mod = ce_mod.index_select(0, rows_p) if ce_mod is not None else None
        else:
            hf_h = hf
        Mh = hf_h.shape[0]
        chunk = _ceil(self.head_chunk_rows, 64) if self.head_chunk_rows > 0 else 0
        if chunk and chunk < Mh:   # fused head + cross-entropy over row chunks: only one chunk of logits exists at a time, none is kept
            logits = None
            buf = torch.empty((chunk, Vp), dtype=BF16, device=dev)
            lps, lses = [], []
            for r0 in range(0, Mh, chunk):
                r1 = min(Mh, r0 + chunk)
                lg = buf[: r1 - r0]
                K.gemm_nt(hf_h[r0:r1], head.w16, out=lg, N=V, epilogue=K.EPI_BIAS, bias=head.bias.detach())
                lp_c, lse_c = K.subs_ce_fwd(lg, x0[r0:r1], ids_h[r0:r1], ce_mod[r0:r1] if ce_mod is not None else None, V, self.text_vocab_size, self.mask_index, restrict)
                lps.append(lp_c)
                lses.append(lse_c)
            log_p, lse_ce = torch.cat(lps), torch.cat(lses)
            del buf
        else:
            chunk = 0
            logits = torch.empty((M, Vp), dtype=BF16, device=dev)[:Mh]
            groups = head_plan.get("groups") if (head_plan is not None and head_rows is not None and restrict) else None
            if groups is not None:
                # (text rows) x ids [0, c_up) and (image rows) x ids [c_al, V): the boundary Vt is rounded to multiples of 8 outwards so that every operand and
                # output pointer keeps its alignment; the few extra columns are never read (the cross-entropy reads a row's valid ids only)
                n_tp, n_ip = groups
                Vt = self.text_vocab_size
                c_al, c_up = Vt // 8 * 8, _ceil(Vt, 8)
                bias = head.bias.detach()
                if n_tp:
                    K.gemm_nt(hf_h[:n_tp], head.w16, out=logits[:n_tp], N=c_up, epilogue=K.EPI_BIAS, bias=bias)
                if n_ip:
                    K.gemm_nt(hf_h[n_tp:], head.w16[c_al:], out=logits[n_tp:, c_al:], N=V - c_al, epilogue=K.EPI_BIAS, bias=bias[c_al:])
            else:
                K.gemm_nt(hf_h, head.w16, out=logits, N=V, epilogue=K.EPI_BIAS, bias=head.bias.detach())
            log_p, lse_ce = K.subs_ce_fwd(logits, x0, ids_h, ce_mod, V, self.text_vocab_size, self.mask_index, restrict)
        if head_rows is not None:   # (padding rows are unmasked: their log p is exactly 0, like every row left out)
            log_p = torch.zeros(M, dtype=log_p.dtype, device=dev).index_copy_(0, rows_p, log_p)
        if save:
            S.update(x_final=x, hf=hf_h, rstdf=rstdf, meanf=meanf, fmod=fmod, logits=logits, head_rows=head_rows, ids_h=ids_h,
                     x0=x0, ce_mod=ce_mod, restrict=restrict, lse_ce=lse_ce, stream_compact=stream_compact, head_chunk=chunk,
                     head_groups=(head_plan.get("groups") if (head_plan is not None and head_rows is not None and restrict and not chunk) else None))
        return log_p.view(B, L), S

    def _plan_masked_rows(self, ids, mod=None):
        """Queue (without synchronising) a stable partition of the row indices with the [MASK] rows first and their count.  With `mod` (the rows' modality,
        0 = text / 1 = image) the [MASK] rows are ordered text first, then image, and both counts are reported: the vocabulary head can then run as two
        (rows of one modality) x (ids valid for that modality) problems (`_masked_rows`, split_head)."""
        is_mask = ids == self.mask_index
        if mod is not None:
            key = (~is_mask).to(torch.int8) * 2 + (mod == 1).to(torch.int8)     # 0 masked text, 1 masked image, 2 / 3 unmasked
            counts = lambda: torch.stack([(is_mask & (mod != 1)).sum(), (is_mask & (mod == 1)).sum()])
        else:
            key = (~is_mask).to(torch.int8)
            counts = lambda: is_mask.sum().view(1)
        if not ids.is_cuda:  # CPU orchestration tests
            c = counts()
            return dict(order=torch.argsort(key, stable=True), count=int(c.sum()), count2=(int(c[0]), int(c[1])) if mod is not None else None, event=None)
        dev = ids.device
        main = torch.cuda.current_stream(dev)
        side = self._side_streams.get(dev) if getattr(self, "_side_streams", None) else None
        if side is None:   # one side stream per device the module has run on
            self._side_streams = dict(getattr(self, "_side_streams", None) or {})
            side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        count_host = torch.empty(2 if mod is not None else 1, dtype=torch.int64).pin_memory()   # per call: forwards on two streams / threads must not share the counter
        side.wait_stream(main)
        with torch.cuda.stream(side):
            order = torch.argsort(key, stable=True)
            count_host.copy_(counts(), non_blocking=True)
            event = torch.cuda.Event()
            event.record(side)
        is_mask.record_stream(side)
        key.record_stream(side)
        order.record_stream(main)
        return dict(order=order, count=None, count2=None, event=event, count_host=count_host, side=side, by_modality=mod is not None)

    def _masked_rows(self, plan, M, always=False):
        """(row indices: the [MASK] rows, then as many unmasked rows as pad the list to a multiple of 64; number of masked rows), or
        None when compaction would not shrink the head.  Unmasked rows have zero loss and zero gradient, so padding with them is exact."""
        if plan.get("n") is not None:
            n = plan["n"]
        elif plan["event"] is not None:
            plan["event"].synchronize()
            ch = plan["count_host"]
            if plan.get("by_modality"):
                plan["count2"] = (int(ch[0]), int(ch[1]))
            n = int(ch.sum())
            torch.cuda.current_stream().wait_stream(plan["side"])
        else:
            n = plan["count"]
        plan["n"] = n
        if "rows" in plan:       # (decided once per forward: the last block's compaction and the head must see the same list)
            return plan["rows"]
        plan["groups"] = None
        n_pad = _ceil(max(n, 1), 64)
        if n_pad >= M:
            plan["rows"] = (plan["order"], n) if always else None   # always: every row, [MASK] rows first
            return plan["rows"]
        c2 = plan.get("count2")
        if c2 is not None and not always and self.split_head and n > 0:
            # two groups - [MASK] text rows, [MASK] image rows - each padded to a multiple of 64 with unmasked rows (zero loss, zero gradient: exact)
            n_t, n_i = c2
            n_tp, n_ip = (_ceil(n_t, 64) if n_t else 0), (_ceil(n_i, 64) if n_i else 0)
            pad_t, pad_i = n_tp - n_t, n_ip - n_i
            if n_tp + n_ip < M and pad_t + pad_i <= M - n:
                o = plan["order"]
                plan["rows"] = (torch.cat([o[:n_t], o[n:n + pad_t], o[n_t:n], o[n + pad_t:n + pad_t + pad_i]]), n)
                plan["groups"] = (n_tp, n_ip)
                return plan["rows"]
        plan["rows"] = (plan["order"][:n_pad], n)
        return plan["rows"]

    # -------------------------------------------------------------------------------------------- engine: backward
    def _alloc_grads(self, params, dev):
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += _ceil(p.numel(), 64)
        # GEMM weights (92 % of the buffer) are fully overwritten by their wgrad kernel; only atomically accumulated gradients
        # (norm / bias / qk-norm vectors, embeddings) and the alignment gaps need zeroing.
        flat = torch.empty(total, dtype=F32, device=dev)
        gemm_w = {id(l.weight) for l in self._lins.values()}
        lo, gaps = None, []
        for p, o in zip(params, offs):
            end = o + _ceil(p.numel(), 64)
            if id(p) in gemm_w:
                if lo is not None:
                    gaps.append(flat[lo:o])
                    lo = None
                if end > o + p.numel():
                    gaps.append(flat[o + p.numel():end])
            elif lo is None:
                lo = o
        if lo is not None:
            gaps.append(flat[lo:total])
        if gaps:   # ~70 short ranges: one multi-tensor launch instead of one fill kernel each
            torch._foreach_zero_(gaps)
        self._grad_ranges = {id(p): (o, o + _ceil(p.numel(), 64)) for p, o in zip(params, offs)}
        return flat, {id(p): flat[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs)}

    def _notify(self, flat, group):
        """Tell the gradient-sync hook that the flat range covering `group` (contiguous by construction) is final."""
        cb = self.grad_ready_callback
        if cb is None or not group:
            return
        lo = min(self._grad_ranges[id(p)][0] for p in group)
        hi = max(self._grad_ranges[id(p)][1] for p in group)
        cb(flat, lo, hi)

    def _wgrad(self, dY, X, lin: _Lin, G, n_rows=None, bias_done=False, beta=0.0):
        """dW[out,in] = dY[M,out]^T X[M,in] (fp32), db = colsum(dY).  dY/X bf16 [M, *]."""
        Mrows = dY.shape[0]
        outp = dY.shape[1]
        db = None
        if lin.bias is not None and not bias_done:
            db = G[id(lin.bias)] if outp == lin.out else torch.zeros(outp, dtype=F32, device=dY.device)
        few_tiles = K.gemm_tn_wants_splitk(lin.out, lin.inp)
        if Mrows % 64 == 0:  # K-major GEMM reads dY and X in place (transposing LDS reads); bias grad = column sums
            if db is not None:
                K.colsum(dY, db)
            if few_tiles:  # e.g. the 2048 x 2048 out-proj weight: 64 tiles over K = B*L -> split K through a workspace
                K.gemm_tn_splitk(dY, X, G[id(lin.weight)], M=lin.out, N=lin.inp, beta=beta)
            else:
                K.gemm_tn(dY, X, G[id(lin.weight)], M=lin.out, N=lin.inp, beta=beta)
        else:  # short contraction (e.g. adaLN over the padded batch): explicit transposes + NT kernel
            dYt = K.transpose(dY, colsum=db)
            Xt = K.transpose(X)
            K.gemm_nt(dYt, Xt, out=G[id(lin.weight)], M=lin.out, N=lin.inp, K=Mrows, beta=beta)
        if db is not None and outp != lin.out:
            if beta != 0.0:
                G[id(lin.bias)].add_(db[: lin.out])
            else:
                G[id(lin.bias)].copy_(db[: lin.out])

    def _engine_backward(self, S, grad_out, mode):
        params = self._ordered_params()
        B, L = S["B"], S["L"]
        M, d, H, D = B * L, self.hidden_size, self.n_heads, self.head_dim
        dev = S["ids"].device
        nt = K.norm_id(self.norm_type)
        tc, sw = self.time_conditioning, self.sandwich_normalization
        lin = self._lins
        flat, G = self._alloc_grads(params, dev)
        S["grad_flat"] = flat
        mod_flat, any_img, p_drop, seed0 = S["modality"], S.get("any_img"), S["p_drop"], S["seed0"]
        V = self.vocab_size
        head = lin["head"]
        logits = S["logits"]

        # ---- head: d logits -> dhf, dW_head, db_head
        head_rows = S.get("head_rows")
        chunk = S.get("head_chunk", 0) if mode == "logp" else 0
        if mode == "logp":
            g = grad_out.contiguous().view(-1).to(F32)
            if head_rows is not None:
                g = g.index_select(0, head_rows[0])
        if chunk:   # fused head + cross-entropy: per row chunk recompute the logits, form d logits in place, consume them (dgrad, wgrad accumulated)
            hf_h = S["hf"]
            Mh, Vp = hf_h.shape[0], head.outp
            dhf = torch.empty((M, d), dtype=BF16, device=dev)[:Mh]
            buf = torch.empty((chunk, Vp), dtype=BF16, device=dev)
            cm = S["ce_mod"]
            for r0 in range(0, Mh, chunk):
                r1 = min(Mh, r0 + chunk)
                lg = buf[: r1 - r0]
                K.gemm_nt(hf_h[r0:r1], head.w16, out=lg, N=V, epilogue=K.EPI_BIAS, bias=head.bias.detach())
                K.subs_ce_bwd(lg, S["x0"][r0:r1], S["ids_h"][r0:r1], cm[r0:r1] if cm is not None else None, S["lse_ce"][r0:r1], g[r0:r1], V, self.text_vocab_size,
                              self.mask_index, S["restrict"])
                K.gemm_nt_splitk(lg, head.w16t, N=d, out=dhf[r0:r1])
                self._wgrad(lg, hf_h[r0:r1], head, G, beta=0.0 if r0 == 0 else 1.0)
            del buf
            dlogits = None
        else:
            if mode == "logp":
                K.subs_ce_bwd(logits, S["x0"], S["ids_h"], S["ce_mod"], S["lse_ce"], g, V, self.text_vocab_size, self.mask_index, S["restrict"],
                              narrow_txt_rows=S["head_groups"][0] if S.get("head_groups") is not None else -1)
                dlogits = logits
            else:
                dlogits = torch.zeros_like(logits)
                dlogits[:, :V].copy_(grad_out.reshape(M, V))
            groups = S.get("head_groups") if mode == "logp" else None
            if groups is not None:   # the same two (rows of one modality) x (ids of that modality) problems as the forward; d logits is zero everywhere else
                n_tp, n_ip = groups
                Vt, Vp = self.text_vocab_size, head.outp
                c_al, c_up = Vt // 8 * 8, _ceil(Vt, 8)
                k_t = min(_ceil(Vt, 64), Vp)
                hf_h = S["hf"]
                dhf = torch.empty((M, d), dtype=BF16, device=dev)[: dlogits.shape[0]]
                Gw = G[id(head.weight)]
                db = torch.zeros(Vp, dtype=F32, device=dev)   # (column sums over widths that are multiples of 8; d logits is zero in the padding columns)
                if n_tp:
                    K.gemm_nt_splitk(dlogits[:n_tp, :k_t], head.w16t[:, :k_t], N=d, out=dhf[:n_tp])
                    K.colsum(dlogits[:n_tp, :c_up], db[:c_up])
                    if c_al:
                        K.gemm_tn(dlogits[:n_tp, :c_al], hf_h[:n_tp], Gw[:c_al], M=c_al, N=d)
                elif c_al:
                    Gw[:c_al].zero_()
                if n_ip:
                    K.gemm_nt_splitk(dlogits[n_tp:, c_al:], head.w16t[:, c_al:], N=d, out=dhf[n_tp:])
                    K.colsum(dlogits[n_tp:, c_al:], db[c_al:])
                    K.gemm_tn(dlogits[n_tp:, c_up:V], hf_h[n_tp:], Gw[c_up:], M=V - c_up, N=d)
                else:
                    Gw[c_up:].zero_()
                if c_up > c_al:   # the ids between the two aligned boundaries (at most 15) take every row
                    K.gemm_tn(dlogits[:, c_al:c_up], hf_h, Gw[c_al:c_up], M=c_up - c_al, N=d)
                G[id(head.bias)].copy_(db[:V])
            else:
                # few output tiles (compacted rows x d) over K = V: split K so that all CUs work (falls back to the plain kernel otherwise)
                dhf = K.gemm_nt_splitk(dlogits, head.w16t, N=d, out=torch.empty((M, d), dtype=BF16, device=dev)[: dlogits.shape[0]])
                self._wgrad(dlogits, S["hf"], head, G)
        stream_compact = bool(S.get("stream_compact"))
        if head_rows is not None and not stream_compact:  # scatter the masked rows' gradient back; every other row of d(final norm output) is exactly zero
            rows_p, n_masked = head_rows
            dhf = torch.zeros((M, d), dtype=dhf.dtype, device=dev).index_copy_(0, rows_p, dhf)   # (padding rows: d logits = 0, so their rows of dhf are 0 too)
        del dlogits
        S["logits"] = None

        dc = dmodf = None
        Bp = S.get("Bp")
        if tc:
            dc = torch.zeros((Bp, self.cond_dim), dtype=F32, device=dev)
            dmodf = torch.zeros((Bp, 2 * d), dtype=F32, device=dev)
            # the adaLN_modulation backwards leave their input gradients as partial tiles in one buffer; summed once, where dc is consumed (`_ada_collect`)
            self._ada_buf, self._ada_off = None, 0
            if Bp <= K.SMALL_BATCH_LINEAR_MAX_B and self.cond_dim <= K.SMALL_BATCH_LINEAR_MAX_IN and self.cond_dim % 8 == 0:
                tiles = K.small_batch_linear_bwd_tiles(2 * d) + self.n_blocks * K.small_batch_linear_bwd_tiles(6 * d)
                self._ada_buf = torch.empty((tiles, Bp, self.cond_dim), dtype=F32, device=dev)
        fl = self.output_layer
        fmod = S["fmod"]
        # (compacted last block: its residual-stream gradient lives on the compacted rows until the block's attention is reached)
        dx = torch.empty((dhf.shape[0] if stream_compact else M, d), dtype=F32, device=dev)
        # Without adaLN every pre-norm backward is immediately followed by the backward of the residual branch that produced the norm's input, on
        # the dx it has just updated: the pair runs as ONE fused pass (K.norm_residual_bwd).  `pend` carries the norm half to the branch that
        # consumes it - the final norm and each block's norm1 pair with the MLP branch of the block below them in the schedule, so a block's
        # gradient range is reported (and its activations dropped) only after that fused pass.
        pend = None
        # adaLN-Zero (round 5): the same pairing with the modulated / gated forms of the fused pass where the kernel covers the shape (K.norm_residual_bwd_ada)
        tc_fused = tc and not stream_compact and K.norm_residual_bwd_ada_ok(M, d, L)
        if tc:
            K.norm_bwd(dhf, S["x_final"], S["rstdf"], S["meanf"], fl.norm_final.weight.detach(), nt, L, dx, G[id(fl.norm_final.weight)], accumulate=False,
                       mod=fmod, dmod=dmodf, mod_idx=(0, 1), modality=mod_flat, any_img=any_img)
            self._ada_backward(dmodf, lin["head.ada"], S["c"], dc, G)
            self._notify(flat, list(fl.parameters()))
        else:
            pend = dict(dy=dhf, x=S["x_final"], rstd=S["rstdf"], mean=S["meanf"], w=fl.norm_final.weight.detach(), dw=G[id(fl.norm_final.weight)], accumulate=False,
                        done=lambda: self._notify(flat, list(fl.parameters())))

        def branch_bwd(pend, branch, **kw):
            """residual-branch backward, fused with the pending norm backward when there is one (dbias: += column sums of the result)"""
            if pend is None:
                dbias = kw.pop("dbias", None)
                out = K.residual_bwd(dx, branch, L, **kw)
                if dbias is not None:
                    K.colsum(out, dbias)
                return out
            if pend.get("mod") is not None or kw.get("gate_idx") is not None:   # adaLN-Zero: modulated norm and / or gated branch, still one pass per row
                out = K.norm_residual_bwd_ada(pend["dy"], pend["x"], pend["rstd"], pend["mean"], pend["w"], nt, L, dx, pend["dw"], branch, accumulate=pend["accumulate"],
                                              w_b=kw.get("w_b"), rstd_b=kw.get("rstd"), mean_b=kw.get("mean"), dw_b=kw.get("dw_b"), p_drop=kw.get("p_drop", 0.0),
                                              seed=kw.get("seed", 0), dbias=kw.get("dbias"), mod_n=pend.get("mod"), dmod_n=pend.get("dmod"),
                                              mod_idx=pend.get("mod_idx", (0, 1)), modality=mod_flat, any_img=any_img, mod_r=kw.get("mod"), dmod_r=kw.get("dmod"),
                                              gate_idx=kw.get("gate_idx"), modality_r=kw.get("modality"))
            else:
                out = K.norm_residual_bwd(pend["dy"], pend["x"], pend["rstd"], pend["mean"], pend["w"], nt, L, dx, pend["dw"], branch, accumulate=pend["accumulate"],
                                          w_b=kw.get("w_b"), rstd_b=kw.get("rstd"), mean_b=kw.get("mean"), dw_b=kw.get("dw_b"), p_drop=kw.get("p_drop", 0.0),
                                          seed=kw.get("seed", 0), dbias=kw.get("dbias"))
            pend["done"]()
            return out

        for i in reversed(range(self.n_blocks)):
            blk, R = self.blocks[i], S["blocks"][i]
            if R.get("ckpt"):   # activation checkpointing: rebuild this block's activations from its saved input, right before they are consumed
                with torch.no_grad():
                    R = S["block_fwd"](i, R["x_in"], None, last_rows=R["rows_c"], recompute=True)[2]
            at = blk.attention
            mod = R.get("mod")
            dmod = torch.zeros((Bp, 6 * d), dtype=F32, device=dev) if tc else None
            f1, f2 = lin[f"{i}.fc1"], lin[f"{i}.fc2"]
            # MLP branch
            if tc:   # (pend: the modulated norm1 of block i + 1 when the fused adaLN pass is in use)
                du2 = branch_bwd(pend, R["u2"], w_b=blk.post_ff_norm.weight.detach() if sw else None, rstd=R["rstd_m"], mean=R["mean_m"], norm_type=nt,
                                 mod=mod, dmod=dmod, gate_idx=5, modality=mod_flat, dw_b=G[id(blk.post_ff_norm.weight)] if sw else None,
                                 p_drop=p_drop, seed=seed0 + 4 * i + 2)
                pend = None
            else:
                # (the mlp.2 bias gradient = column sums of du2 comes out of the same pass)
                du2 = branch_bwd(pend, R["u2"], w_b=blk.post_ff_norm.weight.detach() if sw else None, rstd=R["rstd_m"], mean=R["mean_m"], norm_type=nt,
                                 dw_b=G[id(blk.post_ff_norm.weight)] if sw else None, p_drop=p_drop, seed=seed0 + 4 * i + 2, dbias=G[id(f2.bias)])
                pend = None
            # dgrad through mlp.2 with the GELU' multiply and the mlp.0 bias gradient (column sums of du1) fused into the epilogue
            du1 = K.gemm_nt(du2, f2.w16t, N=4 * d, epilogue=K.EPI_DGELU, aux=R["u1"], bias=G[id(f1.bias)])
            lo, lq = lin[f"{i}.out"], lin[f"{i}.qkv"]
            # few-tile regime (every weight of the block is at most half a round of tiles: UniDisc-S): the four weight gradients wait for the end of the block's
            # backward and share ONE split-K launch + ONE reduce (K.gemm_tn_multi) instead of three split-K launches + four reduce passes
            multi = None
            if (self.multi_wgrads and du2.is_cuda and not tc and R.get("rows_c") is None and M % 64 == 0 and lo.bias is None and lq.bias is None
                    and all(K.gemm_tn_wants_splitk(l_.out, l_.inp) and l_.out % 256 == 0 and l_.inp % 256 == 0 for l_ in (f1, f2, lo, lq))):
                multi = [(du2, R["g"], f2), (du1, R["h2"], f1)]
            if multi is None:
                self._wgrad(du2, R["g"], f2, G, bias_done=not tc)
            dh2 = f1.dgrad(du1, du1.shape[0], S["dgrad_form"].get(f"{i}.fc1"))
            if multi is None:
                self._wgrad(du1, R["h2"], f1, G, bias_done=True)
            del du1, du2
            # norm2 backward + attention branch
            if tc_fused:
                p2 = dict(dy=dh2, x=R["x_mid"], rstd=R["rstd2"], mean=R["mean2"], w=blk.norm2.weight.detach(), dw=G[id(blk.norm2.weight)], accumulate=True,
                          done=lambda: None, mod=mod, dmod=dmod, mod_idx=(3, 4))
                if sw:
                    da = branch_bwd(p2, R["a_out"], w_b=blk.pre_residual_norm.weight.detach(), rstd=R["rstd_a"], mean=R["mean_a"], norm_type=nt,
                                    dw_b=G[id(blk.pre_residual_norm.weight)])
                else:
                    da = branch_bwd(p2, R["a_out"], mod=mod, dmod=dmod, gate_idx=2, p_drop=p_drop, seed=seed0 + 4 * i + 1)
            elif tc:
                K.norm_bwd(dh2, R["x_mid"], R["rstd2"], R["mean2"], blk.norm2.weight.detach(), nt, L, dx, G[id(blk.norm2.weight)], accumulate=True,
                           mod=mod, dmod=dmod, mod_idx=(3, 4), modality=mod_flat, any_img=any_img)
                if sw:
                    da = K.residual_bwd(dx, R["a_out"], L, w_b=blk.pre_residual_norm.weight.detach(), rstd=R["rstd_a"], mean=R["mean_a"], norm_type=nt,
                                        dw_b=G[id(blk.pre_residual_norm.weight)])
                else:
                    da = K.residual_bwd(dx, R["a_out"], L, mod=mod, dmod=dmod, gate_idx=2, p_drop=p_drop, seed=seed0 + 4 * i + 1)
            else:
                p2 = dict(dy=dh2, x=R["x_mid"], rstd=R["rstd2"], mean=R["mean2"], w=blk.norm2.weight.detach(), dw=G[id(blk.norm2.weight)], accumulate=True,
                          done=lambda: None)
                if sw:
                    da = branch_bwd(p2, R["a_out"], w_b=blk.pre_residual_norm.weight.detach(), rstd=R["rstd_a"], mean=R["mean_a"], norm_type=nt,
                                    dw_b=G[id(blk.pre_residual_norm.weight)])
                else:
                    da = branch_bwd(p2, R["a_out"], p_drop=p_drop, seed=seed0 + 4 * i + 1)
            pair_out = None
            do = lo.dgrad(da, da.shape[0], S["dgrad_form"].get(f"{i}.out"))
            if R.get("rows_c") is not None:
                # back to all rows: the rows left out have a zero gradient in both the attention output and the residual stream
                self._wgrad(da, R["o_c"], lo, G)
                rows_c = R["rows_c"]
                do = torch.zeros((M, d), dtype=do.dtype, device=dev).index_copy_(0, rows_c, do)
                dx = torch.zeros((M, d), dtype=F32, device=dev).index_copy_(0, rows_c, dx)
            else:
                # The out-proj weight gradient (2048 x 2048: 64 tiles) waits for the qkv one (6144 x 2048: 192 tiles of 256 rows) where the two fill the chip
                # exactly once TOGETHER (K.gemm_tn_pair): one launch instead of 256 tiles of 192 rows + a split-K launch + its reduce pass; few tiles over a long
                # contraction (UniDisc-S: 27 + 9) share ONE split-K launch
                if multi is not None:
                    multi.append((da, R["o"], lo))
                elif (self.pair_wgrads and da.is_cuda and lo.bias is None and lq.bias is None and lo.inp == lq.inp
                        and K.gemm_tn_pair_ok(lq.out, lo.out, lq.inp, M)
                        and ((lq.out // 256 + lo.out // 256) * (lq.inp // 256) % 256 == 0 or (lq.out // 256 + lo.out // 256) * (lq.inp // 256) <= 128)):
                    pair_out = (da, R["o"])
                else:
                    self._wgrad(da, R["o"], lo, G)
            dqkr = torch.empty((M, 2 * d), dtype=BF16, device=dev)
            dqkv = torch.empty((M, 3 * d), dtype=BF16, device=dev)
            K.attention_bwd(R["qkr"], R["qkv"], R["o"], do, R["lse"], dqkr, dqkv, B, L, H, D, S["sid"], S["doc_ranges"], q_prescaled=True)
            qn = self.qk_norm
            K.qknorm_rope_bwd(dqkr, R["qkv"], dqkv, S["cos"], S["sin"], L, D, gq=at.q_norm.weight.detach() if qn else None,
                              gk=at.k_norm.weight.detach() if qn else None, stats=R["qstats"], dgq=G[id(at.q_norm.weight)] if qn else None,
                              dbq=G[id(at.q_norm.bias)] if qn else None, dgk=G[id(at.k_norm.weight)] if qn else None,
                              dbk=G[id(at.k_norm.bias)] if qn else None, q_scale=self.attn_q_scale)
            dh1 = lq.dgrad(dqkv, dqkv.shape[0], S["dgrad_form"].get(f"{i}.qkv"))
            if multi is not None:
                multi.append((dqkv, R["h1"], lq))
                if not K.gemm_tn_multi([(dy_, x_, G[id(l_.weight)]) for dy_, x_, l_ in multi]):
                    for dy_, x_, l_ in multi:    # (shapes outside the shared launch: the plain calls)
                        self._wgrad(dy_, x_, l_, G, bias_done=True)
                multi = None
            elif pair_out is not None:
                K.gemm_tn_pair(dqkv, R["h1"], G[id(lq.weight)], pair_out[0], pair_out[1], G[id(lo.weight)])
                pair_out = None
            else:
                self._wgrad(dqkv, R["h1"], lq, G)
            if tc_fused:   # norm1 (modulated) pairs with the gated MLP branch of block i - 1; this block's adaLN_modulation backward needs the shift / scale
                def done_tc(i=i, blk=blk, dmod=dmod):   # gradients of that pass, so it waits for it
                    self._ada_backward(dmod, lin[f"{i}.ada"], S["c"], dc, G)
                    S["blocks"][i] = None
                    self._notify(flat, list(blk.parameters()))
                pend = dict(dy=dh1, x=R["x_in"], rstd=R["rstd1"], mean=R["mean1"], w=blk.norm1.weight.detach(), dw=G[id(blk.norm1.weight)], accumulate=True,
                            done=done_tc, mod=mod, dmod=dmod, mod_idx=(0, 1))
            elif tc:
                K.norm_bwd(dh1, R["x_in"], R["rstd1"], R["mean1"], blk.norm1.weight.detach(), nt, L, dx, G[id(blk.norm1.weight)], accumulate=True,
                           mod=mod, dmod=dmod, mod_idx=(0, 1), modality=mod_flat, any_img=any_img)
                self._ada_backward(dmod, lin[f"{i}.ada"], S["c"], dc, G)
                S["blocks"][i] = None  # free this block's activations
                self._notify(flat, list(blk.parameters()))
            else:   # norm1 pairs with the MLP branch of block i - 1: keep what it needs alive, report this block's range after that pass
                def done(i=i, blk=blk):
                    S["blocks"][i] = None
                    self._notify(flat, list(blk.parameters()))
                pend = dict(dy=dh1, x=R["x_in"], rstd=R["rstd1"], mean=R["mean1"], w=blk.norm1.weight.detach(), dw=G[id(blk.norm1.weight)], accumulate=True, done=done)
        if pend is not None:   # block 0's norm1 (or the final norm of a model without blocks): nothing below it to pair with
            K.norm_bwd(pend["dy"], pend["x"], pend["rstd"], pend["mean"], pend["w"], nt, L, dx, pend["dw"], accumulate=pend["accumulate"], mod=pend.get("mod"),
                       dmod=pend.get("dmod"), mod_idx=pend.get("mod_idx", (0, 1)), modality=mod_flat if pend.get("mod") is not None else None,
                       any_img=any_img if pend.get("mod") is not None else None)
            pend["done"]()

        # ---- embeddings
        if S.get("cnt_j") is not None:
            n_cnt = self.img_count_embedding.shape[0]
            if dx.is_cuda and dx.dtype == F32:   # sums formed in LDS per run of equal indices (an index_add_ here is 19 M contended atomics: 225 us)
                K.rowgroup_sum(dx, S["cnt_j"], G[id(self.img_count_embedding)])
            else:
                gext = torch.zeros((n_cnt + 1, dx.shape[1]), dtype=dx.dtype, device=dx.device).index_add_(0, S["cnt_j"], dx)   # (row n_cnt collects the positions without an image index)
                G[id(self.img_count_embedding)].add_(gext[:n_cnt])
        K.embedding_bwd(S["ids"], dx, G[id(self.vocab_embed.embedding)], self.mask_index,
                        modality=S["emb_mod"] if self.modality_embed is not None else None,
                        dEm=G[id(self.modality_embed.embedding)] if self.modality_embed is not None else None)
        tail = list(self.vocab_embed.parameters()) + (list(self.modality_embed.parameters()) if self.modality_embed is not None else [])
        if getattr(self, "img_count_embedding", None) is not None:
            tail.append(self.img_count_embedding)
        if tc:  # c = silu(W2 silu(W0 te + b0) + b2)
            dc16 = torch.empty((Bp, self.cond_dim), dtype=BF16, device=dev)
            self._ada_collect(dc)
            K.cast_f32_bf16(dc, dc16)
            dl2 = K.silu_bwd(S["l2"], dc16)
            ds1 = K.gemm_nt(dl2, lin["sig2"].w16t, N=lin["sig2"].inp)
            self._wgrad(dl2, S["s1"], lin["sig2"], G)
            dl1 = K.silu_bwd(S["l1"], ds1)
            self._wgrad(dl1, S["te"], lin["sig0"], G)
            tail += list(self.sigma_map.parameters())
        self._notify(flat, tail)
        if self.grad_sync_finish is not None:
            self.grad_sync_finish()
        # for the fused optimizer's one-pass global norm; a weak reference: the buffer lives exactly as long as the p.grad views do
        self._last_grad_flat_ref, self._last_grad_numel = weakref.ref(flat), sum(p.numel() for p in params)
        return [G[id(p)] for p in params]

    def _ada_backward(self, dmod, lin: _Lin, c, dc, G):
        """adaLN_modulation backward: dmod fp32 [Bp, n*d] (atomically accumulated) -> dW, db, dc += dmod W."""
        buf = getattr(self, "_ada_buf", None)
        if buf is not None and lin.w16 is not None:
            # one launch (round 5): the input gradient as a GEMM is M = Bp rows, ONE output tile, K = 6 d - a single workgroup walking 12 288 k took ~140 us per block
            tiles = K.small_batch_linear_bwd_tiles(lin.out)
            parts = buf[self._ada_off:self._ada_off + tiles]
            self._ada_off += tiles
            K.small_batch_linear_bwd(dmod, c, lin.w16, G[id(lin.weight)], G[id(lin.bias)] if lin.bias is not None else None, dx_parts=parts)
            return
        dmod16 = torch.empty(dmod.shape, dtype=BF16, device=dmod.device)
        K.cast_f32_bf16(dmod, dmod16)
        self._wgrad(dmod16, c, lin, G)
        K.gemm_nt(dmod16, lin.w16t, out=dc, N=lin.inp, beta=1.0)


    def _ada_collect(self, dc):
        """dc += the partial input-gradient tiles this pass's adaLN_modulation backwards left behind"""
        buf, off = getattr(self, "_ada_buf", None), getattr(self, "_ada_off", 0)
        self._ada_buf, self._ada_off = None, 0
        if buf is not None and off:
            dc += buf[:off].sum(0)
        return dc


class _DitFn(torch.autograd.Function):
    """One autograd node for the whole backbone (+ fused SUBS cross-entropy in "logp" mode)."""

    @staticmethod
    def forward(ctx, module: DIT, mode: str, inputs: dict, *params):
        need = bool(inputs.get("save", False))  # grad mode is off inside Function.forward, so the caller decides
        out, S = module._engine_forward(inputs, mode, save=need)
        ctx.module, ctx.mode, ctx.S = module, mode, S if need else None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.S is None:
            raise RuntimeError("unidisc_amd.DIT: backward called but activations were not saved")
        grads = ctx.module._engine_backward(ctx.S, grad_out, ctx.mode)
        ctx.S = None
        req = [p.requires_grad for p in ctx.module._ordered_params()]
        return (None, None, None, *[g if r else None for g, r in zip(grads, req)])
