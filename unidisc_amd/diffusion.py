"""Trainer-level hot path: the reference's ``model.py::Diffusion`` methods that run every training step.

Mirrors, with the same names / argument meaning / error behaviour:
  ``update_batch`` (model.py:157-395, token-dataset branch), ``get_cond_dict`` (:397-418), ``training_step`` (:420-422),
  ``q_xt`` (:424-587), ``_sample_t`` (:589-619), ``_subs_parameterization`` (:621-658), ``_process_sigma`` (:660-672),
  ``forward`` (:674-795), ``compute_loss`` (:797-1173) and the ``Loss`` record (model_utils.py:110-120).

Host logic (RNG draws, masks, weights, reductions on [B] / [B,L] tensors) is plain torch on the device — the
reference's draw order ``rand(B)`` → ``rand(B,L)`` → ``rand(B,1)``×2 is preserved so masks are bit-exact for a
given generator state.  All O(B·L·d) and O(B·L·V) work goes through ``unidisc_amd.DIT`` (HIP kernels).
``compute_loss`` uses the fused path ``backbone.forward_logp`` and never materialises [B,L,V] log-probs;
``forward`` still returns full SUBS log-probs (or logits) for samplers.
"""
from __future__ import annotations

import math
import os

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from . import kernels as K
from .dit import DIT, ModalityMask, cfg_get
from .noise_schedule import LogLinearNoise, get_noise


@dataclass
class Loss:  # model_utils.py:110-120
    loss: torch.FloatTensor
    img_loss: torch.FloatTensor = None
    txt_loss: torch.FloatTensor = None
    nlls: torch.FloatTensor = None
    token_mask: torch.FloatTensor = None
    txt_nlls: torch.FloatTensor = None
    img_nlls: torch.FloatTensor = None
    extra_losses: dict = None
    modality_mask: torch.FloatTensor = None


class _LinearLoss(torch.autograd.Function):
    """loss(log_p) with a precomputed value and gradient coefficient: the diffusion loss is linear in the per-token log-probabilities"""

    @staticmethod
    def forward(ctx, log_p, coef, value):
        ctx.save_for_backward(coef)
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        (coef,) = ctx.saved_tensors
        return (g * coef).to(coef.dtype), None, None


class Diffusion:
    def __init__(self, config, tokenizer, device, disable_init=False, backbone: Optional[torch.nn.Module] = None):
        self.config, self.tokenizer, self.device = config, tokenizer, torch.device(device)
        if disable_init:
            return
        self.init(config, tokenizer, device, backbone)

    # ---- model_setup.init (model_setup.py:47-327), hot-path subset
    def init(self, config, tokenizer, device, backbone=None):
        m, tr = cfg_get(config, "model"), cfg_get(config, "trainer")
        self.global_step = 0
        self.current_run_fwd_bwd_pass = 0
        prec = str(cfg_get(tr, "precision", "bf16"))
        self.dtype = torch.float32 if ("fp32" in prec or "no" in prec) else (torch.bfloat16 if "bf16" in prec else torch.float16)
        if self.dtype != torch.bfloat16:
            raise NotImplementedError(f"unidisc_amd: trainer.precision={prec}; only bf16 (the reference's training precision) is implemented in HIP")
        # trainer options that change the loss / the batch and are not built: refuse them instead of training silently with another objective
        for flag, why in (("ar_llm_loss", "the extra auto-regressive LLM loss term (model.py:1076-1136)"),
                          ("force_remove_img_tokens", "dropping image tokens from the batch (model.py:316-319, :1054)"),
                          ("add_label", "the class-label token written by update_batch (model.py:321-334)")):
            if cfg_get(tr, flag, False):
                raise NotImplementedError(f"unidisc_amd: trainer.{flag} — {why} — is not on the denoising hot path (SURVEY.md §8)")
        # trainer.low_precision_loss (model.py:747, :924) keeps the SUBS log-probabilities in the autocast dtype instead of widening them to fp32 before the gather.  Under
        # bf16 autocast they ARE bf16 values either way (widening is exact) and every later product meets an fp32 weight, so the reference's loss is the same number with
        # the flag on or off; this path takes log p from an fp32 log-sum-exp in both cases.  Accepted, nothing to switch.
        self.image_model = bool(cfg_get(m, "image_model", False))
        self.unified_model = bool(cfg_get(m, "unified_model", False))
        self.antithetic_sampling = cfg_get(tr, "antithetic_sampling", True)
        self.importance_sampling = cfg_get(tr, "importance_sampling", False)
        self.change_of_variables = cfg_get(tr, "change_of_variables", False)
        if self.importance_sampling or self.change_of_variables:
            raise NotImplementedError("unidisc_amd: importance_sampling / change_of_variables are off in every shipped config and not implemented")
        if self.image_model is False or self.unified_model:  # model_setup.py:89-98
            forced = cfg_get(m, "force_text_vocab_size", None)
            self.vocab_size = forced if forced is not None else len(tokenizer)
            if tokenizer is None or getattr(tokenizer, "mask_token", None) is None:
                self.mask_index = self.vocab_size
                self.vocab_size += 1
            else:
                self.mask_index = tokenizer.mask_token_id
        if self.image_model:  # :99-113
            if self.unified_model:
                self.text_vocab_size = self.vocab_size
                self.vocab_size += cfg_get(m, "image_vocab_size")
                self.image_vocab_size = cfg_get(m, "image_vocab_size")
            else:
                self.vocab_size = cfg_get(m, "image_vocab_size") + 1
                self.mask_index = self.vocab_size - 1
                self.text_vocab_size = 0
        else:
            self.text_vocab_size = self.vocab_size
        self.parameterization = cfg_get(config, "parameterization", "subs")
        if self.parameterization != "subs":
            raise NotImplementedError(f"unidisc_amd: parameterization={self.parameterization}; only SUBS is on the denoising hot path")
        if cfg_get(config, "backbone", "dit") != "dit":
            raise NotImplementedError("unidisc_amd: only backbone=dit is implemented")
        self.static_txt_sl = slice(None, cfg_get(m, "txt_length"))
        self.static_img_sl = slice(-cfg_get(m, "img_length"), None) if cfg_get(m, "img_length", 0) else slice(0, 0)
        if backbone is None:  # model_setup.py:148-161
            backbone = DIT(config, vocab_size=self.vocab_size, text_vocab_size=self.text_vocab_size, mask_index=self.mask_index, autocast_dtype=self.dtype,
                           device=device, static_img_sl=self.static_img_sl, static_txt_sl=self.static_txt_sl)
        self.backbone = backbone
        self.T = cfg_get(config, "T", 0)
        if self.T:
            raise NotImplementedError("unidisc_amd: discrete-time T>0 (D3PM loss) is not on the denoising hot path")
        self.noise = get_noise(config)
        self.sampling_eps = cfg_get(tr, "sampling_eps", 1e-3)
        self.time_conditioning = cfg_get(config, "time_conditioning", False)
        self.neg_infinity = -1000000.0  # model_setup.py:269

    rng_device = None  # set to "cpu" to draw t / masks from the CPU generator (bit-reproducible across devices)
    _generic_qxt = False   # tests: route q_xt through the tensor statements even where the fused launch applies

    def _rand(self, *shape, device):
        if self.rng_device is None:
            return torch.rand(*shape, device=device)
        return torch.rand(*shape, device=self.rng_device).to(device)

    @property
    def training(self):
        return self.backbone.training

    @property
    def allow_slicing(self):  # model.py:1283-1285
        return not self.backbone.training

    def txt_sl(self, batch=None):
        return batch["modality_mask"][..., 0]

    def img_sl(self, batch=None):
        return batch["modality_mask"][..., 1]

    # ---- the batch contract of model.py:157-395 (token-dataset and pre-tokenised branches), in this module's own structure
    def update_batch(self, batch):
        """Dataset batch -> model batch.  Output keys and values are the reference's (`input_ids`, `attention_mask`, `modality`, `modality_mask`,
        `batch_contains_img`, `txt_sl`, `img_sl`, `sample_ids`), produced in three stages:
          1. everything onto the device in the DATASET's dtypes (80 KiB per step at 1.4 B), so that
          2. the joint sequence comes from ONE gather launch (`_assemble_on_device`, tokens.hip) wherever the batch has the token-dataset shape, and from
             `_joint_sequence_generic` (slice writes into preallocated tensors) for the shapes the kernel does not cover;
          3. modality / attention-mask / sample-id fields derived without in-place edits of the caller's tensors."""
        if batch is None:
            return batch
        cfg, tr, m = self.config, cfg_get(self.config, "trainer"), cfg_get(self.config, "model")
        data = cfg_get(cfg, "data")
        out = {}
        for key, val in (batch.items() if not isinstance(batch, dict) else dict(batch).items()):
            if isinstance(val, (list, tuple)) and val and all(isinstance(v, torch.Tensor) for v in val) and key in ("img_input_ids", "txt_input_ids", "sample_ids"):
                val = torch.stack(list(val), dim=0)
            out[key] = val.to(self.device) if isinstance(val, torch.Tensor) else val
        fused = False
        if self.image_model or cfg_get(data, "force_image_dataset", False):
            has_token_fields = "txt_input_ids" in out or "img_input_ids" in out
            assembled = "input_ids" in out and "modality" in out
            if "img" in out and not has_token_fields:
                raise NotImplementedError("unidisc_amd: raw-image batches need the VQ tokenizer, which is outside the denoising hot path")
            if has_token_fields:
                fused = self._assemble_on_device(out, tr, m)
                if not fused:
                    self._joint_sequence_generic(out, tr, m)
            elif cfg_get(tr, "multimodal_batches", False) or assembled:
                # pre-tokenised multimodal batches, and batches token_data.TokenBatcher assembled on the device already
                out["input_ids"] = out["input_ids"].long()
                if cfg_get(tr, "force_shift_image_batches", False):
                    out["input_ids"] = out["input_ids"] + (out["modality"] == 1).long() * self.text_vocab_size
            else:
                raise NotImplementedError("unidisc_amd: raw-image batches need the VQ tokenizer, which is outside the denoising hot path")
            if out["input_ids"].shape[1] != cfg_get(m, "length") and not cfg_get(tr, "ar_inpainting", False):
                raise AssertionError(f"input ids are not the correct length input ids shape: {out['input_ids'].shape}, model length: {cfg_get(m, 'length')}")
        if "sample_ids" in out:
            out["sample_ids"] = out["sample_ids"].long()
        self._modality_fields(out, tr, m, data, checked=fused)
        self._attention_fields(out, tr, data)
        return out

    def _joint_sequence_generic(self, b, tr, m):
        """`input_ids` / `attention_mask` / `modality` of a token batch the gather kernel does not take (host-dtype surprises, text-less or user-supplied
        modality, CPU tests): [text | image + Vt] written into preallocated [B, Lt + Li] tensors."""
        img = b.pop("img_input_ids")
        txt = b.get("txt_input_ids")
        B, Li = img.shape[0], img.shape[-1]
        Lt = txt.shape[-1] if txt is not None else 0
        ids = torch.empty((B, Lt + Li), dtype=torch.int64, device=img.device)
        keep = torch.ones((B, Lt + Li), dtype=torch.bool, device=img.device)
        ids[:, Lt:] = img
        if txt is not None:   # image ids live behind the text vocabulary only in a joint sequence
            b["txt_input_ids"] = txt.long()
            ids[:, Lt:] += self.text_vocab_size
            ids[:, :Lt] = txt
            keep[:, :Lt] = b["txt_attention_mask"]
        b["input_ids"], b["attention_mask"] = ids, keep
        if "modality" not in b:
            if cfg_get(tr, "ignore_text_in_unified", False):
                b["modality"] = torch.ones_like(ids)
            else:
                if not (cfg_get(m, "txt_length") > 0 and cfg_get(m, "img_length") > 0):
                    raise AssertionError("a token batch without `modality` needs model.txt_length > 0 and model.img_length > 0")
                b["modality"] = (torch.arange(Lt + Li, device=ids.device) >= Lt).long().expand(B, -1).contiguous()

    def _modality_fields(self, b, tr, m, data, checked=False):
        mm = cfg_get(tr, "multimodal_batches", False)
        mod = b.get("modality")
        if mod is not None:
            mod = mod.long()
            if mm and mod.ndim == 2 and mod.shape[-1] == 1:    # one modality per sample -> per token
                mod = mod.expand(-1, cfg_get(m, "length")).contiguous()
        elif self.image_model and not mm:                     # static layout: the image occupies the last img_length positions
            mod = torch.zeros_like(b["input_ids"], dtype=torch.int64)
            mod[:, self.static_img_sl] = 1
        elif cfg_get(data, "txt_only", False):
            mod = torch.zeros_like(b["input_ids"], dtype=torch.int64)
        if mod is None:
            return
        if not checked:   # (the gather kernel writes 0 / 1 by construction)
            mod = mod.clamp_min(0)          # padding (-1, PackingCollate) counts as text
            self._check_modality_range(mod)
        b["modality"] = mod
        is_img = mod == 1
        b["modality_mask"] = torch.stack((~is_img, is_img), dim=-1)     # == one_hot(modality, 2).bool() once the range check holds
        b["batch_contains_img"] = is_img.any(dim=-1)
        b["txt_sl"], b["img_sl"] = self.txt_sl(b), self.img_sl(b)

    def _attention_fields(self, b, tr, data):
        am = b["attention_mask"]
        am = torch.ones_like(am, dtype=torch.bool) if cfg_get(tr, "force_full_attention_mask", False) else am.to(torch.bool)
        if cfg_get(data, "require_sample_ids", False):
            if "sample_ids" not in b:
                raise AssertionError("data.require_sample_ids: the batch has no `sample_ids`")
            sid = torch.where(am, b["sample_ids"], torch.full_like(b["sample_ids"], -1))     # padding belongs to no sample ...
            am = am & (sid != -1)                                                             # ... and a position without a sample is padding
            b["sample_ids"] = sid
        b["attention_mask"] = am
        if cfg_get(tr, "interleaved", False) and "sample_ids" not in b:
            b["sample_ids"] = torch.zeros_like(b["modality"], dtype=torch.int64)

    def _assemble_on_device(self, batch, tr, m):
        """model.py:183-212 for a token batch that is already on the device in the dataset's own dtypes (int32 text, int16 image ids, bool text mask): joint ids
        (image ids shifted by the text vocabulary), attention mask (text mask | ones) and modality map in ONE launch (tokens.hip, the TokenBatcher's kernel) instead
        of nine tensor statements.  Same values bit for bit; `txt_input_ids` stays in the batch as int64 like in the reference."""
        txt, img, tm = batch.get("txt_input_ids"), batch.get("img_input_ids"), batch.get("txt_attention_mask")
        ok = (isinstance(txt, torch.Tensor) and isinstance(img, torch.Tensor) and isinstance(tm, torch.Tensor) and txt.is_cuda and img.is_cuda and tm.is_cuda
              and txt.dtype == torch.int32 and img.dtype == torch.int16 and tm.dtype == torch.bool and txt.dim() == 2 and img.dim() == 2 and tm.shape == txt.shape
              and txt.is_contiguous() and img.is_contiguous() and tm.is_contiguous() and txt.shape[0] == img.shape[0]
              and "modality" not in batch and "sample_ids" not in batch and "attention_mask" not in batch and "input_ids" not in batch
              and not cfg_get(tr, "ignore_text_in_unified", False) and cfg_get(m, "txt_length") > 0 and cfg_get(m, "img_length") > 0)
        if not ok:
            return False
        ids, mask, modality = K.assemble_joint_tokens(txt, tm, img, self.text_vocab_size)
        batch.pop("img_input_ids")
        batch["txt_input_ids"] = txt.to(torch.int64)
        batch["input_ids"], batch["attention_mask"], batch["modality"] = ids, mask, modality
        return True

    _checks = ()   # queued device-side batch checks: (event, pinned result)

    def _check_modality_range(self, modality):
        """model.py:311 `assert modality.min() == 0 and modality.max() == 1`.  On the device the reference's form is a host synchronisation at the top of
        EVERY step (the host can never run ahead of the GPU across a step boundary: measured 0.5 ms of idle GPU per 88 ms step at 1.4 B, 2.7 ms per 31 ms
        step at UniDisc-S).  Here the two extrema are queued to pinned memory and the same AssertionError is raised at the step's first natural host wait
        (`_flush_checks`, called after the backbone forward and at the next `update_batch`), before any gradient of that batch is used."""
        self._flush_checks()
        if not modality.is_cuda or os.environ.get("UDM_SYNC_CHECKS") == "1":   # (the knob: the reference's immediate, synchronising form)
            assert modality.min() == 0 and modality.max() == 1
            return
        lo, hi = torch.aminmax(modality)
        host = torch.empty(2, dtype=modality.dtype).pin_memory()
        host.copy_(torch.stack((lo, hi)), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._checks = list(self._checks) + [(ev, host)]

    def _flush_checks(self):
        checks = self._checks
        if not checks:
            return
        self._checks = ()
        for ev, host in checks:
            ev.synchronize()
            assert int(host[0]) == 0 and int(host[1]) == 1, f"batch['modality'] must contain both 0 and 1 and nothing else (min {int(host[0])}, max {int(host[1])})"

    def get_cond_dict(self, batch):  # model.py:397-418
        ret = dict()
        if "cond_input_ids" in batch or "img_label" in batch:
            raise NotImplementedError("unidisc_amd: image / label conditioning is outside the denoising hot path")
        if cfg_get(cfg_get(self.config, "model"), "use_attention_mask", False):   # model.py:405-406: the padding mask goes to the backbone's attention (a key mask)
            ret["attention_mask"] = batch["attention_mask"]
        if cfg_get(cfg_get(self.config, "trainer"), "multimodal_batches", False):
            ret["modality"] = batch["modality"]
        return ret

    def training_step(self, batch, batch_idx):  # model.py:420-422
        batch = self.update_batch(batch)
        return self.compute_loss(batch, prefix="train", batch_idx=batch_idx)

    # ---- forward corruption q(x_t | x_0), absorbing state (contract of model.py:424-587)
    def q_xt(self, x, move_chance, allow_move_mask=None, return_ignore_batch_mask_for_metrics=False, mask_image_square=False, mask_text_region=False,
             batch=None):
        """x_t = [MASK] where a position "moves", x_0 elsewhere.  A position moves when its uniform falls below the sample's move chance; in training a
        sample's whole text (or image) side may be masked instead (`trainer.mask_entire_modality`).  Draw order - rand(B, L), then the per-sample draws, then
        (interleaved) one draw per block - is the reference's, so masks are bit-exact for a given generator state.
        Returns x_t, or (x_t, ignore_batch_mask_for_metrics, None, should_mask_txt, should_mask_img, move_indices)."""
        tr = cfg_get(self.config, "trainer")
        if mask_image_square or mask_text_region:
            raise NotImplementedError("unidisc_amd: square / region masking are evaluation-time options outside the hot path")
        for flag in ("joint_ar_nar_prob", "first_token_dropout"):
            if cfg_get(tr, flag, None) is not None:
                raise NotImplementedError(f"unidisc_amd: trainer.{flag} is not on the denoising hot path")
        if cfg_get(tr, "discrete_diffusion_mode", "absorbing") != "absorbing":
            raise NotImplementedError("unidisc_amd: only absorbing-state diffusion is implemented")
        r_move = self._rand(*x.shape, device=x.device)
        mask_prob = cfg_get(tr, "mask_entire_modality", None)
        whole = mask_prob is not None and self.backbone.training
        multimodal, interleaved = cfg_get(tr, "multimodal_batches", False), cfg_get(tr, "interleaved", False)
        if whole and batch is None:
            raise AssertionError("q_xt: trainer.mask_entire_modality needs the batch (modality masks)")
        # per-sample uniforms of the whole-modality lottery: text first, image second (none for the image side under mask_txt_only)
        r_txt = r_img = None
        p_txt = p_img = 0.0
        if whole:
            r_txt = self._rand(x.shape[0], 1, device=x.device)
            if cfg_get(tr, "mask_txt_only", False):
                p_txt = mask_prob
            else:
                r_img = self._rand(x.shape[0], 1, device=x.device)
                p_txt = p_img = mask_prob / 2

        kernel_ok = (not self._generic_qxt and x.dim() == 2 and x.dtype == torch.int64 and move_chance.numel() == x.shape[0] and not (whole and (interleaved or not multimodal)))
        if kernel_ok:
            # comparisons and selects on [B, L] / [B, 1] as ONE launch (udm_qxt_absorbing): per-token move, the lottery with its both-sides-drawn rule,
            # the row REPLACEMENT by the modality mask, x_t
            xt, move, smt, smi, ignore = K.qxt_absorbing(x.contiguous(), r_move.float().contiguous(), move_chance.float(), self.mask_index, r_txt=r_txt, r_img=r_img,
                                                         p_txt=p_txt, p_img=p_img, modality_mask=batch["modality_mask"].contiguous() if whole else None)
            if allow_move_mask is not None:    # positions the caller protects never move (applied after the lottery)
                move = move & allow_move_mask
                xt = torch.where(move, self.mask_index, x)
            return (xt, ignore, None, smt, smi, move) if return_ignore_batch_mask_for_metrics else xt

        move = r_move < move_chance
        smt = smi = ignore = None
        if whole:
            smt = r_txt < p_txt
            smi = (r_img < p_img) if r_img is not None else torch.zeros_like(smt)
            if interleaved and multimodal:
                block_move, ignore = self._interleaved_block_lottery(batch, mask_prob, x.shape, x.device)
                move = move | block_move
                if allow_move_mask is not None:    # protected positions never move, whole-block masks included (model.py:564-565 sits behind every branch)
                    move = move & allow_move_mask
                xt = torch.where(move, self.mask_index, x)
                return (xt, ignore, None, smt, smi, move) if return_ignore_batch_mask_for_metrics else xt
            # a sample that drew BOTH sides keeps its per-token mask
            both = smt & smi
            smt, smi = smt & ~both, smi & ~both
            if multimodal:    # the row's mask is REPLACED: "all text masked, all image clean" (or the reverse)
                txt_cols, img_cols = batch["modality_mask"][..., 0], batch["modality_mask"][..., 1]
                move = torch.where(smt, txt_cols, torch.where(smi, img_cols, move))
            else:             # static layout: the side is masked ON TOP of the per-token mask; a text-only sample has no image side to mask
                smi = smi & ~batch["txt_sl"].all(dim=-1, keepdim=True)
                cols = torch.arange(x.shape[1], device=x.device)
                txt_cols = torch.zeros(x.shape[1], dtype=torch.bool, device=x.device)
                img_cols = torch.zeros_like(txt_cols)
                txt_cols[cols[self.static_txt_sl]] = True
                img_cols[cols[self.static_img_sl]] = True
                move = move | (smt & txt_cols) | (smi & img_cols)
            ignore = smt | smi
        if allow_move_mask is not None:
            move = move & allow_move_mask
        xt = torch.where(move, self.mask_index, x)
        return (xt, ignore, None, smt, smi, move) if return_ignore_batch_mask_for_metrics else xt

    def _interleaved_block_lottery(self, batch, mask_prob, shape, dev):
        """Packed / interleaved rows (model.py:483-522): every (modality, packed sample) block of more than 4 tokens is masked as a whole with probability
        2 p (k + 1) / n  (k: index of the block inside its sample, n: blocks of that sample); one uniform per block, drawn after the two per-row draws
        (the reference's call order).  Everything stays on the device; the ONE host read is the number of candidate blocks (it sizes the uniform draw - the
        reference reads every block boundary back).  Blocks are numbered in (row, position) order, the order of the reference's draws.
        Returns (block mask [B, L], rows that had a block masked [B])."""
        batch_size, seq_len = shape
        mod, sid = batch["modality"], batch["sample_ids"]
        x_is_cuda = mod.is_cuda
        if x_is_cuda and os.environ.get("UDM_INTERLEAVED_KERNELS", "1") != "0":
            # two launches (udm_interleaved_block_lottery, tokens.hip) instead of ~40 tensor statements with a sort and three contended scatter_adds; bit-identical to the
            # statements below (tests/test_gpu_kernels.py).  Device generator: one uniform per POSSIBLE candidate (a candidate block is longer than 4 tokens: at most
            # seq_len // 5 per row), no host read.  Replay (rng_device = "cpu"): the reference draws exactly n_cand uniforms, so the count is read back first.
            if self.rng_device is None:
                r = self._rand(batch_size * (seq_len // 5 + 1), 1, device=dev)
            else:
                _, _, n_dev = K.interleaved_block_lottery(mod, sid, torch.empty(0, device=dev), mask_prob)
                n_cand = int(n_dev)
                if not n_cand:
                    return torch.zeros((batch_size, seq_len), dtype=torch.bool, device=dev), torch.zeros((batch_size,), device=dev, dtype=torch.bool)
                r = self._rand(n_cand, 1, device=dev)
            accum, rows_hit, _ = K.interleaved_block_lottery(mod, sid, r, mask_prob)
            return accum, rows_hit
        N = batch_size * seq_len
        chg = torch.ones((batch_size, seq_len), dtype=torch.bool, device=dev)
        chg[:, 1:] = (mod[:, 1:] != mod[:, :-1]) | (sid[:, 1:] != sid[:, :-1])
        gid = chg.reshape(-1).cumsum(0) - 1                                          # block of every position
        one = torch.ones(N, dtype=torch.int64, device=dev)
        ar = torch.arange(N, device=dev)
        blen = torch.zeros(N, dtype=torch.int64, device=dev).scatter_add_(0, gid, one)
        bsid = torch.full((N,), -1, dtype=torch.int64, device=dev).scatter_(0, gid, sid.reshape(-1))
        brow = torch.zeros(N, dtype=torch.int64, device=dev).scatter_(0, gid, ar // seq_len)
        cand = (bsid >= 0) & (blen > 4)
        if self.rng_device is None and x_is_cuda:
            # device generator (production): one uniform per POSSIBLE candidate block - at most seq_len // 5 per row, blocks being longer than 4 - indexed by the
            # candidate's rank, so the host never reads the count back (that read stalled the launch queue at the top of every step: 2.7 ms of idle GPU per 81 ms
            # step of the packed 4608-token workload).  The draws of a block do not depend on how many follow it; the stream position after the step differs from a
            # draw of exactly n_cand values, which only a bit-exact REPLAY needs - and replays run with rng_device = "cpu" (below: the exact count).
            n_cand = batch_size * (seq_len // 5 + 1)
        else:
            n_cand = int(cand.sum())
            if not n_cand:
                return torch.zeros((batch_size, seq_len), dtype=torch.bool, device=dev), torch.zeros((batch_size,), device=dev, dtype=torch.bool)
        # k = index of the block among the candidate blocks of its (row, sample id), n = their number: stable sort by that key
        key = torch.where(cand, brow * (seq_len + 1) + bsid, torch.full_like(brow, (batch_size + 1) * (seq_len + 1)))
        order = torch.argsort(key, stable=True)
        skey = key[order]
        newg = torch.ones(N, dtype=torch.bool, device=dev)
        newg[1:] = skey[1:] != skey[:-1]
        g_first = torch.cummax(torch.where(newg, ar, torch.zeros_like(ar)), 0).values
        g_id = newg.cumsum(0) - 1
        g_size = torch.zeros(N, dtype=torch.int64, device=dev).scatter_add_(0, g_id, one)
        k = torch.empty_like(ar).scatter_(0, order, ar - g_first)
        n = torch.empty_like(ar).scatter_(0, order, g_size[g_id])
        r = self._rand(n_cand, 1, device=dev).reshape(-1)
        r_blk = r[(cand.cumsum(0) - 1).clamp(min=0)]
        thr = mask_prob * ((k + 1) / n) * 2                                      # fp32, in the reference's order of operations
        hit = cand & (r_blk < thr)
        rows_hit = torch.zeros(batch_size, dtype=torch.int64, device=dev).scatter_add_(0, brow, hit.long()) > 0
        return hit[gid].reshape(batch_size, seq_len), rows_hit

    def _sample_t(self, n, device):
        """Diffusion times of a batch (contract of model.py:589-619): t = eps_s + (1 - eps_s) u with u ~ U[0, 1), stratified over the batch when
        `antithetic_sampling` (sample i owns the i-th of n equal strata).  The GPU step gets the same values from `udm_sample_t_noise`."""
        tr = cfg_get(self.config, "trainer")
        if cfg_get(tr, "joint_ar_nar_timestep_warmup_steps", None) is not None:
            raise NotImplementedError("unidisc_amd: joint AR/NAR timestep warm-up is not on the denoising hot path")
        u = self._rand(n, device=device)
        if self.antithetic_sampling:
            u = torch.remainder(u / n + torch.arange(n, device=device) / n, 1)
        forced = cfg_get(tr, "force_timestep", None)
        if forced is not None:
            u = torch.full_like(u, forced)
        return (self.sampling_eps + (1 - self.sampling_eps) * u).to(torch.float32)

    def _restrict(self):
        return bool(cfg_get(cfg_get(self.config, "model"), "force_argmax_valid_indices", False))

    def _subs_parameterization(self, logits, xt, batch=None, modality=None, **kwargs):
        """model.py:621-658 on the HIP path: bf16 logits [B,L,V] -> SUBS log-probs [B,L,V] (same dtype)."""
        B, L, V = logits.shape
        flat = logits.reshape(B * L, V)
        if flat.stride(0) % 8 != 0 or flat.stride(1) != 1 or flat.data_ptr() % 16 != 0 or flat.dtype != torch.bfloat16:
            buf = torch.zeros((B * L, (V + 7) // 8 * 8), dtype=torch.bfloat16, device=logits.device)
            buf[:, :V] = flat
            flat = buf
        mod = modality
        if self._restrict() and mod is None and batch is not None and cfg_get(cfg_get(self.config, "trainer"), "multimodal_batches", False):
            mod = batch["modality"]
        if self._restrict() and mod is None:  # static slices (model.py:634-635)
            mod = torch.zeros((B, L), dtype=torch.int64, device=logits.device)
            mod[:, self.static_img_sl] = 1
        out = K.subs_logprobs(flat, xt.reshape(-1).contiguous() if xt is not None else None, mod.reshape(-1).contiguous() if mod is not None else None, V,
                              self.text_vocab_size, self.mask_index, self._restrict(), out_dtype=logits.dtype if logits.dtype == torch.float32 else torch.bfloat16)
        return out.view(B, L, V)

    def _process_sigma(self, sigma):  # model.py:660-672
        if sigma is None:
            assert cfg_get(cfg_get(self.config, "trainer"), "allow_null_sigma", False)
            return sigma
        if sigma.ndim > 1:
            sigma = sigma.squeeze(-1)
            assert sigma.ndim == 1, sigma.shape
        if not self.time_conditioning and cfg_get(cfg_get(self.config, "model"), "force_time_conditioning", False):
            sigma = torch.zeros_like(sigma)
        return sigma

    def forward(self, x, sigma, batch=None, forward_attention_mask=None, return_additional_loss=False, x_img_emb=None, disable_ar_shift=False,
                continuous_mode=False, joint_ar_nar_mask=None, return_logits=False, block_mask=None, update_cache_slice=None, **kwargs):
        """Returns log score (model.py:674-795): SUBS log-probs [B,L,V] (bf16), or raw logits when ``return_logits``."""
        sigma = self._process_sigma(sigma)
        logits = self.backbone(x, sigma, continuous_mode=continuous_mode, x_img_emb=x_img_emb, block_mask=block_mask,
                               update_cache_slice=update_cache_slice, **kwargs)
        if return_logits:
            return logits
        if logits.requires_grad:
            raise RuntimeError("unidisc_amd.Diffusion.forward: full SUBS log-probs are an inference product (no autograd); "
                               "training uses compute_loss (fused path) — wrap sampler calls in torch.no_grad()")
        return self._subs_parameterization(logits, xt=x, batch=batch, **kwargs)

    # ---- sampler inner loop: the `ddpm_cache` predictor (config.sampling.predictor default) with classifier-free guidance
    # (config.eval.cfg, one pass over [x ; x_uncond] or two with config.eval.split_cfg_batches); no attention caching
    def _sample_prior(self, *batch_dims):  # model_eval.py:1734-1735
        return self.mask_index * torch.ones(*batch_dims, dtype=torch.int64, device=self.device)

    def _row_modality(self, rows, B, L, modality):
        """Per-row modality for the SUBS vocabulary restriction (model.py:627-635), or None when it is off."""
        if not self._restrict():
            return None
        if modality is None:  # static slices
            modality = torch.zeros((B, L), dtype=torch.int64, device=rows.device)
            modality[:, self.static_img_sl] = 1
        return modality.reshape(-1).to(torch.int64).index_select(0, rows)

    def get_cfg_weight(self, t):  # model_eval.py:1737-1758
        ev = cfg_get(self.config, "eval", None)
        c = cfg_get(ev, "cfg", None)
        lo, hi = cfg_get(ev, "cfg_min_timestep", None), cfg_get(ev, "cfg_max_timestep", None)
        if not cfg_get(ev, "force_cfg_value", False):
            if c == -1:
                c = torch.linspace(0, 10, t.shape[0]).to(t.device)
            if lo is not None and hi is not None:
                w = (c * ((t - hi) / (lo - hi)))[:, None]
            else:
                w = (c * (1 - t))[:, None]
        else:
            w = c
        if lo is not None:
            w = torch.where(t > lo, w, torch.tensor(0.0, device=t.device))
        if hi is not None:
            w = torch.where(t < hi, w, torch.tensor(0.0, device=t.device))
        return w if isinstance(w, torch.Tensor) else torch.tensor(w)

    def _guided_masked_logits(self, x, t, sigma, x0_unmask, modality, sample_ids):
        """CFG branch of `_ddpm_forward` (model_eval.py:1763-1817) for the [MASK] rows of x: ONE backbone pass over [x ; x with the conditioning
        masked] (the reference's non-split form, :1787-1803), head on the same positions of both halves.  Returns the cache tuple
        (logits_cond [R, Vp], rows [R], n, logits_uncond [R, Vp], per-row weights [n]) or None when the guidance weight is zero everywhere."""
        ev = cfg_get(self.config, "eval", None)
        if cfg_get(ev, "cfg", None) is None or x0_unmask is None or not bool(x0_unmask.any()):
            return None
        w = self.get_cfg_weight(t)
        if not bool((w > 0).any()):
            return None
        B, L = x.shape
        x_uncond = x.clone()
        x_uncond[x0_unmask] = self.mask_index
        if cfg_get(ev, "split_cfg_batches", False):
            # model_eval.py:1770-1784: two backbone passes of batch B instead of one of 2 B (half the activation memory); the head runs on the
            # SAME [MASK] positions of both (plan_ids = x), so the two logits blocks line up row for row
            logits, rows, n = self.backbone.forward_masked_logits(x, sigma, modality=modality, sample_ids=sample_ids, plan_ids=x)
            logits_u, _, n_u = self.backbone.forward_masked_logits(x_uncond, sigma, modality=modality, sample_ids=sample_ids, plan_ids=x)
            assert n_u == n
        else:
            cat2 = (lambda v: None if v is None else torch.cat([v, v], 0))
            both, rows, n2 = self.backbone.forward_masked_logits(torch.cat([x, x_uncond], 0), cat2(sigma), modality=cat2(modality), sample_ids=cat2(sample_ids),
                                                                 plan_ids=torch.cat([x, x], 0))
            n = n2 // 2   # the stable partition lists the [MASK] rows of the first half, then the same positions of the second half
            logits, logits_u = both, both[n:]
        b_of = torch.div(rows[:n], L, rounding_mode="floor")
        w_b = w.to(torch.float32).reshape(-1)
        w_rows = (w_b.index_select(0, b_of) if w_b.numel() == B else w_b.expand(B).index_select(0, b_of)).contiguous()
        return logits[:n], rows[:n], n, logits_u[:n], w_rows

    @torch.no_grad()
    def _ddpm_caching_update(self, x, t, dt, p_x0=None, x0=None, x0_unmask=None, modality=None, sample_ids=None, u=None, seed=None, **kwargs):
        """model_eval.py:2073-2106 for the [MASK] rows only, fused (`udm_ddpm_sample_rows`): no [B, L, V] probabilities are built.
        `p_x0` is this implementation's cache: the masked rows' logits of the last forward (reused while x does not change, like the
        reference's p_x0 cache).  `u`: explicit uniforms [B, L, V] (parity runs); otherwise Philox keyed by `seed`.
        Returns (cache, x_next, nfe) like the reference."""
        if t.ndim > 1:
            t = t.squeeze(-1)
        B, L = x.shape
        nfe = 0
        if p_x0 is None:
            sigma_t, _ = self.noise(t)
            sig = self._process_sigma(sigma_t)
            block_mask = kwargs.get("block_mask")
            p_x0 = self._guided_masked_logits(x, t, sig, x0_unmask, modality, sample_ids) if block_mask is None else None
            if p_x0 is None:
                p_x0 = self.backbone.forward_masked_logits(x, sig, modality=modality, sample_ids=sample_ids, block_mask=block_mask)
            nfe = 1
        logits, rows, n = p_x0[:3]
        logits_u, w_rows = (p_x0[3], p_x0[4]) if len(p_x0) == 5 else (None, None)
        x_next = x.clone()
        if n > 0:
            rows_n = rows[:n]
            b_of = torch.div(rows_n, L, rounding_mode="floor")
            t_rows = t.to(torch.float32).index_select(0, b_of).contiguous()
            s_rows = (t.to(torch.float32) - dt).index_select(0, b_of).contiguous()
            u_rows = u.reshape(B * L, -1).index_select(0, rows_n).contiguous() if u is not None else None
            tok = K.ddpm_sample_rows(logits[:n], self.vocab_size, self.text_vocab_size, self.mask_index, t=t_rows, s=s_rows,
                                     modality=self._row_modality(rows_n, B, L, modality), restrict=self._restrict(), u=u_rows,
                                     seed=int(seed if seed is not None else torch.initial_seed()), logits_u=logits_u, w=w_rows)
            x_next.view(-1).index_copy_(0, rows_n, tok)
        return p_x0, x_next, nfe

    # ---- `eval.attention_caching` (model_eval.py:2296-2366): the logits cache of the [MASK] rows, moved between the full view and the text slice
    @staticmethod
    def _cache_to_text(cache, L, Lt):
        """(logits, rows, n) over [B, L] -> the rows inside positions < Lt, re-indexed for [B, Lt] (the reference slices p_x0[:, txt_sl])"""
        if cache is None:
            return None
        logits, rows, n = cache[:3]
        r = rows[:n]
        keep = (r % L) < Lt
        r2 = r[keep]
        return logits[:n][keep], torch.div(r2, L, rounding_mode="floor") * Lt + r2 % L, int(keep.sum())

    @staticmethod
    def _cache_to_full(saved, text_cache, L, Lt):
        """model_eval.py:2312-2323: the full cache saved when the state was sliced takes the text slice's current cache back (None when either is)"""
        if saved is None or text_cache is None:
            return None
        lg, rows, n = saved[:3]
        keep = (rows[:n] % L) >= Lt
        lt, rt, nt = text_cache[:3]
        rows_t = torch.div(rt[:nt], Lt, rounding_mode="floor") * L + rt[:nt] % Lt
        return torch.cat([lg[:n][keep], lt[:nt]], 0), torch.cat([rows[:n][keep], rows_t], 0), int(keep.sum()) + nt

    # ---- `maskgit` predictor (model_eval.py:2964-3001 schedule, :3046-3114 update)
    @staticmethod
    def adap_sche(x, step, mask_index, mode="arccos"):
        """Per-sample unmasking schedule [B, step]: how many tokens each step reveals (model_eval.py:2964-3001)."""
        num_masked = (x == mask_index).sum(dim=-1)
        r = torch.linspace(1, 0, step)
        if mode == "root":
            val = 1 - r ** 0.5
        elif mode == "linear":
            val = 1 - r
        elif mode == "square":
            val = 1 - r ** 2
        elif mode == "cosine":
            val = torch.cos(r * math.pi * 0.5)
        elif mode == "arccos":
            val = torch.arccos(r) / (math.pi * 0.5)
        else:
            return None
        val = val.to(x.device)
        out = []
        for n in num_masked:
            sche = ((val / val.sum()) * n).round()
            sche[sche == 0] = 1
            sche[-1] += n - sche.sum()
            sche[-1] = max(sche[-1], 0)
            out.append(sche.int())
        return torch.stack(out, dim=0)

    @torch.no_grad()
    def _nucleus_draw(self, logits, logits_u, w_rows, row_modality, top_p, temperature, seed):
        """Token per [MASK] row from the nucleus-filtered SUBS distribution (`nucleus_sampling_batch`, model_eval.py:2642-2685, quirks included: the
        temperature divides probabilities, the top id always stays).  Sort / cumsum / multinomial over [rows, V] are device tensor ops in row
        chunks (an evaluation-time sampler variant: no fused kernel)."""
        V, Vt = self.vocab_size, self.text_vocab_size
        out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
        gen = torch.Generator(device=logits.device).manual_seed(int(seed) + 2)
        ids = torch.arange(V, device=logits.device)
        for lo in range(0, logits.shape[0], 2048):
            z = logits[lo:lo + 2048, :V].float()
            if logits_u is not None:
                wv = w_rows[lo:lo + 2048, None]
                z = (1 + wv) * z - wv * logits_u[lo:lo + 2048, :V].float()
            bad = (ids == self.mask_index)[None].expand_as(z)
            if row_modality is not None:
                is_img = row_modality[lo:lo + 2048, None] == 1
                bad = bad | torch.where(is_img, ids[None] < Vt, ids[None] >= Vt)
            p = torch.softmax(z.masked_fill(bad, float("-inf")), dim=-1)
            sp, si = torch.sort(p / temperature, descending=True, dim=-1)
            keep = sp.cumsum(-1) <= top_p
            keep[:, 0] = True
            fp = sp * keep
            out[lo:lo + 2048] = si.gather(-1, torch.multinomial(fp / fp.sum(-1, keepdim=True), 1, generator=gen)).squeeze(-1)
        return out

    @torch.no_grad()
    def _maskgit_nucleus_update(self, x, t, dt, **kwargs):
        """`_maskgit_nucleus_update` (model_eval.py:3118-3167): the maskgit step with the token drawn from the nucleus-filtered distribution
        (config.eval.top_p / temperature); the confidence stays log p(token) under the UNfiltered distribution."""
        ev = cfg_get(self.config, "eval", None)
        return self._maskgit_update(x, t, dt, nucleus=(float(cfg_get(ev, "top_p", 0.95)), float(cfg_get(ev, "temperature", 0.9))), **kwargs)

    @torch.no_grad()
    def _maskgit_update(self, x, t, dt, schedule=None, step=None, x0=None, x0_unmask=None, modality=None, sample_ids=None, pred=None, gumbel=None, seed=None,
                        nucleus=None, **kwargs):
        """One `maskgit` step on the [MASK] rows only: token ~ p (or the replayed `pred`) and its log-probability from the fused row kernel
        (`udm_categorical_sample_rows`), confidence = log p + r_temp * gumbel * t, each sample keeps its num_unmask most confident predictions
        (threshold = k-th largest, as the reference).  `gumbel` [B, L]: explicit noise (replay); otherwise drawn on the device."""
        B, L = x.shape
        t_col = t if t.ndim > 1 else t[:, None]
        copy_flag = x != self.mask_index
        num_unmask = torch.minimum(schedule[:, step].to(torch.int64).to(x.device), (~copy_flag).sum(dim=-1))
        if bool(torch.all(num_unmask <= 0)):
            return x, 0
        r_temp = float(cfg_get(cfg_get(self.config, "eval", None), "maskgit_r_temp", 10))
        sigma_t, _ = self.noise(t_col.squeeze(-1))
        sig = self._process_sigma(sigma_t)
        cache = self._guided_masked_logits(x, t_col.squeeze(-1), sig, x0_unmask, modality, sample_ids)
        if cache is None:
            cache = self.backbone.forward_masked_logits(x, sig, modality=modality, sample_ids=sample_ids)
        logits, rows, n = cache[:3]
        logits_u, w_rows = (cache[3], cache[4]) if len(cache) == 5 else (None, None)
        rows_n = rows[:n]
        given = pred.reshape(-1).index_select(0, rows_n).contiguous() if pred is not None else None
        if given is None and nucleus is not None and n > 0:
            given = self._nucleus_draw(logits[:n], logits_u, w_rows, self._row_modality(rows_n, B, L, modality), nucleus[0], nucleus[1],
                                       int(seed if seed is not None else torch.initial_seed()))
        tok, logp = K.categorical_sample_rows(logits[:n], self.vocab_size, self.text_vocab_size, self.mask_index,
                                              modality=self._row_modality(rows_n, B, L, modality), restrict=self._restrict(), given=given,
                                              seed=int(seed if seed is not None else torch.initial_seed()), logits_u=logits_u, w=w_rows)
        if gumbel is None:
            g = torch.Generator(device=x.device).manual_seed(int(seed if seed is not None else torch.initial_seed()) + 1)
            u = torch.rand(B, L, device=x.device, generator=g).clamp_(1e-20, 1.0)
            gumbel = -torch.log(-torch.log(u))
        conf = torch.full((B * L,), float("-inf"), dtype=torch.float32, device=x.device)
        conf.index_copy_(0, rows_n, logp)
        conf = conf.view(B, L) + torch.where(copy_flag, torch.zeros((), device=x.device), r_temp * gumbel.to(torch.float32) * t_col.to(torch.float32))
        conf = torch.where(copy_flag, torch.full_like(conf, float("-inf")), conf)
        top, _ = torch.topk(conf, k=int(num_unmask.max()), dim=-1)
        thr = top.gather(-1, torch.clamp(num_unmask - 1, min=0)[:, None])
        thr = torch.where((num_unmask <= 0)[:, None], torch.full_like(thr, float("inf")), thr)
        pred_full = x.clone()
        pred_full.view(-1).index_copy_(0, rows_n, tok)
        return torch.where(conf >= thr, pred_full, x), 1

    @torch.no_grad()
    def _first_hitting_update(self, x, t, dt, schedule=None, step=None, x0=None, x0_unmask=None, modality=None, sample_ids=None, u=None, pos_u=None, seed=None,
                              **kwargs):
        """`_first_hitting_update` (model_eval.py:3005-3043) on the [MASK] rows: token = `_sample_categorical(p_x0)` through the fused row kernel (explicit
        uniforms `u` [B, L, V] replay the reference's draw, else Philox), then the num_unmask [MASK] positions with the largest lottery values
        `pos_u` [B, L] are revealed."""
        B, L = x.shape
        t_col = t if t.ndim > 1 else t[:, None]
        copy_flag = x != self.mask_index
        sigma_t, _ = self.noise(t_col.squeeze(-1))
        sig = self._process_sigma(sigma_t)
        cache = self._guided_masked_logits(x, t_col.squeeze(-1), sig, x0_unmask, modality, sample_ids)
        if cache is None:
            cache = self.backbone.forward_masked_logits(x, sig, modality=modality, sample_ids=sample_ids)
        num_unmask = torch.minimum(schedule[:, step].to(torch.int64).to(x.device), (~copy_flag).sum(dim=-1))
        if bool(torch.all(num_unmask <= 0)):
            return x, 1
        logits, rows, n = cache[:3]
        logits_u, w_rows = (cache[3], cache[4]) if len(cache) == 5 else (None, None)
        rows_n = rows[:n]
        u_rows = u.reshape(B * L, -1).index_select(0, rows_n).contiguous() if u is not None else None
        base = int(seed if seed is not None else torch.initial_seed())
        tok, _ = K.categorical_sample_rows(logits[:n], self.vocab_size, self.text_vocab_size, self.mask_index, modality=self._row_modality(rows_n, B, L, modality),
                                           restrict=self._restrict(), u=u_rows, seed=base, logits_u=logits_u, w=w_rows)
        if pos_u is None:
            pos_u = torch.rand(B, L, device=x.device, generator=torch.Generator(device=x.device).manual_seed(base + 1))
        rv = torch.where(~copy_flag, pos_u.to(torch.float32), torch.full((), -1.0, device=x.device))
        _, indices = torch.sort(rv, dim=-1, descending=True)
        final = torch.arange(L, device=x.device).expand(B, L) < num_unmask[:, None]
        result = torch.zeros_like(copy_flag)
        result.scatter_(-1, indices, final)
        pred_full = x.clone()
        pred_full.view(-1).index_copy_(0, rows_n, tok)
        return torch.where(result, pred_full, x), 1

    @torch.no_grad()
    def sample(self, num_steps=None, eps=1e-5, x0=None, x0_unmask=None, batch_size=None, modality=None, sample_ids=None, seed=None, noise=None,
               noise_removal=True, return_nfe=False, predictor=None, replay=None):
        """Token sampler: the `ddpm_cache` path of `_sample` (model_eval.py:2109-2455) without the decode / logging stack: prior = all [MASK]
        (x0 / x0_unmask conditioning kept fixed), timesteps = linspace(1, eps, steps + 1), one fused update per step with the logits
        cache reused while nothing changes, final arg-max of the log-probs (`noise_removal`).  Returns token ids [B, L]
        (and the number of backbone evaluations).  `noise`: optional list of uniforms [B, L, V] per step (replay of a recorded run)."""
        assert (x0 is None) == (x0_unmask is None)
        sampling = cfg_get(self.config, "sampling", None)
        if num_steps is None:
            num_steps = int(cfg_get(sampling, "steps", 1000)) if sampling is not None else 1000
        B = x0.shape[0] if x0 is not None else int(batch_size)
        L = x0.shape[1] if x0 is not None else int(cfg_get(cfg_get(self.config, "model"), "length"))
        x = self._sample_prior(B, L)
        if x0 is not None:
            x0 = x0.to(self.device).to(torch.int64)
            x0_unmask = x0_unmask.to(self.device).bool()
            x = torch.where(x0_unmask, x0, x)
        if modality is not None:
            modality = modality.to(self.device)
        timesteps = torch.linspace(1, eps, num_steps + 1, device=self.device)
        dt = (1 - eps) / num_steps
        base_seed = int(seed if seed is not None else torch.initial_seed())
        cache, nfe = None, 0
        if predictor is None:
            predictor = cfg_get(sampling, "predictor", "ddpm_cache") if sampling is not None else "ddpm_cache"
        if predictor not in ("ddpm_cache", "maskgit", "maskgit_nucleus", "first_hitting"):
            raise NotImplementedError(f"unidisc_amd.Diffusion.sample: predictor {predictor!r} is not built (ddpm_cache, maskgit, maskgit_nucleus, first_hitting)")
        schedule = None
        if predictor in ("maskgit", "maskgit_nucleus", "first_hitting"):   # model_eval.py:2274-2290
            schedule = self.adap_sche(x, num_steps, self.mask_index, "linear" if predictor == "first_hitting" else "arccos")
        # eval.attention_caching (model_eval.py:2296-2366): every `ratio` steps a full joint update, the step after it a full update in which image
        # queries see image keys only, every other step on the text slice alone (x, modality and the logits cache sliced to static_txt_sl)
        ev = cfg_get(self.config, "eval", None)
        caching = bool(cfg_get(ev, "attention_caching", False)) if ev is not None else False
        ratio = int(cfg_get(ev, "attention_caching_txt_to_img_ratio", 10)) if caching else 0
        self.sample_step_modes = []
        if caching:
            if predictor != "ddpm_cache" or x0 is not None or cfg_get(ev, "cfg", None) is not None or sample_ids is not None:
                raise NotImplementedError("unidisc_amd.Diffusion.sample: eval.attention_caching is built for the unconditional ddpm_cache predictor only")
            from .dit import ModalityMask
            Lt = int(cfg_get(cfg_get(self.config, "model"), "txt_length"))
            self.backbone.set_flex_attention_cache(B, L, self.device, None)
        sliced, saved = False, None
        for i in range(num_steps):
            t = timesteps[i] * torch.ones(B, 1, device=self.device)
            block_mask = None
            if caching:
                if i % ratio == 0:
                    if sliced:   # the saved full tensors take the text slice back
                        x_full, mod_full, cache_full = saved
                        x_full[:, :Lt] = x
                        cache = self._cache_to_full(cache_full, cache, L, Lt)
                        x, modality, sliced, saved = x_full, mod_full, False, None
                    self.sample_step_modes.append("full")
                elif (i - 1) % ratio == 0:
                    block_mask = ModalityMask(torch.zeros(B, dtype=torch.bool, device=self.device), torch.ones(B, dtype=torch.bool, device=self.device), Lt)
                    self.sample_step_modes.append("build")
                else:
                    if not sliced:
                        saved = (x.clone(), modality, cache)
                        cache = self._cache_to_text(cache, L, Lt)
                        x, modality, sliced = x[:, :Lt].contiguous(), (None if modality is None else modality[:, :Lt].contiguous()), True
                    self.sample_step_modes.append("text")
            if predictor in ("maskgit", "maskgit_nucleus"):   # replay: list of (pred [B, L] or None, gumbel [B, L] or None) per step
                pr, gm = replay[i] if replay is not None else (None, None)
                upd = self._maskgit_update if predictor == "maskgit" else self._maskgit_nucleus_update
                x, n = upd(x, t, dt, schedule=schedule, step=i, x0=x0, x0_unmask=x0_unmask, modality=modality, sample_ids=sample_ids,
                           pred=pr, gumbel=gm, seed=base_seed + 7919 * i)
                nfe += n
                if x0 is not None:
                    x = torch.where(x0_unmask, x0, x)
                continue
            if predictor == "first_hitting":   # replay: list of (u [B, L, V] or None, pos_u [B, L] or None) per step
                uu, pu = replay[i] if replay is not None else (None, None)
                x, n = self._first_hitting_update(x, t, dt, schedule=schedule, step=i, x0=x0, x0_unmask=x0_unmask, modality=modality, sample_ids=sample_ids,
                                                  u=uu, pos_u=pu, seed=base_seed + 7919 * i)
                nfe += n
                continue
            cache, x_next, n = self._ddpm_caching_update(x, t, dt, p_x0=cache, x0=x0, x0_unmask=x0_unmask, modality=modality, sample_ids=sample_ids,
                                                         u=noise[i] if noise is not None else None, seed=base_seed + 7919 * i, block_mask=block_mask)
            nfe += n
            if self.time_conditioning or not torch.equal(x_next, x):
                cache = None  # the reference's `if not allclose(x_next, x) or time_conditioning: p_x0_cache = None`
            x = x_next
            if x0 is not None:
                x = torch.where(x0_unmask, x0, x)
        if sliced:   # model_eval.py:2425-2440
            x_full, mod_full, _ = saved
            x_full[:, :Lt] = x
            x, modality = x_full, mod_full
        if caching:
            self.backbone.reset_kv_cache()
        if noise_removal:  # x = forward(x, sigma(t_last)).argmax(-1): unmasked positions keep their token, masked ones take the best valid id
            t = timesteps[-1] * torch.ones(B, device=self.device)
            sigma_t, _ = self.noise(t)
            logits, rows, n = self.backbone.forward_masked_logits(x, self._process_sigma(sigma_t), modality=modality, sample_ids=sample_ids)
            nfe += 1
            if n > 0:
                tok = K.ddpm_sample_rows(logits[:n], self.vocab_size, self.text_vocab_size, self.mask_index,
                                         modality=self._row_modality(rows[:n], B, L, modality), restrict=self._restrict(), greedy=True)
                x = x.clone()
                x.view(-1).index_copy_(0, rows[:n], tok)
            if x0 is not None:
                x = torch.where(x0_unmask, x0, x)
        return (x, nfe) if return_nfe else x

    # ---- model.py:797-1173, SUBS / continuous-time branch
    def compute_loss(self, batch, prefix, batch_idx=-1):
        cfg, tr = self.config, cfg_get(self.config, "trainer")
        kwargs = self.get_cond_dict(batch)
        modality_mask = batch.get("modality_mask", None)
        x0, attention_mask = batch["input_ids"], batch.get("attention_mask", None)
        if x0.shape[1] > cfg_get(cfg_get(cfg, "model"), "length"):
            raise NotImplementedError("unidisc_amd: sequence sub-sampling (text8-crop) is not on the denoising hot path")
        if (isinstance(self.noise, LogLinearNoise) and cfg_get(tr, "joint_ar_nar_timestep_warmup_steps", None) is None
                and cfg_get(tr, "force_timestep", None) is None):
            # `_sample_t` + the schedule + the move chance: seventeen statements on [B] tensors as one launch, bit-identical (tests/test_gpu_kernels.py); the draw stays here
            t, sigma, dsigma, mc = K.sample_t_noise(self._rand(x0.shape[0], device=x0.device).float(), antithetic=bool(self.antithetic_sampling),
                                                    sampling_eps=self.sampling_eps, noise_eps=self.noise.eps)
            move_chance = mc[:, None]
        else:
            t = self._sample_t(x0.shape[0], x0.device)
            sigma, dsigma = self.noise(t)
            move_chance = 1 - torch.exp(-sigma[:, None])
        unet_conditioning = sigma[:, None]
        xt, ignore_batch_mask_for_metrics, joint_ar_nar_mask, should_mask_txt, should_mask_img, move_indices = self.q_xt(
            x0, move_chance, return_ignore_batch_mask_for_metrics=True, batch=batch)
        m = cfg_get(cfg, "model")
        p_txt, p_img = cfg_get(m, "flex_attention_txt_masking_prob", None), cfg_get(m, "flex_attention_img_masking_prob", None)
        if (p_img is not None or p_txt is not None) and self.backbone.training:
            # modality attention dropout (model.py:863-878): per sample, text queries are restricted to text keys and / or image queries to image keys
            if p_img is None or p_txt is None:
                raise ValueError("unidisc_amd: set both model.flex_attention_txt_masking_prob and flex_attention_img_masking_prob (the reference compares both)")
            assert xt.shape[1] == cfg_get(m, "img_length") + cfg_get(m, "txt_length")
            txt_drop = self._rand(xt.shape[0], device=xt.device) < p_txt
            img_drop = self._rand(xt.shape[0], device=xt.device) < p_img
            if should_mask_txt is not None:   # a modality that is masked out entirely must not be left seeing only itself
                txt_drop = txt_drop & ~should_mask_txt.squeeze(-1)
                img_drop = img_drop & ~should_mask_img.squeeze(-1)
            kwargs["block_mask"] = ModalityMask(txt_drop, img_drop, cfg_get(m, "txt_length"))
            drop_any = (txt_drop | img_drop).unsqueeze(-1)
            ignore_batch_mask_for_metrics = drop_any if ignore_batch_mask_for_metrics is None else (ignore_batch_mask_for_metrics | drop_any)
        if cfg_get(tr, "interleaved_training_flex_attention", False):
            kwargs["sample_ids"] = batch["sample_ids"]  # the document mask is derived from sample_ids inside the attention kernel

        # fused backbone + SUBS + gather: log p_theta(x0 | xt) per token, fp32 (model.py:908-925, :967)
        log_p_theta = self.backbone.forward_logp(xt, x0, self._process_sigma(unet_conditioning), modality=kwargs.get("modality"),
                                                 sample_ids=kwargs.get("sample_ids"), restrict_modality=self._restrict(),
                                                 block_mask=kwargs.get("block_mask"), attention_mask=kwargs.get("attention_mask"))
        self._flush_checks()   # (the forward has already waited for this step's [MASK]-row count: the queued batch checks are complete, no extra wait)
        self._last = dict(t=t, sigma=sigma, dsigma=dsigma, xt=xt, move_indices=move_indices, log_p_theta=log_p_theta, modality=kwargs.get("modality"))

        # The loss arithmetic of model.py:1010-1160 (schedule weights, masked mean or the modality-weighted text / image sum with the optional text-loss cap,
        # the reported per-token NLLs and fractions) runs as ONE launch - K.diffusion_loss - instead of ~75 small tensor statements between the forward and
        # the backward; the differentiable loss is that launch's value with d loss / d log_p = its coefficient tensor (the loss is linear in log_p).
        if cfg_get(tr, "no_ce_weighting", False):
            w_std = torch.ones_like(sigma)
            w_loss = w_std
        else:
            w_std = dsigma / torch.expm1(sigma)
            gamma = cfg_get(tr, "softmin_snr", None)
            w_loss = dsigma / (torch.expm1(sigma) + (1 / gamma)) if gamma is not None else w_std
        weighted = cfg_get(tr, "text_loss_weight", None) is not None and cfg_get(tr, "img_loss_weight", None) is not None
        use_mm = (cfg_get(tr, "multimodal_batches", False) or weighted) and modality_mask is not None
        if weighted and modality_mask is None:
            raise ValueError("unidisc_amd: trainer.text_loss_weight / img_loss_weight need batches with a modality map")
        std_nlls, coef, sc = K.diffusion_loss(
            log_p_theta.detach().float(), w_loss.float(), w_std.float(), attention_mask, modality_mask if use_mm else None, weighted=weighted,
            full_mask=bool(cfg_get(tr, "force_full_attention_mask_loss_only", False)),
            text_w=float(cfg_get(tr, "text_loss_weight", None)) if weighted else 1.0,   # an explicit 0.0 stays 0.0 (model.py:1041-1044 multiplies by the configured value)
            img_w=float(cfg_get(tr, "img_loss_weight", None)) if weighted else 1.0, ratio=cfg_get(tr, "set_max_txt_loss_ratio", None) if weighted else None)
        loss = _LinearLoss.apply(log_p_theta, coef, sc[0])
        loss_dict = dict(loss=loss, extra_losses=dict())
        if cfg_get(tr, "log_seperate_modal_losses", False):
            loss_dict.update(dict(std_txt_loss=std_nlls * modality_mask[..., 0], std_img_loss=std_nlls * modality_mask[..., 1]))
        if cfg_get(tr, "mask_entire_modality", None) is not None and self.backbone.training:
            loss_dict["batch_ignore_loss"] = ignore_batch_mask_for_metrics.reshape(-1)  # (reference: .squeeze(-1), which breaks its own interleaved branch at B = 1)
        if use_mm:
            loss_dict["extra_losses"]["trainer/img_frac"] = sc[4]
            loss_dict["extra_losses"]["trainer/txt_frac"] = sc[3]
            loss_dict["extra_losses"]["trainer/attention_mask_valid_frac"] = sc[5]
            if "batch_ignore_loss" in loss_dict:
                loss_dict["extra_losses"]["trainer/ignore_batch_metrics_frac"] = loss_dict["batch_ignore_loss"].sum() / loss_dict["batch_ignore_loss"].numel()
        if weighted:
            loss_dict.update(dict(txt_loss=sc[1], img_loss=sc[2]))
        if "batch_ignore_loss" in loss_dict:
            attention_mask = torch.where(loss_dict["batch_ignore_loss"][:, None].repeat(1, attention_mask.shape[-1]),
                                         torch.full_like(attention_mask, False), attention_mask)
        losses = Loss(loss=loss_dict["loss"], img_loss=loss_dict.get("img_loss", 0), txt_loss=loss_dict.get("txt_loss", 0), nlls=std_nlls,
                      txt_nlls=loss_dict.get("std_txt_loss", 0), img_nlls=loss_dict.get("std_img_loss", 0), token_mask=attention_mask,
                      modality_mask=modality_mask, extra_losses=loss_dict.get("extra_losses", None))
        if cfg_get(tr, "disable_torchmetrics", False):
            raise NotImplementedError("Torchmetrics disabled")
        if prefix == "train":
            return losses
        # model.py:1163-1171: validation / test update the metric collections the trainer attached (any object with .update(values, mask))
        if prefix == "val":
            if getattr(self, "valid_metrics", None) is None:
                raise RuntimeError("unidisc_amd.compute_loss(prefix='val'): attach `valid_metrics` (and optionally `valid_txt_metrics` / `valid_img_metrics`) first")
            self.valid_metrics.update(losses.nlls, losses.token_mask)
            if getattr(self, "valid_txt_metrics", None) is not None:
                self.valid_txt_metrics.update(losses.txt_nlls, losses.modality_mask[..., 0] & losses.token_mask)
                self.valid_img_metrics.update(losses.img_nlls, losses.modality_mask[..., 1] & losses.token_mask)
            return None
        if prefix == "test":
            if getattr(self, "test_metrics", None) is None:
                raise RuntimeError("unidisc_amd.compute_loss(prefix='test'): attach `test_metrics` first")
            self.test_metrics.update(losses.nlls, losses.token_mask)
            return None
        raise ValueError(f"Invalid prefix: {prefix}")
