"""Fused AdamW + global-norm clipping for the engine's parameters.

Mirrors what the reference's training loop does between backward and the next forward (model.py:1516-1545):
``accelerator.clip_grad_norm_(backbone.parameters(), trainer.gradient_clip_val)`` then ``torch.optim.AdamW(fused=True).step()``
(model_setup.py:385-424; config.optim: lr 3e-4, betas (0.9, 0.999), eps 1e-8, weight_decay 0) — in hand-written HIP kernels
(csrc/optim.hip).  For the GEMM weights the update also refreshes the bf16 shadows W / Wᵀ the forward and dgrad GEMMs read, so the
per-forward weight cast (autocast in the reference, ``cast_transpose`` here: ≈3 % of a 1.4 B step) leaves the training step.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import kernels as K


class FusedAdamW:
    """``opt = FusedAdamW(backbone, lr=..., max_grad_norm=1.0); loss.backward(); opt.step(); opt.zero_grad()``.

    ``lr`` may be changed between steps (``opt.lr = scheduler(...)``), like ``param_group["lr"]``.  ``grad_norm`` holds the pre-clip global
    norm of the last step as a device tensor (no host synchronisation anywhere in ``step``)."""

    def __init__(self, backbone: torch.nn.Module, lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: Optional[float] = None, maintain_shadows: bool = True, params: Optional[Iterable[torch.nn.Parameter]] = None,
                 ema_decay: Optional[float] = None, ema_use_num_updates: bool = True):
        self.backbone = backbone
        self.params = [p for p in (params if params is not None else backbone.parameters()) if p.requires_grad]
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("FusedAdamW: parameters must be contiguous fp32 masters")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm is not None else None
        self.maintain_shadows = bool(maintain_shadows) and hasattr(backbone, "refresh_weight_shadows")
        self.step_count = 0
        self.state = {id(p): (torch.zeros_like(p), torch.zeros_like(p)) for p in self.params}
        self._gsq = None
        self.grad_norm = None
        # parameter EMA (reference models/ema.py `ExponentialMovingAverage`, stepped right after optimizer.step at model.py:1541-1545; config
        # trainer.ema, 0 = off): the shadow starts as a copy of the parameters and is updated inside the optimizer kernels
        self.ema_decay = float(ema_decay) if ema_decay else None
        self.ema_num_updates = 0 if ema_use_num_updates else None
        self.ema = {id(p): p.detach().clone() for p in self.params} if self.ema_decay else None
        self._ema_backup = None

    def _ema_decay_now(self):   # models/ema.py:46-49
        d = self.ema_decay
        if self.ema_num_updates is not None:
            self.ema_num_updates += 1
            d = min(d, (1 + self.ema_num_updates) / (10 + self.ema_num_updates))
        return d

    @torch.no_grad()
    def ema_store_and_copy(self):
        """`ema.store(params); ema.copy_to(params)` (model.py evaluation with EMA weights): parameters <- EMA, originals kept for `ema_restore`."""
        if self.ema is None:
            raise RuntimeError("FusedAdamW: no EMA (ema_decay not set)")
        self._ema_backup = [p.detach().clone() for p in self.params]
        for p in self.params:
            p.copy_(self.ema[id(p)])
        if hasattr(self.backbone, "refresh_weight_shadows"):
            self.backbone.refresh_weight_shadows(force=True)

    @torch.no_grad()
    def ema_restore(self):
        if self._ema_backup is None:
            raise RuntimeError("FusedAdamW: ema_restore without ema_store_and_copy")
        for p, b in zip(self.params, self._ema_backup):
            p.copy_(b)
        self._ema_backup = None
        if hasattr(self.backbone, "refresh_weight_shadows"):
            self.backbone.refresh_weight_shadows(force=True)

    # ------------------------------------------------------------------------------------------------
    def _grad_sumsq(self, grads):
        """Device scalar Σ g² over all gradients: one pass over the engine's flat gradient buffer when every p.grad is a view of it."""
        dev = grads[0].device
        if self._gsq is None or self._gsq.device != dev:
            self._gsq = torch.zeros(1, dtype=torch.float32, device=dev)
        # The engine's backward returns views of ONE flat buffer.  Autograd installs DETACHED copies of the views as p.grad (no `_base`) and the buffer's own Python
        # wrapper is gone by now (round 4: a weak reference to it was dead at every step, so 341 sum-of-squares launches ran instead of one) - but the gradients
        # still share its STORAGE.  When every gradient lives in one storage of the expected size, that storage is this backward's buffer (alignment gaps zeroed
        # by `_alloc_grads`): one pass over all of it.
        st = grads[0].untyped_storage()
        covered = getattr(self.backbone, "_last_grad_numel", -1)
        total = sum(g.numel() for g in grads)
        if (total == covered and st.nbytes() % 4 == 0 and st.nbytes() // 4 <= total + 64 * len(grads)
                and all(g.dtype == torch.float32 and g.untyped_storage().data_ptr() == st.data_ptr() for g in grads)):
            flat = torch.empty(0, dtype=torch.float32, device=dev).set_(st, 0, (st.nbytes() // 4,))
            K.sumsq(flat, self._gsq)
            return self._gsq
        self._gsq.zero_()
        for g in grads:  # generic path (accumulated / foreign gradients): one pass per tensor
            part = torch.zeros(1, dtype=torch.float32, device=dev)
            K.sumsq(g.reshape(-1), part)
            self._gsq += part
        return self._gsq

    @torch.no_grad()
    def step(self):
        todo = [p for p in self.params if p.grad is not None]
        if not todo:
            return
        for p in todo:
            if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                raise ValueError("FusedAdamW: gradients must be contiguous fp32")
        self.step_count += 1
        gsq = None
        if self.max_grad_norm is not None:
            gsq = self._grad_sumsq([p.grad for p in todo])
            self.grad_norm = gsq.sqrt()
        lins = {}
        if self.maintain_shadows:
            if getattr(self.backbone, "_lins", None) is None:
                self.backbone.refresh_weight_shadows(force=True)
            lins = {id(l.weight): l for l in self.backbone._lins.values()}
        b1, b2 = self.betas
        ed = self._ema_decay_now() if self.ema is not None else 0.0
        flat_items, shadow_items = [], []
        for p in todo:
            m, v = self.state[id(p)]
            lin = lins.get(id(p))
            e = self.ema[id(p)] if self.ema is not None else None
            if lin is not None and lin.w16 is not None:
                if p.is_cuda and p.dim() == 2 and p.is_contiguous() and all(t.data_ptr() % 16 == 0 for t in (p, p.grad, m, v) + ((e,) if e is not None else ())):
                    shadow_items.append((p, p.grad, m, v, e, lin.w16, lin.w16t))   # every GEMM weight with its bf16 shadows: ONE launch below
                else:
                    K.adamw_step_shadow(p, p.grad, m, v, self.lr, b1, b2, self.eps, self.weight_decay, self.step_count, gsq, self.max_grad_norm, lin.w16, lin.w16t,
                                        ema=e, ema_decay=ed)
            elif p.is_cuda and p.is_contiguous() and all(t.data_ptr() % 16 == 0 for t in (p, p.grad, m, v) + ((e,) if e is not None else ())):
                flat_items.append((p, p.grad, m, v, e))     # every norm / bias / embedding tensor: ONE launch below (a 1.4 B DiT has ~250 of them)
            else:
                K.adamw_step(p, p.grad, m, v, self.lr, b1, b2, self.eps, self.weight_decay, self.step_count, gsq, self.max_grad_norm, ema=e, ema_decay=ed)
        if shadow_items:
            key = tuple(t.data_ptr() if t is not None else 0 for it in shadow_items for t in it)
            if getattr(self, "_smulti_key", None) != key:
                self._smulti_jobs, self._smulti_key = K.adamw_shadow_jobs(shadow_items, shadow_items[0][0].device), key
            K.adamw_step_shadow_multi(self._smulti_jobs, self.lr, b1, b2, self.eps, self.weight_decay, self.step_count, gsq, self.max_grad_norm, ema_decay=ed)
        if flat_items:
            key = tuple(t.data_ptr() if t is not None else 0 for it in flat_items for t in it)
            if getattr(self, "_multi_key", None) != key:     # the table holds raw pointers: rebuilt whenever one moves (the flat gradient buffer is re-allocated per backward
                self._multi_jobs, self._multi_key = K.adamw_jobs(flat_items, flat_items[0][0].device), key   # only if the caching allocator hands out another block)
            K.adamw_step_multi(self._multi_jobs, self.lr, b1, b2, self.eps, self.weight_decay, self.step_count, gsq, self.max_grad_norm, ema_decay=ed)
        if self.maintain_shadows:
            # the shadows are current: the next forward must not re-cast (kernel writes do not bump tensor versions, so record them)
            self.backbone.recast_every_forward = False
            self.backbone._shadow_versions = [l.weight._version for l in self.backbone._lins.values()]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def state_dict(self):
        return dict(step=self.step_count, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, max_grad_norm=self.max_grad_norm,
                    exp_avg=[self.state[id(p)][0] for p in self.params], exp_avg_sq=[self.state[id(p)][1] for p in self.params],
                    ema_decay=self.ema_decay, ema_num_updates=self.ema_num_updates,
                    dropout_fwd_count=int(getattr(self.backbone, "_fwd_count", 0)),   # position of the engine's dropout stream (resume must not replay masks)
                    ema=[self.ema[id(p)] for p in self.params] if self.ema is not None else None)

    def load_state_dict(self, sd):
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        for p, m, v in zip(self.params, sd["exp_avg"], sd["exp_avg_sq"]):
            self.state[id(p)][0].copy_(m)
            self.state[id(p)][1].copy_(v)
        if self.ema is not None and sd.get("ema") is not None:
            self.ema_num_updates = sd.get("ema_num_updates", self.ema_num_updates)
            for p, e in zip(self.params, sd["ema"]):
                self.ema[id(p)].copy_(e)
        if "dropout_fwd_count" in sd and hasattr(self.backbone, "_fwd_count"):
            self.backbone._fwd_count = int(sd["dropout_fwd_count"])
        # a resume usually reloads the master weights as well (writes the version counters do not see): rebuild the bf16 shadows on the next forward
        if hasattr(self.backbone, "invalidate_shadows"):
            self.backbone.invalidate_shadows()
            self.backbone.recast_every_forward = True
