"""Checkpoint I/O compatible with the reference's model files.

The reference writes the backbone with ``accelerator.save_model(self.backbone, dir)`` / ``safetensors.torch.save_model`` →
``<dir>/model.safetensors`` (model_setup.py:914-923) and publishes the same layout on the Hub (``aswerdlow/unidisc_*``, README.md:24-25);
``accelerator.save_state`` puts the same file into its state directory (main.py:765-823).  Keys are ``DIT.state_dict()`` names, possibly
behind wrapper prefixes (``_orig_mod.`` from torch.compile, ``module.`` from DDP, ``backbone.`` when the whole Diffusion module was saved).
``unidisc_amd.DIT`` keeps those parameter names, so loading is a strict ``load_state_dict`` after prefix stripping; weights are kept as fp32
masters whatever dtype the file holds.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

_PREFIXES = ("_orig_mod.", "module.", "backbone.", "model.")
_FILES = ("model.safetensors", "pytorch_model.bin", "model.bin")


def _resolve(path: str) -> str:
    if os.path.isdir(path):
        for name in _FILES:
            f = os.path.join(path, name)
            if os.path.isfile(f):
                return f
        raise FileNotFoundError(f"unidisc_amd.checkpoint: none of {_FILES} under {path}")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"unidisc_amd.checkpoint: {path} does not exist")
    return path


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Tensors of a reference checkpoint file / directory with wrapper prefixes removed (no model needed)."""
    f = _resolve(path)
    if f.endswith(".safetensors"):
        from safetensors.torch import load_file

        sd = load_file(f, device="cpu")
    else:
        sd = torch.load(f, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
    out = {}
    for k, v in sd.items():
        stripped = True
        while stripped:
            stripped = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k, stripped = k[len(p):], True
        out[k] = v
    return out


def load_backbone_checkpoint(backbone: torch.nn.Module, path: str, strict: bool = True) -> Tuple[list, list]:
    """Load a reference backbone checkpoint into ``unidisc_amd.DIT`` (fp32 masters).  Returns (missing, unexpected) like load_state_dict;
    with ``strict`` (default) any mismatch of names or shapes raises.  The bf16 weight shadows are rebuilt on the next forward."""
    sd = read_state_dict(path)
    own = backbone.state_dict()
    cast = {k: (v.to(own[k].dtype) if k in own and v.is_floating_point() else v) for k, v in sd.items()}
    res = backbone.load_state_dict(cast, strict=strict)
    if hasattr(backbone, "invalidate_shadows"):
        backbone.invalidate_shadows()  # force a re-cast
    return list(res.missing_keys), list(res.unexpected_keys)


def save_backbone_checkpoint(backbone: torch.nn.Module, path: str) -> str:
    """Write ``<path>/model.safetensors`` in the layout the reference's loader (``accelerator.load_state`` / ``load_model``) expects."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    f = os.path.join(path, "model.safetensors")
    sd = {k: v.detach().to("cpu").contiguous() for k, v in backbone.state_dict().items()}
    save_file(sd, f, metadata={"format": "pt"})
    return f
