"""Import shim for the read-only reference checkout (TEST INFRASTRUCTURE ONLY).

Used only by ``oracle/make_golden.py`` in the build container, where
``/root/reference`` exists.  Nothing in the product package, ``bench.py`` or the
``-m gpu`` tests imports this file: the reference never travels to the GPU box.

The reference imports a number of third-party packages that are absent from
this image (omegaconf, hydra, diffusers, tensordict, torchmetrics, wandb, ...).
None of them is on the denoising hot path; we satisfy the imports with
attribute-stub modules and provide tiny real
classes where the reference subclasses something at import time
(``model_utils.py:123`` MeanMetric, ``utils.py:59`` CosineLRScheduler).

The only stubbed symbol whose *arithmetic* can reach the hot path is
``diffusers.models.embeddings.get_2d_rotary_pos_embed_lumina`` (``models/dit.py:12,1052``);
callers may install a restatement through :func:`install_lumina_rope` — results that
depend on it are marked "parity unpinned" (diffusers 0.32.2 is not vendored).
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = "/root/reference"

MISSING = [
    "omegaconf", "hydra", "diffusers", "tensordict", "torchmetrics", "wandb", "image_utils", "ipdb", "timm",
    "torchvision", "lightning", "webdataset", "cv2", "torchinfo", "jaxtyping", "torchtnt", "mup", "peft",
    "submitit", "rich", "torch_fidelity", "cleanfid", "mauve", "open_clip", "hpsv2", "ImageReward",
    "pycocoevalcap", "deepspeed", "pynvml", "flash_attn", "flash_attn_interface", "fvcore", "viztracer",
    "lovely_tensors", "PIL", "matplotlib", "torch_xla", "ml_dtypes", "lpips", "T2IBenchmark", "clip",
    "fairscale", "xformers", "bitsandbytes", "gradio", "streamlit", "fasthtml", "evaluate", "nltk",
]


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        m = MagicMock(name=f"{self.__name__}.{k}")
        setattr(self, k, m)
        return m


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        root = name.split(".")[0]
        if root in MISSING:
            try:  # prefer a real install if the image has one
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(name, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return None
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


_installed = False


def install():
    """Make ``import model`` / ``import models.dit`` work from /root/reference."""
    global _installed
    if _installed:
        return
    import torch

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.meta_path.insert(0, _Finder())
    agg = importlib.import_module("torchmetrics.aggregation")
    agg.MeanMetric = type("MeanMetric", (torch.nn.Module,), {"__init__": lambda self, **kw: torch.nn.Module.__init__(self)})
    importlib.import_module("tensordict").TensorDict = type("TensorDict", (dict,), {})
    importlib.import_module("timm.scheduler").CosineLRScheduler = type("CosineLRScheduler", (), {})
    oc = importlib.import_module("omegaconf")

    class _OC:
        @staticmethod
        def create(x):
            return x

        @staticmethod
        def register_new_resolver(*a, **k):
            return None

    oc.OmegaConf = _OC
    _installed = True


def install_lumina_rope(fn):
    """Install ``fn(dim, h, w, linear_factor, ntk_factor) -> complex [h,w,dim/2]`` as the diffusers symbol."""
    install()
    emb = importlib.import_module("diffusers.models.embeddings")
    emb.get_2d_rotary_pos_embed_lumina = fn
    if "models.dit" in sys.modules:
        sys.modules["models.dit"].get_2d_rotary_pos_embed_lumina = fn


class Cfg:
    """Attribute bag whose missing keys raise AttributeError (the reference leans on getattr(cfg, k, default))."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __getattr__(self, k):
        raise AttributeError(k)

    def __contains__(self, k):
        return k in self.__dict__

    def get(self, k, d=None):
        return self.__dict__.get(k, d)
