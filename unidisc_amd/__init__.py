"""unidisc_amd — MI355X-native (gfx950) implementation of UniDisc's discrete-diffusion denoising hot path.

Drop-in boundary: :class:`unidisc_amd.dit.DIT` for ``models/dit.py::DIT`` and
:class:`unidisc_amd.diffusion.Diffusion` for the hot-path methods of ``model.py::Diffusion``.
Compute runs in hand-written HIP kernels behind the C ABI in ``include/unidisc_hip.h``; there is no CPU fallback.
"""
from .config import Cfg, make_config, MODEL_PRESETS  # noqa: F401
from .dit import DIT, ModalityMask  # noqa: F401
from .diffusion import Diffusion, Loss  # noqa: F401
from .noise_schedule import LogLinearNoise  # noqa: F401
from .optim import FusedAdamW  # noqa: F401
from .zero import ShardedAdamW, ShardedGradSync, wrap_sharded  # noqa: F401
from .checkpoint import load_backbone_checkpoint, save_backbone_checkpoint, read_state_dict  # noqa: F401
from .token_data import TokenShard, TokenBatcher, WeightedDatasetSampler, PackingCollate  # noqa: F401

__all__ = ["DIT", "ModalityMask", "Diffusion", "Loss", "LogLinearNoise", "Cfg", "make_config", "MODEL_PRESETS", "load_backbone_checkpoint",
           "save_backbone_checkpoint", "read_state_dict", "FusedAdamW", "ShardedAdamW", "ShardedGradSync", "wrap_sharded", "TokenShard", "TokenBatcher", "WeightedDatasetSampler", "PackingCollate"]
