"""the reference's backbone checkpoints (model.safetensors in DIT.state_dict() names, possibly behind wrapper prefixes)
load into unidisc_amd.DIT, and what unidisc_amd saves has exactly the reference's names, shapes and values.  The reference-side
tensors are the golden fixtures' `params` (dumped from the imported reference by oracle/make_golden.py)."""
import os

import pytest
import torch

import fake_kernels
from golden_utils import CASE_NAMES, Golden, rel_err
from oracle import unidisc_oracle as O
from product_utils import build_product, product_config
from unidisc_amd import Diffusion, load_backbone_checkpoint, read_state_dict, save_backbone_checkpoint


def _reference_style_file(g, path, prefix="", dtype=None, name="model.safetensors"):
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    sd = {prefix + k: (v.to(dtype) if dtype is not None else v).contiguous() for k, v in g.params().items()}
    save_file(sd, os.path.join(path, name), metadata={"format": "pt"})
    return path


@pytest.mark.parametrize("name", CASE_NAMES)
@pytest.mark.parametrize("prefix", ["", "_orig_mod.", "module._orig_mod.", "backbone."])
def test_load_reference_checkpoint(tmp_path, name, prefix):
    g = Golden(name)
    ckpt = _reference_style_file(g, str(tmp_path / "ckpt"), prefix=prefix)
    diff = Diffusion(product_config(g.case), None, "cpu")  # fresh random init
    missing, unexpected = load_backbone_checkpoint(diff.backbone, ckpt)
    assert missing == [] and unexpected == []
    sd = diff.backbone.state_dict()
    for k, v in g.params().items():
        assert sd[k].dtype == torch.float32 and torch.equal(sd[k], v.float()), k


def test_bf16_checkpoint_becomes_fp32_masters_and_wrong_shapes_raise(tmp_path):
    g = Golden("b_small")
    ckpt = _reference_style_file(g, str(tmp_path / "bf16"), dtype=torch.bfloat16)
    diff = Diffusion(product_config(g.case), None, "cpu")
    load_backbone_checkpoint(diff.backbone, ckpt)
    for k, v in g.params().items():
        assert diff.backbone.state_dict()[k].dtype == torch.float32
        assert torch.equal(diff.backbone.state_dict()[k], v.to(torch.bfloat16).float())
    bad = dict(read_state_dict(ckpt))
    k0 = next(k for k, v in bad.items() if v.dim() == 2)
    bad[k0] = bad[k0][:-1]
    from safetensors.torch import save_file

    os.makedirs(tmp_path / "bad")
    save_file({k: v.contiguous() for k, v in bad.items()}, str(tmp_path / "bad" / "model.safetensors"))
    with pytest.raises(RuntimeError):
        load_backbone_checkpoint(diff.backbone, str(tmp_path / "bad"))
    with pytest.raises(FileNotFoundError):
        load_backbone_checkpoint(diff.backbone, str(tmp_path / "nothing_here"))


def test_save_round_trip_has_reference_schema_and_same_forward(tmp_path, monkeypatch):
    from unidisc_amd import dit as dit_mod, diffusion as diff_mod

    monkeypatch.setattr(dit_mod, "K", fake_kernels)
    monkeypatch.setattr(diff_mod, "K", fake_kernels)
    g = Golden("c_large")
    diff = build_product(g, device="cpu")
    f = save_backbone_checkpoint(diff.backbone, str(tmp_path / "out"))
    assert os.path.basename(f) == "model.safetensors"
    saved = read_state_dict(str(tmp_path / "out"))
    ref = g.params()
    assert set(saved) == set(ref)
    for k in ref:
        assert saved[k].shape == ref[k].shape and torch.equal(saved[k], ref[k].float()), k
    other = Diffusion(product_config(g.case), None, "cpu")
    load_backbone_checkpoint(other.backbone, f)  # a file path works as well as a directory
    other.backbone.train()
    xt = g.t("fp32/xt")
    with torch.no_grad():
        a = diff.backbone(xt, None, modality=g.t("fp32/modality"))
        b = other.backbone(xt, None, modality=g.t("fp32/modality"))
    assert torch.equal(a, b)
    truth = g.t("fp32/logits")
    assert rel_err(b.float(), truth) <= 3 * rel_err(g.t("bf16/logits"), truth) + 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["b_small", "c_large", "e_adaln_sandwich"])
def test_gpu_checkpoint_round_trip_matches_golden_logits(tmp_path, name):
    """N2 through the real kernels (models/dit.py:1095 schema, model_setup.py:914-923 file layout): a reference-style file (bf16-free fp32 masters
    behind the torch.compile + DDP prefixes) -> `load_backbone_checkpoint` into a module that has ALREADY run (stale bf16 shadows must be rebuilt)
    -> GPU forward equals the golden logits of the imported reference; `save_backbone_checkpoint` of that module -> a fresh module -> bit-identical
    forward."""
    from ledger import check

    g = Golden(name)
    dev = "cuda"
    xt = g.t("fp32/xt").to(dev)
    sigma = O.loglinear_noise(g.t("fp32/t"))[0].to(dev)
    modality = g.t("fp32/modality").to(dev) if g.has("fp32/modality") and g.case["multimodal_batches"] else None
    torch.manual_seed(3)
    diff = Diffusion(product_config(g.case), None, dev)   # random init
    diff.backbone.train()
    with torch.no_grad():
        before = diff.backbone(xt, sigma, modality=modality)   # builds shadows of the random weights
    ckpt = _reference_style_file(g, str(tmp_path / "ref"), prefix="module._orig_mod.")
    missing, unexpected = load_backbone_checkpoint(diff.backbone, ckpt)
    assert missing == [] and unexpected == []
    with torch.no_grad():
        after = diff.backbone(xt, sigma, modality=modality)
    truth = g.t("fp32/logits")
    assert not torch.equal(before, after)
    check(f"checkpoint_logits[{name}]", "logits_relrms_vs_fp32_reference", rel_err(after.float().cpu(), truth), 1e-2)
    f = save_backbone_checkpoint(diff.backbone, str(tmp_path / "out"))
    saved, ref = read_state_dict(f), g.params()
    assert set(saved) == set(ref) and all(torch.equal(saved[k], ref[k].float()) for k in ref)
    other = Diffusion(product_config(g.case), None, dev)
    load_backbone_checkpoint(other.backbone, f)
    other.backbone.train()
    with torch.no_grad():
        again = other.backbone(xt, sigma, modality=modality)
    assert torch.equal(again, after)
