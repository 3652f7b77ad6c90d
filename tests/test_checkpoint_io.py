"""the reference's backbone checkpoints (model.safetensors in DIT.state_dict() names, possibly behind wrapper prefixes)
load into unidisc_amd.DIT, and what unidisc_amd saves has exactly the reference's names, shapes and values.  The reference-side
tensors are the golden fixtures' `params` (dumped from the imported reference by oracle/make_golden.py)."""
import os

import pytest
import torch

import fake_kernels
from golden_utils import CASE_NAMES, Golden, rel_err
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
