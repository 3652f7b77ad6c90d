"""CPU checks of the C-ABI boundary: the library builds for gfx950, loads, and exports every symbol the header declares."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from unidisc_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from unidisc_amd import _lib

    header = open(os.path.join(ROOT, "include", "unidisc_hip.h")).read()
    declared = set(re.findall(r"\b(udm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    bound = set(_lib.PROTOTYPES) | set(_lib.EXTRA_SYMBOLS)
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.udm_abi_version() == 1


def test_prototype_arity_matches_header(lib):
    from unidisc_amd import _lib

    header = open(os.path.join(ROOT, "include", "unidisc_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for name, args in _lib.PROTOTYPES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", header, flags=re.S)
        assert m, name
        n = len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert n == len(args), (name, n, len(args))


def test_no_cpu_fallback():
    from unidisc_amd import DIT, kernels, make_config, MODEL_PRESETS

    cfg = make_config(**MODEL_PRESETS["tiny"], txt_length=16, img_length=16)
    m = DIT(cfg, 65, 41, 40)
    ids = torch.zeros(2, 32, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(ids, None, modality=torch.zeros_like(ids))
    with pytest.raises(RuntimeError, match="GPU tensors"):
        kernels.gemm_nt(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))


def test_state_dict_schema_matches_reference():
    from golden_utils import CASE_NAMES, Golden
    from product_utils import build_product

    for name in CASE_NAMES:
        g = Golden(name)
        diff = build_product(g, "cpu")  # strict load inside
        sd = diff.backbone.state_dict()
        assert set(sd) == set(g.params()), name
        assert all(v.dtype == torch.float32 for v in sd.values())
