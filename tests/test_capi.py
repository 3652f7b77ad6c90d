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


def test_bench_flops_per_token_match_survey():
    """F_tok = 6 P_mm + 12 n L_att d -> 0.737 (UniDisc-S, L=384), 8.597 (1.4 B, L=1280), 8.522 GFLOP/token (interleaved, L_att=1152)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert abs(bench.flops_per_token(12, 768, 40193, 384) / 1e9 - 0.737) < 1e-3
    assert abs(bench.flops_per_token(24, 2048, 48385, 1280) / 1e9 - 8.597) < 1e-3
    assert abs(bench.flops_per_token(24, 2048, 48385, 1152) / 1e9 - 8.522) < 1e-3
    for name, w in bench.WORKLOADS.items():   # every workload names a configuration of BASELINE.json
        assert w["desc"] and w["batch"] > 0 and w["txt_length"] + w["img_length"] > 0, name
    b = bench.synthetic_batch("unidisc-1.4b-interleaved-l4608", 2, 0)
    assert b["input_ids"].shape == (2, 4608) and int(b["sample_ids"].max()) == 3 and int((b["modality"] == 1).sum()) == 2 * 4096
