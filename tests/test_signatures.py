"""The drop-in boundary's CALL SIGNATURES are pinned against the imported reference (tests/golden/signatures.json, written by
oracle/make_golden_signatures.py from /root/reference): a caller of `models/dit.py::DIT` / `model.py::Diffusion` can pass the same positional and keyword
arguments to `unidisc_amd.DIT` / `unidisc_amd.Diffusion`.  Rule per callable: the reference's parameters appear in the product in the same ORDER with the same
kind and the same default; the product may append keyword parameters WITH defaults (extensions: `backbone=`, `autocast_dtype=` ...), never insert before or
between the reference's positional ones."""
import dataclasses
import inspect
import json
import os

import pytest

import unidisc_amd
from unidisc_amd.diffusion import Loss

SIG = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "signatures.json")))
CASES = [("DIT", m) for m in SIG["DIT"]] + [("Diffusion", m) for m in SIG["Diffusion"]]


def _describe(fn):
    return [dict(name=p.name, kind=p.kind.name, has_default=p.default is not inspect.Parameter.empty,
                 default=None if p.default is inspect.Parameter.empty else repr(p.default)) for p in inspect.signature(fn).parameters.values()]


@pytest.mark.parametrize("cls,method", CASES)
def test_signature_accepts_every_reference_call(cls, method):
    ref = SIG[cls][method]
    got = _describe(getattr(getattr(unidisc_amd, cls), method))
    by_name = {p["name"]: (i, p) for i, p in enumerate(got)}
    ref_named = [p for p in ref if p["kind"] not in ("VAR_KEYWORD", "VAR_POSITIONAL")]
    # every named reference parameter exists with the same kind and default
    for p in ref_named:
        assert p["name"] in by_name, f"{cls}.{method}: parameter {p['name']!r} of the reference is missing"
        q = by_name[p["name"]][1]
        assert q["kind"] == p["kind"], (cls, method, p["name"], q["kind"], p["kind"])
        assert q["has_default"] == p["has_default"] and q["default"] == p["default"], (cls, method, p["name"], q["default"], p["default"])
    # same relative order, and the positional prefix is identical (positional calls bind the same way)
    order = [by_name[p["name"]][0] for p in ref_named]
    assert order == sorted(order), f"{cls}.{method}: parameter order differs from the reference"
    n_pos = len([p for p in ref_named if p["kind"] == "POSITIONAL_OR_KEYWORD"])
    assert [p["name"] for p in got[:n_pos]] == [p["name"] for p in ref_named[:n_pos]]
    # the reference's **kwargs catch-all is kept where it has one
    if any(p["kind"] == "VAR_KEYWORD" for p in ref):
        assert any(p["kind"] == "VAR_KEYWORD" for p in got), f"{cls}.{method}: the reference accepts **kwargs"
    # extensions must be optional
    ref_names = {p["name"] for p in ref}
    for p in got:
        if p["name"] not in ref_names and p["kind"] not in ("VAR_KEYWORD", "VAR_POSITIONAL"):
            assert p["has_default"], f"{cls}.{method}: extension parameter {p['name']!r} must have a default"


def test_loss_record_fields():
    got = [dict(name=f.name, default=repr(f.default) if f.default is not dataclasses.MISSING else None) for f in dataclasses.fields(Loss)]
    assert got == SIG["Loss"]
