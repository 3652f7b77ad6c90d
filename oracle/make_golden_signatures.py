"""Freeze the CALL SIGNATURES of the reference's boundary (build container only; TEST INFRASTRUCTURE).

    python -m oracle.make_golden_signatures        # writes tests/golden/signatures.json

The drop-in boundary of this repository is `models/dit.py::DIT` and the hot-path methods of `model.py::Diffusion`.  State-dict keys are
frozen by the golden parameter fixtures; this script freezes what a caller TYPES: for every boundary callable the ordered parameter list with kinds and
defaults (`inspect.signature` of the imported reference), and the field list of the `Loss` record.  tests/test_signatures.py compares the product's.
A fixture is data: names, kinds and `repr` of defaults - no reference source text.

Reference callables (file:line in /root/reference): models/dit.py:1096 DIT.__init__, :1324 DIT.forward; model.py:157 update_batch, :397 get_cond_dict,
:420 training_step, :424 q_xt, :589 _sample_t, :621 _subs_parameterization, :660 _process_sigma, :674 forward, :797 compute_loss;
model_utils.py:110 Loss.
"""
from __future__ import annotations

import dataclasses
import inspect
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "signatures.json")

DIFFUSION_METHODS = ["update_batch", "get_cond_dict", "training_step", "q_xt", "_sample_t", "_subs_parameterization", "_process_sigma", "forward", "compute_loss"]
DIT_METHODS = ["__init__", "forward"]


def describe(fn):
    """[{name, kind, default}] of a callable; defaults as repr (None / numbers / bools / strings are all the boundary uses)."""
    out = []
    for p in inspect.signature(fn).parameters.values():
        out.append(dict(name=p.name, kind=p.kind.name, default=None if p.default is inspect.Parameter.empty else repr(p.default),
                        has_default=p.default is not inspect.Parameter.empty))
    return out


def main():
    from oracle import ref_shim
    from oracle.cases import lumina_rope_2d

    ref_shim.install()
    ref_shim.install_lumina_rope(lumina_rope_2d)
    import model as refmodel
    import model_utils as refutils
    import models.dit as refdit

    sig = {"reference": "alexanderswerdlow/unidisc (checkout under /root/reference)", "DIT": {}, "Diffusion": {}}
    for m in DIT_METHODS:
        sig["DIT"][m] = describe(getattr(refdit.DIT, m))
    for m in DIFFUSION_METHODS:
        sig["Diffusion"][m] = describe(getattr(refmodel.Diffusion, m))
    sig["Diffusion"]["__init__"] = describe(refmodel.Diffusion.__init__)
    sig["Loss"] = [dict(name=f.name, default=repr(f.default) if f.default is not dataclasses.MISSING else None) for f in dataclasses.fields(refutils.Loss)]
    with open(OUT, "w") as f:
        json.dump(sig, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT, {k: (len(v) if isinstance(v, (dict, list)) else v) for k, v in sig.items()})


if __name__ == "__main__":
    main()
