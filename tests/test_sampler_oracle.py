"""the oracle's restatement of the `ddpm_cache` sampler loop against golden vectors recorded from the imported reference
(oracle/make_golden_sampler.py): identical uniforms in, identical tokens out at every step."""
import os

import numpy as np
import pytest
import torch

from golden_utils import GOLDEN_DIR, Golden
from oracle import unidisc_oracle as O


def oracle_setup(g):
    cfg = g.cfg
    return cfg, g.params(), g.buffers(), O.update_batch(cfg, g.batch())

SAMPLER_CASES = ["c_large", "b_small"]


def load_sampler(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"sampler_{name}.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


@pytest.mark.parametrize("name", SAMPLER_CASES)
def test_oracle_sampler_matches_reference_tokens(name):
    g = Golden(name)
    s = load_sampler(name)
    cfg, P, buffers, batch = oracle_setup(g)
    steps = int(s["steps"])
    us = [s[f"step{i}/u"] for i in range(steps)]
    modality = s["modality"] if "modality" in s else None
    x0, x0_unmask = (s["x0"], s["x0_unmask"].bool()) if "x0" in s else (None, None)
    x_final, xs, x_last, nfe = O.sample_ddpm_cache(cfg, P, buffers, s["x_init"], s["timesteps"], float(s["dt"]), us, x0=x0, x0_unmask=x0_unmask,
                                                   modality=modality, batch=batch)
    for i in range(steps):
        assert torch.equal(xs[i], s[f"step{i}/x_next"]), f"step {i}"
    assert torch.equal(x_last, s["x_before_noise_removal"]) and torch.equal(x_final, s["x_final"])
    assert nfe == int(s["nfe"])
    # per-step probabilities of the restated forward agree with the reference's to fp32 accuracy
    p0, _ = O.ddpm_forward(cfg, P, buffers, s["step0/x"], O.loglinear_noise(s["timesteps"][0] * torch.ones(s["x_init"].shape[0]))[0], modality, batch)
    assert torch.allclose(p0, s["step0/p_x0"], atol=2e-6, rtol=1e-4)
