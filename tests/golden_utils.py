"""Helpers to load the committed golden fixtures (tests/golden/*.npz, made by oracle/make_golden.py)."""
import os

import numpy as np
import torch

from oracle.cases import ATTN_DROPOUT_CASES, CASES, INTERLEAVED_CASES, lumina_rope_2d
from oracle import unidisc_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_NAMES = sorted(CASES)


class Golden:
    def __init__(self, name):
        self.name = name
        self.case = CASES[name] if name in CASES else (INTERLEAVED_CASES[name] if name in INTERLEAVED_CASES else ATTN_DROPOUT_CASES[name])
        self.z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        self.cfg = O.OracleConfig.from_case(self.case)

    def t(self, key):
        return torch.from_numpy(self.z[key])

    def has(self, key):
        return key in self.z.files

    def params(self, requires_grad=False):
        P = {}
        for k in self.z.files:
            if k.startswith("param/"):
                P[k[6:]] = self.t(k).clone().requires_grad_(requires_grad)
        return P

    def grads(self, tag):
        pre = f"{tag}/grad/"
        return {k[len(pre):]: self.t(k) for k in self.z.files if k.startswith(pre)}

    def buffers(self):
        return {k[7:]: self.t(k) for k in self.z.files if k.startswith("buffer/")}

    def batch(self):
        return {k[6:]: self.t(k).clone() for k in self.z.files if k.startswith("batch/")}

    def generator(self):
        # the reference ran with torch.manual_seed(step_seed) on the global CPU generator
        return torch.Generator().manual_seed(self.case["step_seed"])


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
