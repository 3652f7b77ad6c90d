"""Golden index streams for the token data path, drawn from the IMPORTED reference sampler (build container only).

    python -m oracle.make_golden_tokens      # writes tests/golden/token_sampler.npz

TEST INFRASTRUCTURE.  Runs `WeightedDatasetSampler` of /root/reference/unidisc/datasets/sampler.py:12 over dummy datasets of given sizes and
records the first N (dataset_idx, element_idx) pairs per case, with the sizes / weights / multinomial block size / generator seed that
produced them.  The fixture is data only; this script is the recipe.
"""
from __future__ import annotations

import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden", "token_sampler.npz")

CASES = [
    dict(name="three_mixed", sizes=(6, 4, 9), weights=(0.5, 0.2, -1.0), batch_size=16, seed=123, draws=300),
    dict(name="single", sizes=(5,), weights=(1.0,), batch_size=7, seed=1, draws=40),
    dict(name="two_big_block", sizes=(8, 12), weights=(1.0, 3.0), batch_size=100000, seed=7, draws=120),
    dict(name="two_small_block", sizes=(3, 5), weights=(0.3, 0.7), batch_size=8, seed=11, draws=90),
]


def main():
    from oracle import ref_shim
    ref_shim.install()
    from unidisc.datasets.sampler import WeightedDatasetSampler as RefSampler

    out = {}
    for c in CASES:
        class _Combined:
            dataset_names = [f"d{i}" for i in range(len(c["sizes"]))]
            datasets = [list(range(s)) for s in c["sizes"]]
            weights = list(c["weights"])

        g = torch.Generator().manual_seed(c["seed"])
        s = RefSampler(_Combined(), generator=g, batch_size=c["batch_size"])
        it = iter(s)
        pairs = np.array([next(it) for _ in range(c["draws"])], dtype=np.int64)
        n = c["name"]
        out[n + "/sizes"] = np.array(c["sizes"], dtype=np.int64)
        out[n + "/weights"] = np.array(c["weights"], dtype=np.float64)
        out[n + "/batch_size"] = np.array(c["batch_size"], dtype=np.int64)
        out[n + "/seed"] = np.array(c["seed"], dtype=np.int64)
        out[n + "/stream"] = pairs
        out[n + "/len"] = np.array(len(s), dtype=np.int64)
        print(n, "len", len(s), "counts", s.counts, "first", pairs[:6].tolist())
    np.savez_compressed(GOLDEN, **out)
    print("wrote", GOLDEN)


if __name__ == "__main__":
    main()
