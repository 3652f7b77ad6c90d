"""Golden vectors for the packing collate, made by running the IMPORTED reference `PackingCollate` (build container only).

    python -m oracle.make_golden_packing      # writes tests/golden/packing_collate.npz

TEST INFRASTRUCTURE.  Reference code exercised: /root/reference/dataloader.py:543-678 (`process_batch`, `ignore_slice`, `PackingCollate.__call__`).

One substitution, stated here: `tensordict` (un-vendored, pyproject pin) is not installed, so samples travel in `TD` below - a dict of tensors with
the handful of TensorDict behaviours this collate uses (key access, `in`, leading-dim slicing as views, slice assignment from another TD,
`new_zeros`, `shape`, `batch_size`, `auto_batch_size_`, `len`).  The collate's own logic (queueing, the 1/4-length rule, sample-id assignment,
truncation, the trailing-image rule with the <image> / EOS tokens, random refills drawn from the generator) is the reference's code, run as is.
The tokenizer is a three-attribute stand-in (pad / eos ids and a call that maps "<image>" to its id).
"""
from __future__ import annotations

import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden", "packing_collate.npz")

PAD, EOS, IMG_TOK = 0, 2, 7
KEYS = ("input_ids", "attention_mask", "modality", "sample_ids")


class TD:
    def __init__(self, d):
        self.d = dict(d)

    def __contains__(self, k):
        return k in self.d

    def __getitem__(self, k):
        if isinstance(k, str):
            return self.d[k]
        return TD({n: v[k] for n, v in self.d.items()})     # views, like TensorDict indexing

    def __setitem__(self, k, v):
        if isinstance(k, str):
            self.d[k] = v
        else:
            for n in self.d:
                self.d[n][k] = v.d[n] if isinstance(v, TD) else v

    def __delitem__(self, k):
        del self.d[k]

    def __len__(self):
        return next(iter(self.d.values())).shape[0]

    @property
    def shape(self):
        return next(iter(self.d.values())).shape[:1]

    @property
    def batch_size(self):
        return self.shape

    def auto_batch_size_(self):
        return self

    def new_zeros(self, shape):
        return TD({n: v.new_zeros(shape) for n, v in self.d.items()})


class Tok:
    pad_token_id, eos_token_id = PAD, EOS

    def __call__(self, text, add_special_tokens=False):
        assert text == "<image>"
        return {"input_ids": [IMG_TOK]}


def make_sample(g, parts, pad_to=None, with_sample_ids=True, Vt=50, V=90):
    """parts: list of ("t", n) / ("i", n) / ("img_tok",) / ("eos",); padded with PAD / modality 0 / attention 0 / sample id -1 up to pad_to."""
    ids, mod = [], []
    for p in parts:
        if p[0] == "t":
            ids.append(torch.randint(10, Vt, (p[1],), generator=g)); mod.append(torch.zeros(p[1], dtype=torch.int64))
        elif p[0] == "i":
            ids.append(torch.randint(Vt, V, (p[1],), generator=g)); mod.append(torch.ones(p[1], dtype=torch.int64))
        elif p[0] == "img_tok":
            ids.append(torch.tensor([IMG_TOK])); mod.append(torch.zeros(1, dtype=torch.int64))
        elif p[0] == "eos":
            ids.append(torch.tensor([EOS])); mod.append(torch.zeros(1, dtype=torch.int64))
    ids, mod = torch.cat(ids), torch.cat(mod)
    n = ids.numel()
    tot = pad_to or n
    out = dict(input_ids=torch.full((tot,), PAD, dtype=torch.int64), modality=torch.zeros(tot, dtype=torch.int64),
               attention_mask=torch.zeros(tot, dtype=torch.bool), sample_ids=torch.full((tot,), -1, dtype=torch.int64))
    out["input_ids"][:n], out["modality"][:n], out["attention_mask"][:n], out["sample_ids"][:n] = ids, mod, True, 0
    if not with_sample_ids:
        del out["sample_ids"]
    return out


def build_case(name):
    g = torch.Generator().manual_seed({"mixed": 5, "long_samples": 6, "no_packing": 7}[name])
    if name == "mixed":          # two datasets to refill from; rows end inside an image (trailing-image rule), inside text, and exactly full
        seq, B = 96, 4
        ds = [[make_sample(g, [("t", 6), ("img_tok",), ("i", 16), ("t", 5), ("eos",)], pad_to=40) for _ in range(5)],
              [make_sample(g, [("t", 9), ("eos",), ("img_tok",), ("i", 16)], pad_to=32) for _ in range(4)]]
        first = [make_sample(g, [("t", 12), ("img_tok",), ("i", 16), ("eos",)], pad_to=40), make_sample(g, [("t", 30), ("eos",)], pad_to=40),
                 make_sample(g, [("img_tok",), ("i", 16), ("t", 20), ("eos",)], pad_to=40), make_sample(g, [("t", 3), ("img_tok",), ("i", 16)], pad_to=40)]
    elif name == "long_samples":  # samples longer than what is left: truncation, the 1/4-length rule ends a row early
        seq, B = 64, 3
        ds = [[make_sample(g, [("t", 40), ("eos",)], pad_to=48) for _ in range(3)], [make_sample(g, [("t", 10), ("img_tok",), ("i", 16), ("eos",)]) for _ in range(3)]]
        first = [make_sample(g, [("t", 50), ("eos",)], pad_to=56), make_sample(g, [("t", 20), ("img_tok",), ("i", 16), ("t", 8)], pad_to=48),
                 make_sample(g, [("t", 70), ("eos",)])]
    elif name == "no_packing":    # data.disable_packing: one sample per row
        seq, B = 48, 2
        ds = [[make_sample(g, [("t", 8), ("eos",)], pad_to=16) for _ in range(2)]]
        first = [make_sample(g, [("t", 10), ("img_tok",), ("i", 16), ("eos",)], pad_to=32), make_sample(g, [("t", 5), ("img_tok",), ("i", 16)], pad_to=48)]
    else:
        raise KeyError(name)
    return seq, B, ds, first


CASES = ("mixed", "long_samples", "no_packing")


def main():
    from oracle import ref_shim
    ref_shim.install()
    import dataloader as ref

    out = {}
    for name in CASES:
        seq, B, ds, first = build_case(name)

        class _DS:
            datasets = [[TD(s) for s in d] for d in ds]

            def __getitem__(self, k):
                return TD({n: v.clone() for n, v in self.datasets[k[0]][k[1]].d.items()})

        class _Cfg:
            class data:
                disable_packing = name == "no_packing"

        gen = torch.Generator().manual_seed(77)
        collate = ref.PackingCollate(_Cfg, _DS(), seq, gen, tensor_collate=None, tokenizer=Tok())
        res = collate([TD({n: v.clone() for n, v in s.items()}) for s in first])
        out[f"{name}/seq_length"], out[f"{name}/seed"] = np.array(seq), np.array(77)
        out[f"{name}/disable_packing"] = np.array(name == "no_packing")
        for i, s in enumerate(first):
            for k, v in s.items():
                out[f"{name}/first/{i}/{k}"] = v.numpy()
        for di, d in enumerate(ds):
            for i, s in enumerate(d):
                for k, v in s.items():
                    out[f"{name}/ds/{di}/{i}/{k}"] = v.numpy()
        for k in KEYS:
            out[f"{name}/out/{k}"] = res[k].numpy()
        nxt = int(torch.randint(1 << 30, (1,), generator=gen))   # pins how many draws the collate consumed
        out[f"{name}/next_draw"] = np.array(nxt)
        print(name, "rows (valid tokens, samples):", [(int((res["sample_ids"][b] >= 0).sum()), int(res["sample_ids"][b].max()) + 1) for b in range(B)], "next draw", nxt)
    np.savez_compressed(GOLDEN, **out)
    print("wrote", GOLDEN, os.path.getsize(GOLDEN) // 1024, "KiB")


if __name__ == "__main__":
    main()
