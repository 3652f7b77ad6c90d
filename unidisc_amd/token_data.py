"""Token data path: pre-tokenised dataset shards -> batches resident in HBM.

Mirrors, for the plain (non-interleaved) token datasets the 1.4 B stages train on:
  * the on-disk schema of the reference's token TensorDicts (`models/datasets/image_datasets.py:263-281`): `txt_input_ids` int32 [n, Lt],
    `txt_attention_mask` bool [n, Lt], `img_input_ids` int16 [n, Li] (+ `idx`, `dataset_idx`, `write_flag`, ignored here), stored the way
    `TensorDict.memmap_` lays a flat TensorDict out: `<dir>/meta.json` (per-key shape / dtype) + one raw `<key>.memmap` file per key.
    tensordict itself is not installed in this image, so the layout is restated from its convention and is NOT pinned against the library;
    `TokenShard.write` produces the same layout so the round trip is self-consistent.
  * `WeightedDatasetSampler` (`unidisc/datasets/sampler.py:12-140`): which (dataset, element) comes next.  Same draws from the same
    torch.Generator in the same order (multinomial blocks over the still-available datasets, one randperm per dataset pass), so the index
    stream is identical to the reference's - pinned by `tests/golden/token_sampler.npz`.
  * `update_batch`'s token branch (`model.py:183-212`) as one HIP kernel (`kernels.assemble_joint_tokens`).

MI355X-first layout: a shard is uploaded to HBM once (`TokenShard.to_device`; 2.7 KiB per sample at Lt = 128 / Li = 1024, so 288 GB hold the
whole stage-1 token set) and a batch is B row indices + one gather kernel; nothing but 8 B per sample crosses PCIe per step.  Shards that
should stay on the host go through a pinned staging buffer and an async copy on a side stream instead (`resident=False`).

  * `PackingCollate` (`dataloader.py:564-678`): interleaved samples packed into fixed-length rows with per-row sample ids (what the document
    mask, the per-sample rotary positions and the per-block masking of SURVEY §8 row a19 consume) - host integer logic with data-dependent
    control flow over a handful of samples per row; the finished [B, L] rows go to HBM in one copy.  Pinned by `tests/golden/packing_collate.npz`.

Not built (out of this row's scope, stated in DESIGN.md): raw-image and webdataset branches, tokenizer setup, fault-tolerant distributed samplers.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import kernels as K

_FIELDS = {"txt_input_ids": np.int32, "txt_attention_mask": np.bool_, "img_input_ids": np.int16}
_TORCH_NAME = {np.dtype(np.int32): "torch.int32", np.dtype(np.int16): "torch.int16", np.dtype(np.bool_): "torch.bool", np.dtype(np.int64): "torch.int64"}
_NP_OF = {v: k for k, v in _TORCH_NAME.items()}


class TokenShard:
    """One token dataset (a flat TensorDict memmap directory) opened read-only as numpy memmaps."""

    def __init__(self, fields: Dict[str, np.ndarray], name: str = "shard"):
        missing = [k for k in ("txt_input_ids", "img_input_ids") if k not in fields]
        if missing:
            raise KeyError(f"TokenShard {name}: missing field(s) {missing}")
        n = fields["txt_input_ids"].shape[0]
        for k, want in _FIELDS.items():
            if k in fields:
                a = fields[k]
                if a.dtype != np.dtype(want) or a.ndim != 2 or a.shape[0] != n:
                    raise TypeError(f"TokenShard {name}: field {k} must be {np.dtype(want)} [n, L], got {a.dtype} {a.shape}")
        if "txt_attention_mask" in fields and fields["txt_attention_mask"].shape != fields["txt_input_ids"].shape:
            raise TypeError(f"TokenShard {name}: txt_attention_mask shape differs from txt_input_ids")
        self.fields, self.name = fields, name
        self._dev: Optional[Dict[str, torch.Tensor]] = None

    def __len__(self):
        return self.fields["txt_input_ids"].shape[0]

    @property
    def txt_length(self):
        return self.fields["txt_input_ids"].shape[1]

    @property
    def img_length(self):
        return self.fields["img_input_ids"].shape[1]

    @classmethod
    def open(cls, path: str, name: Optional[str] = None) -> "TokenShard":
        """Open a token TensorDict directory.  The layout is restated (tensordict is not installed here: NOT pinned against the library), so the reader is tolerant
        where library versions differ: dtype names with or without the ``torch.`` prefix, per-key entries at the top level of meta.json or under a nested
        mapping, and - for a key without a usable entry - the dtype the schema prescribes with the row length inferred from the file size and the sample count."""
        meta = {}
        mpath = os.path.join(path, "meta.json")
        if os.path.exists(mpath):
            with open(mpath) as f:
                meta = json.load(f)
        n = None
        if isinstance(meta.get("shape"), (list, tuple)) and meta["shape"]:
            n = int(meta["shape"][0])

        def entry(k):
            for holder in (meta, *(v for v in meta.values() if isinstance(v, dict))):
                e = holder.get(k) if isinstance(holder, dict) else None
                if isinstance(e, dict) and "shape" in e and "dtype" in e:
                    return e
            return None

        fields = {}
        pending = []
        for k, want in _FIELDS.items():
            fpath = os.path.join(path, k + ".memmap")
            if not os.path.exists(fpath):
                continue
            info = entry(k)
            if info is None:
                pending.append((k, want, fpath))
                continue
            tname = str(info["dtype"])
            dt = _NP_OF.get(tname if tname.startswith("torch.") else "torch." + tname)
            if dt is None:
                raise TypeError(f"{path}: unsupported dtype {info['dtype']} for {k}")
            fields[k] = np.memmap(fpath, dtype=dt, mode="r", shape=tuple(int(x) for x in info["shape"]))
            n = fields[k].shape[0] if n is None else n
        for k, want, fpath in pending:   # no usable meta entry: the schema's dtype, row length from the file size
            if n is None or n <= 0:
                raise KeyError(f"{path}: no shape information for {k} (meta.json lists neither the field nor the sample count)")
            size = os.path.getsize(fpath) // np.dtype(want).itemsize
            if size % n != 0:
                raise TypeError(f"{path}: {k}.memmap holds {size} elements, not a multiple of the {n} samples")
            fields[k] = np.memmap(fpath, dtype=want, mode="r", shape=(n, size // n))
        return cls(fields, name or os.path.basename(os.path.normpath(path)))

    @staticmethod
    def write(path: str, fields: Dict[str, np.ndarray]) -> None:
        os.makedirs(path, exist_ok=True)
        n = next(iter(fields.values())).shape[0]
        meta = {"shape": [int(n)], "device": "cpu", "_type": "<class 'tensordict._td.TensorDict'>"}
        for k, a in fields.items():
            a = np.ascontiguousarray(a)
            mm = np.memmap(os.path.join(path, k + ".memmap"), dtype=a.dtype, mode="w+", shape=a.shape)
            mm[...] = a
            mm.flush()
            meta[k] = {"device": "cpu", "shape": [int(s) for s in a.shape], "dtype": _TORCH_NAME[a.dtype]}
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(meta, f)

    def to_device(self, device) -> Dict[str, torch.Tensor]:
        """Upload the shard once; later batches are index gathers on the device."""
        if self._dev is None:
            self._dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in self.fields.items() if k in _FIELDS}
        return self._dev

    def rows(self, idx: np.ndarray) -> Dict[str, np.ndarray]:
        return {k: np.ascontiguousarray(v[idx]) for k, v in self.fields.items() if k in _FIELDS}


class WeightedDatasetSampler:
    """Index stream (dataset_idx, element_idx) of `unidisc/datasets/sampler.py:12-140`.

    Datasets are drawn in blocks of `batch_size` multinomial samples over the datasets that still have quota in this epoch; inside a dataset the
    elements follow a random permutation that is renewed when it runs out.  A dataset's epoch quota is round(weight / sum(weights) * lcm(sizes));
    a negative weight means "proportional to size".  When every quota is used up the state resets and the stream goes on (the reference's
    default `raise_stop_iteration = False`).
    """

    def __init__(self, sizes: Sequence[int], weights: Sequence[float], names: Optional[Sequence[str]] = None, generator: Optional[torch.Generator] = None,
                 batch_size: int = 100000, raise_stop_iteration: bool = False):
        if len(sizes) != len(weights):
            raise ValueError("Each dataset must have a corresponding weight")
        self.sizes = [int(s) for s in sizes]
        self.names = list(names) if names is not None else [f"dataset_{i}" for i in range(len(sizes))]
        self.generator, self.batch_size, self.raise_stop_iteration = generator, int(batch_size), raise_stop_iteration
        lcm = int(np.lcm.reduce(self.sizes))
        self.lcm_size = lcm if lcm >= 1 else max(self.sizes) * 1000
        total = sum(self.sizes)
        self.weights = [w if w >= 0 else s / total for w, s in zip(weights, self.sizes)]
        wsum = sum(self.weights)
        self.counts = {n: int(round(w / wsum * self.lcm_size)) for w, n in zip(self.weights, self.names)}
        self._reset()

    def __len__(self):
        return sum(self.counts.values())

    # -- state ---------------------------------------------------------------------------------------------------------------------
    def _reset(self):
        wsum = sum(self.weights)
        self._block: Optional[List[int]] = None          # dataset ids of the current multinomial block
        self._ptr = 0
        self._used = {n: 0 for n in self.names}
        self._avail = list(range(len(self.names)))
        self._avail_w = [w / wsum for w in self.weights]
        self._perm: Optional[Dict[str, Tuple[torch.Tensor, int]]] = None

    def _start_pass(self):
        if self._perm is None:
            self._perm = {n: (torch.randperm(self.sizes[i], generator=self.generator), 0) for i, n in enumerate(self.names)}

    def _draw_block(self):
        s = sum(self._avail_w)
        p = torch.tensor([w / s for w in self._avail_w])
        picks = torch.multinomial(p, self.batch_size, replacement=True, generator=self.generator)
        self._block = torch.tensor(self._avail).to(picks)[picks].tolist()
        self._ptr = 0

    def _quota_left(self):
        return any(self._used[n] < self.counts[n] for n in self.names)

    def state_dict(self):
        return {"block": self._block, "ptr": self._ptr, "used": dict(self._used), "avail": list(self._avail), "avail_w": list(self._avail_w),
                "perm": None if self._perm is None else {n: (t.clone(), i) for n, (t, i) in self._perm.items()},
                "generator": None if self.generator is None else self.generator.get_state()}

    def load_state_dict(self, sd):
        self._block, self._ptr, self._used = sd["block"], sd["ptr"], dict(sd["used"])
        self._avail, self._avail_w = list(sd["avail"]), list(sd["avail_w"])
        self._perm = None if sd["perm"] is None else {n: (t.clone(), i) for n, (t, i) in sd["perm"].items()}
        if sd["generator"] is not None and self.generator is not None:
            self.generator.set_state(sd["generator"])

    # -- stream ----------------------------------------------------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Tuple[int, int]]:
        self._start_pass()
        while self._quota_left() or not self.raise_stop_iteration:
            if not self._avail or (not self.raise_stop_iteration and not self._quota_left()):
                self._reset()
                if self.raise_stop_iteration:
                    return
                self._start_pass()
            if self._block is None or self._ptr >= self.batch_size:
                self._draw_block()
            d = self._block[self._ptr]
            self._ptr += 1
            name = self.names[d]
            perm, i = self._perm[name]
            if i >= len(perm):
                perm, i = torch.randperm(self.sizes[d], generator=self.generator), 0
            self._perm[name] = (perm, i + 1)
            self._used[name] += 1
            if self._used[name] >= self.counts[name]:
                k = self._avail.index(d)
                self._avail.pop(k)
                self._avail_w.pop(k)
                if self._avail:
                    self._draw_block()
            yield d, int(perm[i])
        self._reset()


class TokenBatcher:
    """Batches of the token-dataset schema, assembled on the device, one step ahead of the consumer.

    next() -> dict(input_ids int64 [B, L], attention_mask bool [B, L], modality int64 [B, L], dataset_idx int64 [B]) on `device`: exactly what the
    token branch of `Diffusion.update_batch` produces from a collated batch (model.py:183-212), so the dict can be handed to `training_step`.
    """

    def __init__(self, shards: Sequence[TokenShard], weights: Sequence[float], batch_size: int, text_vocab_size: int, device, seed: int = 0,
                 resident: bool = True, sampler_block: int = 100000):
        if not shards:
            raise ValueError("TokenBatcher: no shards")
        lt, li = shards[0].txt_length, shards[0].img_length
        for s in shards:
            if (s.txt_length, s.img_length) != (lt, li):
                raise ValueError("TokenBatcher: all shards must share (txt_length, img_length); pad or bucket shards of other shapes")
        self.shards, self.B, self.Vt, self.device, self.resident = list(shards), int(batch_size), int(text_vocab_size), torch.device(device), resident
        self.Lt, self.Li = lt, li
        gen = torch.Generator().manual_seed(seed)
        self.sampler = WeightedDatasetSampler([len(s) for s in shards], weights, [s.name for s in shards], generator=gen, batch_size=sampler_block)
        self._it = iter(self.sampler)
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._pending = None
        if resident:   # one concatenated copy in HBM; a global row = shard offset + element
            self._offsets = np.concatenate([[0], np.cumsum([len(s) for s in shards])]).astype(np.int64)
            dev = [s.to_device(self.device) for s in shards]
            self._txt = torch.cat([d["txt_input_ids"] for d in dev]) if len(dev) > 1 else dev[0]["txt_input_ids"]
            self._img = torch.cat([d["img_input_ids"] for d in dev]) if len(dev) > 1 else dev[0]["img_input_ids"]
            if all("txt_attention_mask" in d for d in dev):
                self._msk = torch.cat([d["txt_attention_mask"] for d in dev]) if len(dev) > 1 else dev[0]["txt_attention_mask"]
            else:
                self._msk = None
            self._idx_host = torch.empty(2, self.B, dtype=torch.int64).pin_memory() if self._stream is not None else torch.empty(2, self.B, dtype=torch.int64)
        else:
            pin = (lambda t: t.pin_memory()) if self._stream is not None else (lambda t: t)
            self._stage = [dict(txt=pin(torch.empty(self.B, lt, dtype=torch.int32)), msk=pin(torch.ones(self.B, lt, dtype=torch.bool)),
                                img=pin(torch.empty(self.B, li, dtype=torch.int16))) for _ in range(2)]
        self._slot = 0

    def _draw(self):
        pairs = [next(self._it) for _ in range(self.B)]
        return np.array([p[0] for p in pairs], dtype=np.int64), np.array([p[1] for p in pairs], dtype=np.int64)

    def _launch(self):
        ds, el = self._draw()
        slot, self._slot = self._slot, self._slot ^ 1
        ctx = torch.cuda.stream(self._stream) if self._stream is not None else _NullCtx()
        with ctx:
            if self.resident:
                self._idx_host[slot].copy_(torch.from_numpy(self._offsets[ds] + el))
                idx = self._idx_host[slot].to(self.device, non_blocking=True)
                ids, mask, modality = K.assemble_joint_tokens(self._txt, self._msk, self._img, self.Vt, idx=idx)
            else:
                st = self._stage[slot]
                for b in range(self.B):   # rows of different shards: B small contiguous copies out of the page cache
                    f = self.shards[ds[b]].fields
                    st["txt"][b].copy_(torch.from_numpy(np.ascontiguousarray(f["txt_input_ids"][el[b]])))
                    st["img"][b].copy_(torch.from_numpy(np.ascontiguousarray(f["img_input_ids"][el[b]])))
                    if "txt_attention_mask" in f:
                        st["msk"][b].copy_(torch.from_numpy(np.ascontiguousarray(f["txt_attention_mask"][el[b]])))
                    else:
                        st["msk"][b].fill_(True)
                txt, msk, img = (st[k].to(self.device, non_blocking=True) for k in ("txt", "msk", "img"))
                ids, mask, modality = K.assemble_joint_tokens(txt, msk, img, self.Vt)
            batch = dict(input_ids=ids, attention_mask=mask, modality=modality, dataset_idx=torch.from_numpy(ds).to(self.device, non_blocking=True))
            ev = torch.cuda.Event() if self._stream is not None else None
            if ev is not None:
                ev.record(self._stream)
        return batch, ev

    def next(self) -> Dict[str, torch.Tensor]:
        if self._pending is None:
            self._pending = self._launch()
        batch, ev = self._pending
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in batch.values():
                t.record_stream(torch.cuda.current_stream(self.device))
        self._pending = self._launch()   # the batch after this one is assembled on the side stream while the step runs
        return batch

    __next__ = next

    def __iter__(self):
        return self


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class PackingCollate:
    """Pack variable-length interleaved samples into rows of `seq_length` tokens (reference: `PackingCollate`, dataloader.py:564-678).

    A sample is a dict of 1-D tensors `input_ids`, `attention_mask`, `modality` (0 text / 1 image) and optionally `sample_ids` (0 on its valid
    prefix, -1 on its padding; derived from the first pad token when absent).  Row i starts with the i-th sample of the incoming batch and is
    topped up with samples drawn from `dataset` - (dataset index, element index), two `torch.randint` draws from `generator` each, the same
    draws in the same order as the reference - until it is full, or until what is left is shorter than a quarter of the next sample.  Each
    packed sample gets the next sample id of its row; a sample that does not fit is truncated.  A row that ends inside an image loses that
    image (and its `<image>` token), after an EOS has been put behind the text before it.  Unused positions: pad id, attention 0,
    modality -1, sample id -1.  Returns a dict of [B, seq_length] tensors with the dtypes of the first sample.
    """

    KEYS = ("input_ids", "attention_mask", "modality", "sample_ids")

    def __init__(self, config, dataset, seq_length, generator, tensor_collate=None, tokenizer=None, *, pad_token_id=None, eos_token_id=None,
                 image_token_id=None):
        self.dataset, self.seq_length, self.generator, self.tensor_collate = dataset, int(seq_length), generator, tensor_collate
        if tokenizer is not None:
            pad_token_id, eos_token_id = tokenizer.pad_token_id, tokenizer.eos_token_id
            toks = tokenizer("<image>", add_special_tokens=False)["input_ids"]
            if len(toks) != 1:
                raise ValueError("PackingCollate: '<image>' must be a single token")
            image_token_id = toks[0]
        if pad_token_id is None or eos_token_id is None or image_token_id is None:
            raise ValueError("PackingCollate needs a tokenizer or pad_token_id / eos_token_id / image_token_id")
        self.padding_token_id, self.eos_token_id, self.image_token_id = int(pad_token_id), int(eos_token_id), int(image_token_id)
        data = getattr(config, "data", None)
        self.disable_packing = bool(getattr(data, "disable_packing", False)) if data is not None else False

    @staticmethod
    def _clean(sample):
        return {k: v for k, v in sample.items() if k not in ("write_flag", "dataset_idx")}

    def _draw(self):
        d = int(torch.randint(len(self.dataset.datasets), (1,), generator=self.generator).item())
        e = int(torch.randint(len(self.dataset.datasets[d]), (1,), generator=self.generator).item())
        return self._clean(self.dataset[(d, e)])

    def _valid_length(self, sample):
        """Number of leading positions that belong to the sample (everything before its first sample id of -1)."""
        if "sample_ids" not in sample:   # before the first pad token: the sample; from it on: padding
            seen_pad = torch.cumsum((sample["input_ids"] == self.padding_token_id).long(), 0) > 0
            sample["sample_ids"] = torch.where(seen_pad, -1, 0).to(sample["input_ids"].dtype)
        sid = sample["sample_ids"]
        if not bool(((sid == 0) | (sid == -1)).all()):   # already packed text (several documents per sample): text only
            assert bool((sample["modality"] == 0).all())
        neg = (sid == -1).nonzero()
        if neg.numel():
            return int(neg[0])
        assert bool(sample["attention_mask"].all())
        return sid.numel()

    def __call__(self, batch):
        if self.tensor_collate is not None:
            batch = [self.tensor_collate(b) for b in batch] if isinstance(batch, list) else self.tensor_collate(batch)
        batch = [self._clean(dict(b)) for b in batch]
        B, S = len(batch), self.seq_length
        proto = batch[0]
        dt = lambda k: proto[k].dtype if k in proto else proto["input_ids"].dtype
        ids = torch.full((B, S), self.padding_token_id, dtype=dt("input_ids"))
        att = torch.zeros((B, S), dtype=dt("attention_mask"))
        mod = torch.full((B, S), -1, dtype=dt("modality"))
        sids = torch.full((B, S), -1, dtype=dt("sample_ids"))
        for i in range(B):
            used, n_packed, queue = 0, 0, [batch[i]]
            while used < S and not (self.disable_packing and n_packed > 0):
                sample = queue.pop(0) if queue else self._draw()
                room = S - used
                if room < sample["input_ids"].shape[0] // 4:
                    if used > 0:
                        break
                    continue
                n = min(self._valid_length(sample), room)
                ids[i, used:used + n], att[i, used:used + n], mod[i, used:used + n] = sample["input_ids"][:n], sample["attention_mask"][:n], sample["modality"][:n]
                sids[i, used:used + n] = n_packed
                used += n
                n_packed += 1
            if mod[i, -1] == 1:   # the row ends inside an image: drop that image
                is_img = mod[i] == 1
                changes = (is_img[:-1] != is_img[1:]).nonzero().flatten() + 1
                if changes.numel():
                    start = int(changes[-1])
                    if start > 0 and ids[i, start - 1] == self.image_token_id:
                        start -= 1
                    if start > 0 and ids[i, start - 1] != self.eos_token_id:
                        ids[i, start], att[i, start], mod[i, start] = self.eos_token_id, 1, 0
                        start += 1
                    ids[i, start:], att[i, start:], mod[i, start:], sids[i, start:] = self.padding_token_id, 0, -1, -1
        return dict(input_ids=ids, attention_mask=att, modality=mod, sample_ids=sids)

    def to_device(self, packed, device):
        """One host-to-device copy per field of the finished rows."""
        return {k: v.to(device, non_blocking=True) for k, v in packed.items()}
