"""Token data path: sampler index stream vs the reference's (golden), shard schema round trip, batch assembly vs the
oracle's `update_batch` and the reference's own `input_ids` / `attention_mask` / `modality` (golden fixtures)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_kernels  # noqa: E402

from unidisc_amd import token_data as TD  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HAS_GPU = torch.cuda.is_available()


def _golden_batch(name="b_small"):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True)
    return z, {k: z["batch/" + k] for k in ("txt_input_ids", "img_input_ids", "txt_attention_mask")}


@pytest.mark.parametrize("case", ["three_mixed", "single", "two_big_block", "two_small_block"])
def test_weighted_sampler_stream_equals_reference(case):
    z = np.load(os.path.join(GOLDEN, "token_sampler.npz"))
    g = torch.Generator().manual_seed(int(z[case + "/seed"]))
    s = TD.WeightedDatasetSampler(z[case + "/sizes"].tolist(), z[case + "/weights"].tolist(), generator=g, batch_size=int(z[case + "/batch_size"]))
    assert len(s) == int(z[case + "/len"])
    it = iter(s)
    ref = z[case + "/stream"]
    mine = np.array([next(it) for _ in range(len(ref))], dtype=np.int64)
    assert np.array_equal(mine, ref)   # integer stream: exact


def test_weighted_sampler_state_dict_resumes_the_stream():
    g = torch.Generator().manual_seed(5)
    s = TD.WeightedDatasetSampler([6, 4, 9], [0.5, 0.2, -1.0], generator=g, batch_size=16)
    it = iter(s)
    head = [next(it) for _ in range(37)]
    sd = s.state_dict()
    tail = [next(it) for _ in range(50)]
    s2 = TD.WeightedDatasetSampler([6, 4, 9], [0.5, 0.2, -1.0], generator=torch.Generator().manual_seed(999), batch_size=16)
    s2.load_state_dict(sd)
    it2 = iter(s2)
    assert [next(it2) for _ in range(50)] == tail and len(head) == 37


def test_weighted_sampler_rejects_mismatched_weights():
    with pytest.raises(ValueError):
        TD.WeightedDatasetSampler([3, 4], [1.0])


def test_shard_schema_round_trip(tmp_path):
    _, f = _golden_batch()
    TD.TokenShard.write(str(tmp_path / "shard0"), f)
    meta = __import__("json").load(open(tmp_path / "shard0" / "meta.json"))
    assert meta["txt_input_ids"]["dtype"] == "torch.int32" and meta["img_input_ids"]["dtype"] == "torch.int16" and meta["txt_attention_mask"]["dtype"] == "torch.bool"
    assert meta["shape"] == [f["txt_input_ids"].shape[0]]
    sh = TD.TokenShard.open(str(tmp_path / "shard0"))
    assert len(sh) == f["txt_input_ids"].shape[0] and sh.txt_length == f["txt_input_ids"].shape[1] and sh.img_length == f["img_input_ids"].shape[1]
    for k in f:
        assert np.array_equal(np.asarray(sh.fields[k]), f[k])
    with pytest.raises(TypeError):
        TD.TokenShard(dict(txt_input_ids=f["txt_input_ids"].astype(np.int64), img_input_ids=f["img_input_ids"]))
    with pytest.raises(KeyError):
        TD.TokenShard(dict(txt_input_ids=f["txt_input_ids"]))


@pytest.mark.parametrize("variant", ["no_torch_prefix", "nested_entries", "sample_count_only", "wrong_size"])
def test_shard_open_tolerates_layout_variants(tmp_path, variant):
    """The TensorDict memmap layout is restated, not pinned (tensordict is not installed): the reader accepts dtype names without the `torch.` prefix,
    per-key entries under a nested mapping, and a meta.json that only carries the sample count (dtypes from the schema, row length from the file size);
    a file whose size does not fit the sample count is an error, not a silent reshape."""
    import json
    _, f = _golden_batch()
    d = tmp_path / "shard"
    TD.TokenShard.write(str(d), f)
    meta = json.load(open(d / "meta.json"))
    keys = [k for k in meta if isinstance(meta[k], dict)]
    if variant == "no_torch_prefix":
        for k in keys:
            meta[k]["dtype"] = meta[k]["dtype"].replace("torch.", "")
    elif variant == "nested_entries":
        meta = {"shape": meta["shape"], "data": {k: meta[k] for k in keys}}
    else:
        meta = {"shape": meta["shape"]}
        if variant == "wrong_size":
            with open(d / "img_input_ids.memmap", "ab") as fh:
                fh.write(b"\0\0")
    json.dump(meta, open(d / "meta.json", "w"))
    if variant == "wrong_size":
        with pytest.raises(TypeError):
            TD.TokenShard.open(str(d))
        return
    sh = TD.TokenShard.open(str(d))
    for k in f:
        assert np.array_equal(np.asarray(sh.fields[k]), f[k]), k


def _check_batcher(device, resident, monkey_K=None):
    z, f = _golden_batch()
    n = f["txt_input_ids"].shape[0]
    Vt = int(z["case/text_vocab_size"]) if "case/text_vocab_size" in z.files else 32001
    rng = np.random.default_rng(0)
    f2 = {k: np.ascontiguousarray(v[rng.permutation(n)]) for k, v in f.items()}
    shards = [TD.TokenShard(f, "a"), TD.TokenShard(f2, "b")]
    B = 3
    tb = TD.TokenBatcher(shards, [0.6, 0.4], B, Vt, device, seed=3, resident=resident, sampler_block=8)
    # the same sampler, replayed on the host, tells which rows each batch must contain
    ref_it = iter(TD.WeightedDatasetSampler([n, n], [0.6, 0.4], ["a", "b"], generator=torch.Generator().manual_seed(3), batch_size=8))
    from oracle import unidisc_oracle as O
    for _ in range(5):
        batch = tb.next()
        pairs = [next(ref_it) for _ in range(B)]
        rows = {k: torch.from_numpy(np.stack([(f if d == 0 else f2)[k][e] for d, e in pairs])) for k in f}
        want = O.update_batch(type("C", (), dict(text_vocab_size=Vt, force_full_attention_mask=False))(), rows)
        assert torch.equal(batch["input_ids"].cpu(), want["input_ids"]) and batch["input_ids"].dtype == torch.int64
        assert torch.equal(batch["attention_mask"].cpu(), want["attention_mask"]) and batch["attention_mask"].dtype == torch.bool
        assert torch.equal(batch["modality"].cpu(), want["modality"])
        assert batch["dataset_idx"].cpu().tolist() == [d for d, _ in pairs]


@pytest.mark.parametrize("resident", [True, False])
def test_batcher_host_logic_with_kernel_double(monkeypatch, resident):
    monkeypatch.setattr(TD, "K", fake_kernels)
    _check_batcher("cpu", resident)


def test_oracle_assembly_equals_reference_update_batch():
    """The checker itself: oracle.update_batch on the fixture's dataset fields == what the reference's update_batch produced."""
    from oracle import unidisc_oracle as O
    for name in ("b_small", "c_large"):
        z, f = _golden_batch(name)
        rows = {k: torch.from_numpy(v) for k, v in f.items()}
        shift = int(z["fp32/input_ids"][0, -1]) - int(f["img_input_ids"][0, -1])
        want = O.update_batch(type("C", (), dict(text_vocab_size=shift, force_full_attention_mask=False))(), rows)
        assert np.array_equal(want["input_ids"].numpy(), z["fp32/input_ids"])
        assert np.array_equal(want["modality"].numpy(), z["fp32/modality"])


# ------------------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("B,Lt,Li,n,use_idx,use_mask", [(8, 128, 1024, 64, True, True), (3, 5, 7, 3, False, True), (1, 1, 300, 9, True, False),
                                                       (5, 257, 0, 11, True, True), (16, 32, 256, 16, False, False)])
def test_assemble_kernel_bit_exact_vs_oracle(B, Lt, Li, n, use_idx, use_mask):
    from oracle import unidisc_oracle as O
    from unidisc_amd import kernels as K
    g = torch.Generator().manual_seed(B * 1000 + Lt)
    Vt = 32001
    txt = torch.randint(0, Vt - 1, (n, Lt), generator=g, dtype=torch.int32)
    img = torch.randint(0, 16384, (n, Li), generator=g, dtype=torch.int32).to(torch.int16)
    msk = torch.rand(n, Lt, generator=g) < 0.8
    idx = torch.randint(0, n, (B,), generator=g) if use_idx else None
    if not use_idx:
        assert B == n
    ids, mask, modality = K.assemble_joint_tokens(txt.cuda(), msk.cuda() if use_mask else None, img.cuda(), Vt, idx=None if idx is None else idx.cuda())
    sel = idx if idx is not None else torch.arange(n)
    rows = dict(txt_input_ids=txt[sel], img_input_ids=img[sel], txt_attention_mask=msk[sel] if use_mask else torch.ones(B, Lt, dtype=torch.bool))
    if Li == 0:   # text-only shard: the oracle's branch needs an image field; restate directly
        want = dict(input_ids=txt[sel].to(torch.int64), attention_mask=rows["txt_attention_mask"], modality=torch.zeros(B, Lt, dtype=torch.int64))
    else:
        want = O.update_batch(type("C", (), dict(text_vocab_size=Vt, force_full_attention_mask=False))(), rows)
    assert torch.equal(ids.cpu(), want["input_ids"])          # integer / byte work: bit-exact
    assert torch.equal(mask.cpu(), want["attention_mask"])
    assert torch.equal(modality.cpu(), want["modality"])


@pytest.mark.gpu
def test_assemble_kernel_reproduces_reference_fixture():
    from unidisc_amd import kernels as K
    for name in ("b_small", "c_large"):
        z, f = _golden_batch(name)
        shift = int(z["fp32/input_ids"][0, -1]) - int(f["img_input_ids"][0, -1])
        ids, mask, modality = K.assemble_joint_tokens(torch.from_numpy(f["txt_input_ids"]).cuda(), torch.from_numpy(f["txt_attention_mask"]).cuda(),
                                                     torch.from_numpy(f["img_input_ids"]).cuda(), shift)
        assert np.array_equal(ids.cpu().numpy(), z["fp32/input_ids"]) and np.array_equal(modality.cpu().numpy(), z["fp32/modality"])


@pytest.mark.gpu
def test_assemble_kernel_rejects_host_tensors_and_wrong_dtypes():
    from unidisc_amd import kernels as K
    txt = torch.zeros(2, 4, dtype=torch.int32)
    img = torch.zeros(2, 4, dtype=torch.int16)
    with pytest.raises((RuntimeError, ValueError)):
        K.assemble_joint_tokens(txt, None, img, 10)
    with pytest.raises(TypeError):
        K.assemble_joint_tokens(txt.cuda().to(torch.int64), None, img.cuda(), 10)


@pytest.mark.gpu
@pytest.mark.parametrize("resident", [True, False])
def test_batcher_on_device(resident):
    _check_batcher("cuda", resident)

# ------------------------------------------------------------------------------------------------ packing collate (dataloader.py:564-678)
def _packing_case(name):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "packing_collate.npz"))
    def group(prefix):
        items = {}
        for k in z.files:
            if k.startswith(prefix):
                idx, field = k[len(prefix):].rsplit("/", 1)
                items.setdefault(idx, {})[field] = torch.from_numpy(z[k])
        return items
    first = group(f"{name}/first/")
    first = [first[str(i)] for i in range(len(first))]
    ds_flat = group(f"{name}/ds/")
    nds = 1 + max(int(k.split("/")[0]) for k in ds_flat)
    datasets = [[ds_flat[f"{d}/{i}"] for i in range(sum(1 for k in ds_flat if k.startswith(f"{d}/")))] for d in range(nds)]
    return z, first, datasets


@pytest.mark.parametrize("name", ["mixed", "long_samples", "no_packing"])
def test_packing_collate_matches_reference(name):
    from unidisc_amd.token_data import PackingCollate

    z, first, datasets = _packing_case(name)

    class DS:
        def __init__(self):
            self.datasets = datasets

        def __getitem__(self, k):
            return {n: v.clone() for n, v in self.datasets[k[0]][k[1]].items()}

    class Cfg:
        class data:
            disable_packing = bool(z[f"{name}/disable_packing"])

    gen = torch.Generator().manual_seed(int(z[f"{name}/seed"]))
    collate = PackingCollate(Cfg, DS(), int(z[f"{name}/seq_length"]), gen, pad_token_id=0, eos_token_id=2, image_token_id=7)
    out = collate([{k: v.clone() for k, v in s.items()} for s in first])
    for k in PackingCollate.KEYS:
        ref = torch.from_numpy(z[f"{name}/out/{k}"])
        assert out[k].dtype == ref.dtype and torch.equal(out[k], ref), (name, k)
    assert int(torch.randint(1 << 30, (1,), generator=gen)) == int(z[f"{name}/next_draw"])   # same number of draws from the generator


def test_packing_collate_derives_sample_ids_and_feeds_update_batch():
    """Samples without `sample_ids` (valid prefix = everything before the first pad token), then the packed rows through `update_batch`'s
    interleaved tail (padding gets sample id -1 / attention False) - the batch layout the a19 path consumes."""
    from unidisc_amd.token_data import PackingCollate

    z, first, datasets = _packing_case("mixed")
    strip = lambda s: {k: v.clone() for k, v in s.items() if k != "sample_ids"}

    class DS:
        def __init__(self):
            self.datasets = [[strip(s) for s in d] for d in datasets]

        def __getitem__(self, k):
            return strip(self.datasets[k[0]][k[1]])

    gen = torch.Generator().manual_seed(int(z["mixed/seed"]))
    out = PackingCollate(None, DS(), int(z["mixed/seq_length"]), gen, pad_token_id=0, eos_token_id=2, image_token_id=7)([strip(s) for s in first])
    for k in PackingCollate.KEYS:
        assert torch.equal(out[k], torch.from_numpy(z[f"mixed/out/{k}"])), k
    valid = out["sample_ids"] >= 0
    assert torch.equal(valid, out["attention_mask"].bool()) and torch.equal(out["modality"] >= 0, valid)
    for b in range(valid.shape[0]):   # documents are contiguous and numbered 0, 1, 2, ... along the row
        s = out["sample_ids"][b][valid[b]]
        assert bool((s[1:] - s[:-1] >= 0).all()) and int(s[0]) == 0 and set(s.tolist()) == set(range(int(s.max()) + 1))
