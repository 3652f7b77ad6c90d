"""The packed-row layout kernels (tokens.hip: udm_interleaved_rope, udm_interleaved_block_lottery, udm_rowgroup_sum_f32) against the tensor-statement forms they
replace (`DIT._rotary_interleaved_torch`, the statement path of `Diffusion._interleaved_block_lottery`, `index_add_`) on random packed layouts: supported and
unsupported image block sizes, several images per sample, samples of one token, padding tails, a row without images, a row that is one image run to its end.
Integer / table-copy work: bit-identical.  The statement forms are pinned against the imported reference by tests/golden/f_interleaved.npz (tests/test_interleaved.py)."""
import os

import pytest
import torch

import unidisc_amd.diffusion as diff_mod
from golden_utils import Golden
from oracle.cases import INTERLEAVED_CASES
from product_utils import build_product

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _layout(B, L, seed, sizes=(64, 256, 1024, 2304, 4096)):
    """random packed rows: samples of [text, image, text, image ...] pieces, image pieces mostly of a supported size that fits, padding at the end"""
    g = torch.Generator().manual_seed(seed)
    mod = torch.zeros(B, L, dtype=torch.int64)
    sid = torch.full((B, L), -1, dtype=torch.int64)
    for b in range(B):
        pos, s = 0, 0
        limit = L - int(torch.randint(0, L // 6 + 1, (1,), generator=g)) if b % 2 else L
        if b == B - 1 and B > 2:      # one row: a single image run up to the very end
            sid[b], mod[b, 7:] = 0, 1
            continue
        while pos < limit:
            n_pieces = int(torch.randint(1, 5, (1,), generator=g))
            for _ in range(n_pieces):
                if pos >= limit:
                    break
                if torch.rand(1, generator=g) < 0.5:
                    n = int(torch.randint(1, 40, (1,), generator=g))           # text (also 1-4 token pieces: never candidates)
                    n = min(n, limit - pos)
                    sid[b, pos:pos + n] = s
                else:
                    fits = [z for z in sizes if z <= limit - pos]
                    n = fits[int(torch.randint(0, len(fits), (1,), generator=g))] if fits and torch.rand(1, generator=g) < 0.8 else int(torch.randint(1, 90, (1,), generator=g))
                    n = min(n, limit - pos)
                    sid[b, pos:pos + n] = s
                    mod[b, pos:pos + n] = 1
                pos += n
            s += 1
    return mod, sid


@pytest.fixture(scope="module")
def product():
    """the golden interleaved configuration widened to rows of 4608 positions (text table long enough for every test row); weights are irrelevant here"""
    from product_utils import product_config
    from unidisc_amd import Diffusion

    case = dict(INTERLEAVED_CASES[sorted(INTERLEAVED_CASES)[0]], txt_length=512, img_length=4096)
    diff = Diffusion(product_config(case), None, DEV)
    return None, diff


@pytest.mark.parametrize("B,L,seed", [(2, 4608, 1), (3, 1536, 2), (5, 777, 3), (1, 300, 4), (8, 2304, 5)])
def test_interleaved_rope_kernel_equals_the_statement_form(product, B, L, seed):
    _, diff = product
    bb = diff.backbone
    mod, sid = _layout(B, L, seed, sizes=tuple(n for n, _ in bb.IMG_BLOCKS))
    assert L <= bb.rotary_cos_emb_txt.shape[0]
    mod, sid = mod.to(DEV), sid.to(DEV)
    c0, s0, j0 = bb._rotary_interleaved_torch(mod, sid)
    c1, s1, j1 = bb._rotary_interleaved(mod, sid)
    assert torch.equal(c0, c1) and torch.equal(s0, s1) and torch.equal(j0, j1)
    assert (j1 >= 0).any() or L < 64


@pytest.mark.parametrize("B,L,seed", [(2, 4608, 11), (3, 1536, 12), (5, 777, 13), (1, 300, 14), (8, 2304, 15)])
def test_block_lottery_kernel_equals_the_statement_form(product, monkeypatch, B, L, seed):
    _, diff = product
    mod, sid = _layout(B, L, seed)
    batch = dict(modality=mod.to(DEV), sample_ids=sid.to(DEV))
    diff.rng_device = "cpu"
    for p in (0.2, 0.9):
        monkeypatch.setenv("UDM_INTERLEAVED_KERNELS", "0")
        torch.manual_seed(seed)
        a0, h0 = diff._interleaved_block_lottery(batch, p, (B, L), torch.device(DEV))
        after0 = torch.rand(1)
        monkeypatch.setenv("UDM_INTERLEAVED_KERNELS", "1")
        torch.manual_seed(seed)
        a1, h1 = diff._interleaved_block_lottery(batch, p, (B, L), torch.device(DEV))
        after1 = torch.rand(1)
        assert torch.equal(a0, a1) and torch.equal(h0, h1)
        assert torch.equal(after0, after1)          # the replay consumed exactly the reference's number of uniforms
        if p > 0.5:
            assert a1.any()
    # device generator: no host read, every candidate still gets its own uniform (statistics, not values: p = 1 masks every candidate block, p = 0 none)
    diff.rng_device = None
    a1, h1 = diff._interleaved_block_lottery(batch, 0.0, (B, L), torch.device(DEV))
    assert not a1.any() and not h1.any()
    diff.rng_device = "cpu"


@pytest.mark.parametrize("M,d,G", [(9216, 2048, 16), (777, 768, 16), (64, 64, 3)])
def test_rowgroup_sum_equals_index_add(M, d, G):
    from unidisc_amd import kernels as K

    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, d, generator=g).to(DEV)
    runs = torch.randint(0, G + 1, (M // 37 + 2,), generator=g)                    # runs of equal indices, G = "no group"
    group = runs.repeat_interleave(37)[:M].to(DEV)
    out = torch.full((G, d), 0.5, device=DEV)
    K.rowgroup_sum(x, group, out)
    ref = torch.full((G + 1, d), 0.5, dtype=torch.float64, device=DEV).index_add_(0, group, x.double())[:G]
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4)
