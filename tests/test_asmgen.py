"""The hand-scheduled attention forward (csrc/asmgen/attn_fwd64.py) checked on the CPU: the generated instruction stream is executed by the emulator of
csrc/asmgen/isa.py for one workgroup (4 waves, every block it walks) against a float64 attention - once with every memory operation landing as LATE as its
wait allows (a missing / too-loose s_waitcnt or barrier shows as stale data) and once with every refill landing at ISSUE (a refill that overtakes a reader
shows as clobbered data) - the hazard lint is clean, and the committed header is the generator's current output with the full clobber list (a dropped
clobber once shipped a kernel descriptor of 256 registers for a 512-register program)."""
import os
import re
import sys

import pytest

ASMGEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unidisc_amd", "csrc", "asmgen")
sys.path.insert(0, ASMGEN)


@pytest.fixture(scope="module")
def gen():
    import attn_fwd64 as g
    import emu_fwd64 as e
    prog, _ = g.build()
    return g, e, prog


def test_hazard_lint_is_clean(gen):
    g, _, prog = gen
    import isa
    assert isa.lint([i for i in prog if i.kind != "raw"]) == []
    assert g.S_.next <= 100 and g.V.next <= 255 and g.A.next <= 256
    assert g.LDS_TOTAL <= 160 * 1024


@pytest.mark.parametrize("mode", ["late", "early"])
@pytest.mark.parametrize("kw", [dict(B=1, H=8, L=512, grid=8, wg_id=0), dict(B=1, H=8, L=1024, grid=8, wg_id=3), dict(B=2, H=4, L=512, grid=8, wg_id=1, spike=True),
                                # the balanced walk (24 blocks on 16 workgroups: one whole block, then half (wg >> 3) & 1 of block 16 + (wg & 7)): both halves, the
                                # rescale path inside a half block (workgroup 11 ends with rows 640..767 of head 3)
                                dict(B=1, H=8, L=768, grid=16, wg_id=2), dict(B=1, H=8, L=768, grid=16, wg_id=11), dict(B=1, H=8, L=768, grid=16, wg_id=11, spike=True, spike_at=(3, 700, 650))],
                         ids=["two_blocks", "four_blocks_three_trips", "rescale_path", "half_block_first_half", "half_block_second_half", "half_block_rescale"])
def test_emulated_workgroup_matches_float64_attention(gen, kw, mode):
    _, e, prog = gen
    r = e.run(mode=mode, prog=prog, seed=11, **kw)
    assert r["blocks"] >= 2 and r["stray_writes"] == 0
    assert r["o_rel"] < 6e-3, r          # bf16 output rounding
    assert r["lse_err"] < 2e-5, r


def test_committed_header_is_current_and_declares_every_register(tmp_path, gen):
    g, _, _ = gen
    out = tmp_path / "gen.h"
    g.emit(str(out))
    committed = open(os.path.join(os.path.dirname(ASMGEN), "attention_fwd64_gen.h")).read()
    fresh = out.read_text()
    cut = lambda s: s[:s.index("#define UDM_FWD64_ASM_ABL")] if "#define UDM_FWD64_ASM_ABL" in s else s   # (a diagnostic build appends ablation variants)
    assert cut(committed) == cut(fresh), "attention_fwd64_gen.h is stale: run `make -C unidisc_amd/csrc`"
    clob = re.search(r"#define UDM_FWD64_CLOBBERS (.*)", committed).group(1)
    for r in ['"v0"', '"v254"', '"a0"', '"a255"', '"s36"', '"s99"', '"vcc"', '"scc"', '"m0"', '"memory"']:
        assert r in clob, r
    assert '"v255"' not in clob


# ---------------------------------------------------------------------------------------------------------------------------------------------
# the K loop of the one-wave-per-SIMD GEMM (csrc/asmgen/gemm_loop.py -> gemm_loop_gen.h, used by gemm_quad.hip)
# ---------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["late", "early"])
@pytest.mark.parametrize("FM,K", [(5, 256), (4, 256), (5, 512)], ids=["320_rows_4_tiles", "256_rows_4_tiles", "320_rows_8_tiles"])
def test_emulated_gemm_loop_matches_numpy(FM, K, mode):
    import emu_gemm
    r = emu_gemm.run(FM=FM, K=K, mode=mode, seed=3)
    assert r["rel_err"] < 1e-6, r          # fp32 accumulation of exact bf16 products, in the kernel's own k order


@pytest.mark.parametrize("mode", ["late", "early"])
def test_emulated_gemm_loop_16x16x32_form_matches_numpy(mode):
    """the measured-and-not-shipped form of the loop on v_mfma_f32_16x16x32_bf16 (pinned accumulators, single-buffered A fragments): still generated, still correct"""
    import emu_gemm
    r = emu_gemm.run(FM=5, K=256, mode=mode, seed=4, mf16=True)
    assert r["rel_err"] < 1e-6, r


def test_gemm_loop_lint_and_header_current(tmp_path):
    import isa
    import gemm_loop as gl
    out = tmp_path / "gen.h"
    progs = gl.emit(str(out))
    for FM, (g, prog) in progs.items():
        assert isa.lint([i for i in prog if i.kind != "raw"]) == [], FM
        assert g.LDS_BYTES <= 160 * 1024 and g.v_last < 192 and 4 * FM + len(g.INPUTS) + 1 <= 30     # (an asm statement takes at most 30 operands)
    committed = open(os.path.join(os.path.dirname(ASMGEN), "gemm_loop_gen.h")).read()
    assert committed == out.read_text(), "gemm_loop_gen.h is stale (or a diagnostic build): run `make -C unidisc_amd/csrc`"
    for FM in (4, 5):
        clob = re.search(rf"#define UDM_QUADLOOP_NT{FM}_CLOBBERS (.*)", committed).group(1)
        for r in ['"v0"', f'"v{progs[FM][0].v_last}"', '"s36"', f'"s{progs[FM][0].s_last}"', '"vcc"', '"scc"', '"m0"', '"memory"']:
            assert r in clob, (FM, r)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# the dK / dV pass of the attention backward (csrc/asmgen/attn_dkv64.py -> attention_dkv64_gen.h, used by attention_dkv64.hip)
# ---------------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gen_dkv():
    import attn_dkv64 as g
    import emu_dkv64 as e
    prog, _ = g.build()
    return g, e, prog


def test_dkv64_hazard_lint_and_budgets(gen_dkv):
    g, _, prog = gen_dkv
    import isa
    assert isa.lint([i for i in prog if i.kind != "raw"], mfma_states=4) == []
    assert g.S_.next <= 100 and g.V.next <= 255 and g.A.next <= 256
    assert g.LDS_TOTAL <= 160 * 1024
    assert (g.NST - 1) * g.STG + g.TILE + 6 * g.PIECE + 3 * 256 + 8 < 65536        # every fragment address is a 16-bit immediate behind ONE base register


@pytest.mark.parametrize("mode", ["late", "early"])
@pytest.mark.parametrize("kw", [dict(B=1, H=8, L=512, grid=8, wg_id=0), dict(B=1, H=8, L=1024, grid=8, wg_id=3), dict(B=2, H=4, L=512, grid=8, wg_id=5),
                                # the balanced walk (24 blocks on 16 workgroups: one whole block, then half (wg >> 3) & 1 of block 16 + (wg & 7)): both halves
                                dict(B=1, H=8, L=768, grid=16, wg_id=2), dict(B=1, H=8, L=768, grid=16, wg_id=11)],
                         ids=["two_blocks", "four_blocks_three_trips", "two_batches", "half_block_first_half", "half_block_second_half"])
def test_emulated_dkv64_workgroup_matches_float64_backward(gen_dkv, kw, mode):
    _, e, prog = gen_dkv
    r = e.run(mode=mode, prog=prog, seed=11, **kw)
    assert r["blocks"] >= 2 and r["stray_writes"] == 0
    assert r["dk_rel"] < 4e-3 and r["dv_rel"] < 4e-3, r      # bf16 output rounding + bf16 P / dS


def test_dkv64_committed_header_is_current(tmp_path, gen_dkv):
    g, _, _ = gen_dkv
    out = tmp_path / "gen.h"
    g.emit(str(out))
    committed = open(os.path.join(os.path.dirname(ASMGEN), "attention_dkv64_gen.h")).read()
    fresh = out.read_text()
    cut = lambda s: s[:s.index("#define UDM_DKV64_ASM_ABL")] if "#define UDM_DKV64_ASM_ABL" in s else s
    assert cut(committed) == cut(fresh), "attention_dkv64_gen.h is stale: run `make -C unidisc_amd/csrc regen`"
    clob = re.search(r"#define UDM_DKV64_CLOBBERS (.*)", committed).group(1)
    for r in ['"v0"', '"v254"', '"a0"', '"a255"', '"s36"', '"s99"', '"vcc"', '"scc"', '"m0"', '"memory"']:
        assert r in clob, r
    assert '"v255"' not in clob


# ---------------------------------------------------------------------------------------------------------------------------------------------
# the dQ pass of the attention backward (csrc/asmgen/attn_dq64.py -> attention_dq64_gen.h, used by attention_dq64.hip)
# ---------------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gen_dq():
    import attn_dq64 as g
    import emu_dq64 as e
    prog, _ = g.build()
    return g, e, prog


def test_dq64_hazard_lint_and_budgets(gen_dq):
    g, _, prog = gen_dq
    import isa
    assert isa.lint([i for i in prog if i.kind != "raw"], mfma_states=4) == []
    assert g.S_.next <= 100 and g.V.next <= 255 and g.A.next <= 256
    assert g.LDS_TOTAL <= 160 * 1024
    assert (g.NST - 1) * g.STG + g.TILE + 6 * g.PIECE + 3 * 256 + 8 < 65536


@pytest.mark.parametrize("mode", ["late", "early"])
@pytest.mark.parametrize("kw", [dict(B=1, H=8, L=512, grid=8, wg_id=0), dict(B=1, H=8, L=1024, grid=8, wg_id=3), dict(B=2, H=4, L=512, grid=8, wg_id=5),
                                dict(B=1, H=8, L=768, grid=16, wg_id=2), dict(B=1, H=8, L=768, grid=16, wg_id=11)],
                         ids=["two_blocks", "four_blocks_three_trips", "two_batches", "half_block_first_half", "half_block_second_half"])
def test_emulated_dq64_workgroup_matches_float64_backward(gen_dq, kw, mode):
    """dQ of every block the workgroup walks, and the planes delta | -lse | -delta it leaves behind for the dK / dV pass (fp32: to rounding)"""
    _, e, prog = gen_dq
    r = e.run(mode=mode, prog=prog, seed=11, **kw)
    assert r["blocks"] >= 2 and r["stray_writes"] == 0
    assert r["dq_rel"] < 4e-3 and r["planes_err"] < 2e-6, r


def test_dq64_committed_header_is_current(tmp_path, gen_dq):
    g, _, _ = gen_dq
    out = tmp_path / "gen.h"
    g.emit(str(out))
    committed = open(os.path.join(os.path.dirname(ASMGEN), "attention_dq64_gen.h")).read()
    fresh = out.read_text()
    cut = lambda s: s[:s.index("#define UDM_DQ64_ASM_ABL")] if "#define UDM_DQ64_ASM_ABL" in s else s
    assert cut(committed) == cut(fresh), "attention_dq64_gen.h is stale: run `make -C unidisc_amd/csrc regen`"
    clob = re.search(r"#define UDM_DQ64_CLOBBERS (.*)", committed).group(1)
    for r in ['"v0"', '"v254"', '"a0"', '"a255"', '"s36"', '"s99"', '"vcc"', '"scc"', '"m0"', '"memory"']:
        assert r in clob, r
