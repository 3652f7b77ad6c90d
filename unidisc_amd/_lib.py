"""ctypes binding of ``libunidisc_hip.so`` (C ABI: ``include/unidisc_hip.h``).

This is the stub a maintainer of the reference adds on the Python side; the reference
itself has no FFI for this path (it is pure PyTorch), so the ABI is defined by this project.

There is NO CPU fallback: if the shared library is missing the first kernel call raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunidisc_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_P, _I64, _I, _F, _U64 = c_void_p, c_int64, c_int, c_float, c_uint64

# name -> argument ctypes (the trailing hipStream_t is a pointer); must mirror include/unidisc_hip.h exactly
PROTOTYPES = {
    "udm_gemm_nt_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _I, _P, _P, _I64, _F, _P],
    "udm_gemm_tn_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, _P],
    "udm_gemm_nn_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P],
    "udm_gemm_nn_ok": [_I64, _I64, _I64],
    "udm_gemm_nt_splitk_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I64, _P],
    "udm_gemm_nn_splitk_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I64, _P],
    "udm_gemm_tn_splitk_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, _P, _I64, _P],
    "udm_gemm_tn_pair_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, _P, _I64, _P],
    "udm_gemm_tn_multi_bf16": [_I, _P, _P, _P, _P, _P, _P, _P, _I64, _F, _P, _I64, _P],
    "udm_gemm_set_cus": [_I],
    "udm_debug_set": [ctypes.c_char_p, _I64],
    "udm_debug_cu_hog": [_I64, _P, _P],
    "udm_transpose_bf16": [_P, _P, _I64, _I64, _I64, _I64, _P, _P],
    "udm_small_batch_linear_bwd": [_P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _I64, _P, _I64, _I64, _I64, _P],
    "udm_small_batch_linear_bwd_blocks": [_I64],
    "udm_cast_transpose_f32_bf16": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P],
    "udm_cast_transpose_multi_f32_bf16": [_P, _I64, _I64, _P],
    "udm_cast_f32_bf16": [_P, _P, _I64, _F, _P],
    "udm_cast_bf16_f32": [_P, _P, _I64, _F, _P],
    "udm_norm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _I, _F, _P],
    "udm_norm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _P, _I64, _P],
    "udm_residual_fwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I64, _I, _F, _F, _U64, _P],
    "udm_residual_norm_fwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I64, _I, _F, _F, _U64, _P, _P, _P, _P, _P],
    "udm_residual_norm_fwd_ada": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I64, _I, _F, _F, _U64, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P],
    "udm_residual_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I, _F, _U64, _P, _I64, _P],
    "udm_norm_residual_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _F, _U64, _P, _I64, _P],
    "udm_norm_residual_bwd_ada": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I, _F, _U64,
                                  _P, _I64, _P],
    "udm_qknorm_rope_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I64, _I64, _I64, _F, _F, _P],
    "udm_qknorm_rope_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _P, _I64, _P],
    "udm_attention_doc_ranges": [_P, _I64, _I64, _P, _P],
    "udm_attention_fwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P],
    "udm_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P],
    "udm_assemble_joint_tokens": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P],
    "udm_interleaved_rope": [_P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P],
    "udm_interleaved_block_lottery": [_P, _P, _P, _I64, _F, _I64, _I64, _P, _P, _P, _P, _P, _P],
    "udm_rowgroup_sum_f32": [_P, _P, _P, _I64, _I64, _I64, _P],
    "udm_sample_t_noise": [_P, _I64, _I, _F, _F, _F, _P, _P, _P, _P, _P],
    "udm_qxt_absorbing": [_P, _P, _P, _P, _P, _F, _F, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P],
    "udm_categorical_sample_rows": [_P, _P, _P, _I64, _P, _P, _I64, _U64, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P],
    "udm_ddpm_sample_rows_cfg": [_P, _P, _P, _I64, _P, _P, _P, _P, _I64, _U64, _P, _I64, _I64, _I64, _I64, _I, _I, _P],
    "udm_ddpm_sample_rows": [_P, _I64, _P, _P, _P, _P, _I64, _U64, _P, _I64, _I64, _I64, _I64, _I, _I, _P],
    "udm_sumsq_f32": [_P, _I64, _P, _P, _I64, _P],
    "udm_adamw_step_ema": [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I64, _P, _F, _P, _F, _P],
    "udm_adamw_step_shadow_ema": [_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _I64, _P, _F, _P, _I64, _P, _I64, _P, _F, _P],
    "udm_adamw_step_shadow_multi": [_P, _I64, _I64, _F, _F, _F, _F, _F, _I64, _P, _F, _F, _P],
    "udm_adamw_step_multi": [_P, _I64, _I64, _F, _F, _F, _F, _F, _I64, _P, _F, _F, _P],
    "udm_adamw_step": [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I64, _P, _F, _P],
    "udm_adamw_step_shadow": [_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _I64, _P, _F, _P, _I64, _P, _I64, _P],
    "udm_embedding_fwd": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "udm_embedding_bwd": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "udm_subs_ce_fwd": [_P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P],
    "udm_subs_ce_bwd": [_P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _I64, _P],
    "udm_subs_logprobs": [_P, _I64, _P, _P, _P, _I64, _I, _I64, _I64, _I64, _I64, _I, _P],
    "udm_diffusion_loss": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _F, _F, _F, _P],
    "udm_timestep_embedding": [_P, _P, _I64, _I64, _P],
    "udm_silu_fwd": [_P, _P, _I64, _P],
    "udm_silu_bwd": [_P, _P, _P, _I64, _P],
}
EXTRA_SYMBOLS = ["udm_last_error", "udm_abi_version"]
ABI_VERSION = 3   # the UDM_ABI_VERSION of include/unidisc_hip.h that PROTOTYPES was written for (bumped whenever a signature changes)

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


class NotApplicable(RuntimeError):
    """An entry point of SOFT_RC3 answered rc = 3: "these shapes are not mine, nothing was launched" - the caller issues the plain calls instead."""


SOFT_RC3 = {"udm_gemm_tn_pair_bf16", "udm_gemm_tn_multi_bf16"}


def build(verbose: bool = False) -> str:
    """Compile EVERY HIP source of the product for gfx950 in-tree (``make`` in ``csrc/``).  Cross-compiles without a GPU."""
    out = subprocess.run(["make", "-C", CSRC, "-j4", "all"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode != 0:
        raise RuntimeError("building libunidisc_hip.so failed")
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C unidisc_amd/csrc`).  unidisc_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.udm_abi_version.restype = c_int
    lib.udm_abi_version.argtypes = []
    have = lib.udm_abi_version()
    if have != ABI_VERSION:   # a stale build would be called with shifted arguments (silent UB): refuse it
        raise HipLibraryMissing(f"{LIB_PATH} reports C-ABI version {have}, these bindings are written for {ABI_VERSION}: rebuild it (`make -C unidisc_amd/csrc`)")
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    lib.udm_last_error.restype = ctypes.c_char_p
    lib.udm_last_error.argtypes = []
    lib.udm_abi_version.restype = c_int
    lib.udm_abi_version.argtypes = []
    _lib = lib
    return lib


def call(name: str, *args):
    """Invoke an entry point; non-zero return raises RuntimeError(udm_last_error())."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc == 3 and name in SOFT_RC3:
        raise NotApplicable(name)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.udm_last_error().decode(errors='replace')}")
