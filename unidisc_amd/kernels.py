"""Typed Python wrappers over the C ABI (``include/unidisc_hip.h``).

PyTorch supplies device memory and the stream only; every function here enqueues hand-written HIP
kernels on ``torch.cuda.current_stream()``.  Tensors must live on a GPU — there is no CPU path.
"""
from __future__ import annotations

from typing import Optional

import os

import math

import torch

from . import _lib

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_DGELU = 0, 1, 2, 3
NORM_RMS, NORM_LN = 0, 1
BF16, F32 = torch.bfloat16, torch.float32


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("unidisc_amd kernels need GPU tensors (no CPU fallback); got a tensor on " + str(t.device))
    return t.data_ptr()


def require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("unidisc_amd.DIT runs on MI355X only: inputs must be GPU tensors (there is no CPU fallback)")
    _lib.load()  # fail loudly here if the HIP extension is missing


def _s():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _chk_packed_f32(t, name, numel=None):
    """a raw pointer goes into a device job table: the kernel reads `numel` packed fp32 values from it, whatever the tensor really is"""
    if t is None:
        return
    if t.dtype != F32 or not t.is_contiguous() or (numel is not None and t.numel() != numel):
        raise TypeError(f"{name}: expected a contiguous float32 tensor of {numel if numel is not None else 'any'} elements, got {t.dtype} {tuple(t.shape)} "
                        f"stride {t.stride()}")


_SCRATCH = {}


def _scratch(n: int, device) -> torch.Tensor:
    """Per-device fp32 scratch for two-phase column reductions (stream-ordered reuse: every user runs on the current stream)."""
    key = (device.type, device.index)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, dtype=F32, device=device)
        _SCRATCH[key] = buf
    return buf


# CUs the GEMMs may plan for (0 = all 256): set by gemm_set_cus / UDM_GEMM_CUS when a collective's kernels hold CUs during the backward (DDP).  A GEMM of
# exactly one round of one-workgroup tiles (every wgrad / dgrad and the single-round forward GEMMs of the 1.4 B step: 256 tiles) otherwise runs TWO rounds as soon as
# one CU is taken (+40 % on the step with 8 CUs held): `_row_split` cuts it into the whole tile rows that still fit one round and the leftover rows,
# which go through a split-K launch (or small tiles) and cost a fraction of a round.
_CUS = [int(os.environ.get("UDM_GEMM_CUS", "0") or 0) // 8 * 8]


def _row_split(rows, cols, tile_rows_options):
    """rows of the main part (a multiple of the chosen tile height), or None: `rows` x `cols` outputs in (tile height) x 256 tiles, whole tiles only."""
    G = _CUS[0]
    if not G or G >= 256 or cols % 256:
        return None
    best = None
    for bm in tile_rows_options:
        if rows % bm == 0:
            tiles = (rows // bm) * (cols // 256)
            cost = -(-tiles // 256) * bm
            if best is None or cost < best[0]:
                best = (cost, bm, tiles)
    if best is None:
        return None
    _, bm, tiles = best
    if not (G < tiles <= 256):
        return None
    main = (G // (cols // 256)) * bm
    return main if 0 < main < rows else None


def norm_id(norm_type: str) -> int:
    return NORM_RMS if norm_type == "rms" else NORM_LN


def norm_eps(norm_type: str) -> float:
    return 1e-6 if norm_type == "rms" else 1e-5


# ------------------------------------------------------------------------------------------------ GEMM
def gemm_nt(a, b, out=None, *, out_dtype=BF16, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, epilogue=EPI_NONE, bias=None, aux=None,
            ldaux=None, beta=0.0):
    """out[M,N] (+)= a[M,K] @ b[N,K]^T.  a, b bf16 2-D (row stride = stride(0)); out bf16 or fp32."""
    _chk(a, BF16, "gemm_nt a"), _chk(b, BF16, "gemm_nt b")
    M = a.shape[0] if M is None else M
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    ldc = out.stride(0) if ldc is None else ldc
    if aux is not None and ldaux is None:
        ldaux = aux.stride(0)
    if _CUS[0] and epilogue in (EPI_NONE, EPI_BIAS) and out.dtype == BF16 and beta == 0.0 and K % 64 == 0 and K >= 128:
        main = _row_split(M, N, (320, 256, 192))   # a single round of whole tiles that no longer fits the CUs a collective leaves: rows that fit + the rest on small tiles
        if main is not None:
            _lib.call("udm_gemm_nt_bf16", _p(a), _p(b), _p(out), main, N, K, lda, ldb, ldc, 0, epilogue, _p(bias), None, 0, 0.0, _s())
            _lib.call("udm_gemm_nt_bf16", _p(a[main:]), _p(b), _p(out[main:]), M - main, N, K, lda, ldb, ldc, 0, epilogue, _p(bias), None, 0, 0.0, _s())
            return out
    _lib.call("udm_gemm_nt_bf16", _p(a), _p(b), _p(out), M, N, K, lda, ldb, ldc, 1 if out.dtype == F32 else 0, epilogue, _p(bias), _p(aux),
              ldaux or 0, float(beta), _s())
    return out


def gemm_nn_ok(M, N, K):
    """Does the dgrad-from-W form (gemm_nn) cover this shape?  (whole tiles of the one-wave-per-SIMD kernel)"""
    return bool(_lib.load().udm_gemm_nn_ok(int(M), int(N), int(K)))


def gemm_nn(a, b, out=None, *, N=None):
    """out[M,N] (bf16) = a[M,K] @ b[K,N]; a bf16 row-major (K contiguous), b bf16 row-major [K rows, N contiguous] - the dgrad dX = dY W from the
    forward's W shadow [out, in].  Shapes must pass gemm_nn_ok."""
    _chk(a, BF16, "gemm_nn a"), _chk(b, BF16, "gemm_nn b")
    M, Kd = a.shape
    N = b.shape[1] if N is None else N
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    main = _row_split(M, N, (320, 256, 192))
    if main is not None and gemm_nn_ok(main, N, Kd) and gemm_nn_ok(M - main, N, Kd):
        _lib.call("udm_gemm_nn_bf16", _p(a), _p(b), _p(out), main, N, Kd, a.stride(0), b.stride(0), out.stride(0), _s())
        gemm_nn_splitk(a[main:], b, out[main:], N=N)
        return out
    _lib.call("udm_gemm_nn_bf16", _p(a), _p(b), _p(out), M, N, Kd, a.stride(0), b.stride(0), out.stride(0), _s())
    return out


def gemm_nn_splitk(a, b, out, *, N=None):
    """gemm_nn for few output tiles (the leftover tile rows of a row-split dgrad): K split across the available CUs through an fp32 workspace."""
    _chk(a, BF16, "gemm_nn_splitk a"), _chk(b, BF16, "gemm_nn_splitk b")
    M, Kd = a.shape
    N = b.shape[1] if N is None else N
    ws = _scratch(32 * M * N, a.device) if M * N <= (1 << 22) else None   # (at most 32 slices; big problems do not split)
    _lib.call("udm_gemm_nn_splitk_bf16", _p(a), _p(b), _p(out), M, N, Kd, a.stride(0), b.stride(0), out.stride(0), _p(ws), ws.numel() if ws is not None else 0, _s())
    return out


def gemm_tn(a, b, out, *, M=None, N=None, beta=0.0):
    """out[M,N] (fp32) = beta*out + a[K,M]^T @ b[K,N]; a, b bf16 row-major with K (rows) a multiple of 64."""
    _chk(a, BF16, "gemm_tn a"), _chk(b, BF16, "gemm_tn b"), _chk(out, F32, "gemm_tn out")
    K = a.shape[0]
    M = a.shape[1] if M is None else M
    N = b.shape[1] if N is None else N
    main = _row_split(M, N, (256, 192)) if out.stride(0) == N else None   # (the tile heights of the K-major one-wave-per-SIMD kernel, its cost rule)
    if main is not None:
        _lib.call("udm_gemm_tn_bf16", _p(a), _p(b), _p(out), main, N, K, a.stride(0), b.stride(0), out.stride(0), float(beta), _s())
        gemm_tn_splitk(a[:, main:], b, out[main:], M=M - main, N=N, beta=beta)
        return out
    _lib.call("udm_gemm_tn_bf16", _p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), float(beta), _s())
    return out


def gemm_tn_wants_splitk(M, N, K=None):
    """Few output tiles (<= half the chip) of a weight-gradient GEMM over the long row contraction: split K (the rule DIT._wgrad dispatches on)."""
    return ((M + 255) // 256) * ((N + 255) // 256) <= 128 and M * N >= 1 << 16


def gemm_tn_pair_ok(M0, M1, N, K):
    """Can two wgrads share one launch of 256 x 256 tiles (gemm_tn_pair)?"""
    if _CUS[0] and _CUS[0] < 256 and (M0 + M1) // 256 * (N // 256) > _CUS[0]:   # CUs are held: the shared grid would no longer fit one round (the two launches each plan for what is left)
        return False
    return M0 > 0 and M1 > 0 and M0 % 256 == 0 and M1 % 256 == 0 and N % 256 == 0 and K % 64 == 0 and K >= 128


def gemm_tn_pair(a0, b0, out0, a1, b1, out1, *, beta=0.0):
    """out0[M0,N] = beta*out0 + a0[K,M0]^T b0[K,N] and out1[M1,N] = beta*out1 + a1[K,M1]^T b1[K,N] (fp32) in ONE launch: the 256 x 256 tiles of both wgrads
    share a grid (qkv + out-proj weight gradients of a DiT block: 192 + 64 tiles = the 256 CUs exactly once).  Shapes must pass gemm_tn_pair_ok."""
    for t, n in ((a0, "a0"), (b0, "b0"), (a1, "a1"), (b1, "b1")):
        _chk(t, BF16, "gemm_tn_pair " + n)
    _chk(out0, F32, "gemm_tn_pair out0"), _chk(out1, F32, "gemm_tn_pair out1")
    K = a0.shape[0]
    M0, N = out0.shape
    M1 = out1.shape[0]
    if a1.shape[0] != K or b0.shape[0] != K or b1.shape[0] != K or out1.shape[1] != N or not gemm_tn_pair_ok(M0, M1, N, K):
        raise ValueError("gemm_tn_pair: the two problems need the same N and K, with M0, M1, N multiples of 256 and K of 64")
    tiles = ((M0 + M1) // 256) * (N // 256)
    sk = max(1, min(256 // tiles, (K // 64) // 8, 32))   # the rule of udm_gemm_tn_pair_bf16 (sizes the workspace): few tiles over a long K are split in K
    ws = _scratch(sk * (M0 + M1) * N, a0.device) if sk > 1 else None
    try:   # (through _lib.call like every launch: the bench's per-entry-point timer hooks it)
        _lib.call("udm_gemm_tn_pair_bf16", _p(a0), _p(b0), _p(out0), M0, a0.stride(0), b0.stride(0), out0.stride(0), _p(a1), _p(b1), _p(out1), M1, a1.stride(0),
                  b1.stride(0), out1.stride(0), N, K, float(beta), _p(ws), ws.numel() if ws is not None else 0, _s())
    except _lib.NotApplicable:
        # rc = 3 (one-wave-per-SIMD kernels switched off, or a pointer off its 16-byte alignment): nothing was launched, issue the two plain problems
        for a, b, out in ((a0, b0, out0), (a1, b1, out1)):
            (gemm_tn_splitk if gemm_tn_wants_splitk(out.shape[0], N, K) else gemm_tn)(a, b, out, beta=beta)


def gemm_tn_multi(problems, *, beta=0.0):
    """[(a_i [K, M_i], b_i [K, N_i], out_i [M_i, N_i] fp32 contiguous)] -> out_i = beta * out_i + a_i^T b_i for up to four problems over the SAME K in ONE split-K launch
    + ONE reduce pass (the few-tile weight gradients of a small DiT block).  Returns False (nothing launched) when the shapes do not qualify."""
    import ctypes

    n = len(problems)
    if not (1 <= n <= 4):
        return False
    K = problems[0][0].shape[0]
    tiles = 0
    for a, b, out in problems:
        _chk(a, BF16, "gemm_tn_multi a"), _chk(b, BF16, "gemm_tn_multi b"), _chk(out, F32, "gemm_tn_multi out")
        M, N = out.shape
        if a.shape[0] != K or b.shape[0] != K or M % 256 or N % 256 or out.stride(0) != N or a.stride(1) != 1 or b.stride(1) != 1:
            return False
        tiles += (M // 256) * (N // 256)
    cus = _CUS[0] or 256
    sk = min(cus // max(tiles, 1), (K // 64) // 8, 32)
    if tiles > 128 or sk < 2 or K % 64:
        return False
    area = sum(o.numel() for _, _, o in problems)
    ws = _scratch(sk * area, problems[0][0].device)
    P, I64 = ctypes.c_void_p * n, ctypes.c_int64 * n
    try:
        _lib.call("udm_gemm_tn_multi_bf16", n, P(*[a.data_ptr() for a, _, _ in problems]), P(*[b.data_ptr() for _, b, _ in problems]),
                  P(*[o.data_ptr() for _, _, o in problems]), I64(*[o.shape[0] for _, _, o in problems]), I64(*[o.shape[1] for _, _, o in problems]),
                  I64(*[a.stride(0) for a, _, _ in problems]), I64(*[b.stride(0) for _, b, _ in problems]), K, float(beta), _p(ws), ws.numel(), _s())
    except _lib.NotApplicable:
        return False
    return True


def gemm_tn_splitk(a, b, out, *, M=None, N=None, beta=0.0):
    """gemm_tn for few output tiles over a long contraction: K split across CUs through an fp32 workspace (no atomics)."""
    _chk(a, BF16, "gemm_tn_splitk a"), _chk(b, BF16, "gemm_tn_splitk b"), _chk(out, F32, "gemm_tn_splitk out")
    K = a.shape[0]
    M = a.shape[1] if M is None else M
    N = b.shape[1] if N is None else N
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    sk = max(1, min(256 // tiles, (K // 64) // 8, 32))   # the rule of udm_gemm_tn_splitk_bf16 (sizes the workspace)
    ws = _scratch(sk * M * N, a.device) if sk > 1 else None
    _lib.call("udm_gemm_tn_splitk_bf16", _p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), float(beta), _p(ws),
              ws.numel() if ws is not None else 0, _s())
    return out


def gemm_nt_splitk(a, b, out=None, *, N=None):
    """out[M,N] (bf16) = a[M,K] b[N,K]^T for few output tiles over a long contraction: K split across CUs through an fp32 workspace."""
    _chk(a, BF16, "gemm_nt_splitk a"), _chk(b, BF16, "gemm_nt_splitk b")
    M, Kd = a.shape
    N = b.shape[0] if N is None else N
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    tiles = ((M + 319) // 320) * ((N + 255) // 256)
    sk = max(1, min(256 // tiles, (Kd // 64) // 8, 32))   # the rule of udm_gemm_nt_splitk_bf16 (sizes the workspace)
    ws = _scratch(sk * M * N, a.device) if sk > 1 else None
    _lib.call("udm_gemm_nt_splitk_bf16", _p(a), _p(b), _p(out), M, N, Kd, a.stride(0), b.stride(0), out.stride(0), _p(ws),
              ws.numel() if ws is not None else 0, _s())
    return out


def ddpm_sample_rows(logits, V, Vt, mask_id, *, t=None, s=None, modality=None, restrict=False, u=None, seed=0, greedy=False, logits_u=None, w=None):
    """Sampled (or, greedy, arg-max) token per row of `logits` [rows, ld] bf16: one reverse-diffusion update of [MASK] rows.
    `logits_u` (same shape / stride) + `w` fp32 [rows]: classifier-free guidance, z = (1 + w) logits - w logits_u, mixed inside the kernel."""
    _chk(logits, BF16, "ddpm_sample_rows logits")
    M = logits.shape[0]
    out = torch.empty(M, dtype=torch.int64, device=logits.device)
    if (logits_u is None) != (w is None):
        raise ValueError("ddpm_sample_rows: guidance needs both logits_u and w")
    if logits_u is not None:
        _chk(logits_u, BF16, "ddpm_sample_rows logits_u"), _chk(w, F32, "ddpm_sample_rows w")
        if logits_u.shape != logits.shape or logits_u.stride(0) != logits.stride(0) or w.numel() != M:
            raise ValueError("ddpm_sample_rows: logits_u must match logits (shape and row stride) and w must have one weight per row")
    _lib.call("udm_ddpm_sample_rows_cfg", _p(logits), _p(logits_u), _p(w), logits.stride(0), _p(modality), _p(t), _p(s), _p(u),
              u.stride(0) if u is not None else 0, int(seed), _p(out), M, V, Vt, mask_id, 1 if restrict else 0, 1 if greedy else 0, _s())
    return out


def assemble_joint_tokens(txt, txt_mask, img, Vt, idx=None):
    """(input_ids int64, attention_mask bool, modality int64), each [B, Lt + Li], from dataset fields txt int32 [n, Lt], txt_mask bool [n, Lt] (or None),
    img int16 [n, Li]; idx int64 [B] selects rows (None: all n rows in order).  `update_batch` token-dataset branch in one pass."""
    if txt.dtype != torch.int32 or img.dtype != torch.int16 or txt.dim() != 2 or img.dim() != 2 or not (txt.is_contiguous() and img.is_contiguous()):
        raise TypeError("assemble_joint_tokens: txt must be contiguous int32 [n, Lt], img contiguous int16 [n, Li]")
    if txt.shape[0] != img.shape[0] or not (txt.is_cuda and img.is_cuda):
        raise ValueError("assemble_joint_tokens: txt / img must be device tensors with the same number of rows")
    if txt_mask is not None and (txt_mask.dtype != torch.bool or txt_mask.shape != txt.shape or not txt_mask.is_contiguous() or not txt_mask.is_cuda):
        raise TypeError("assemble_joint_tokens: txt_mask must be a contiguous bool device tensor shaped like txt")
    if idx is not None and (idx.dtype != torch.int64 or idx.dim() != 1 or not idx.is_cuda):
        raise TypeError("assemble_joint_tokens: idx must be a 1-D int64 device tensor")
    B = txt.shape[0] if idx is None else idx.shape[0]
    Lt, Li = txt.shape[1], img.shape[1]
    ids = torch.empty(B, Lt + Li, dtype=torch.int64, device=txt.device)
    mask = torch.empty(B, Lt + Li, dtype=torch.bool, device=txt.device)
    modality = torch.empty(B, Lt + Li, dtype=torch.int64, device=txt.device)
    _lib.call("udm_assemble_joint_tokens", _p(txt), _p(txt_mask), _p(img), _p(idx), B, Lt, Li, int(Vt), _p(ids), _p(mask), _p(modality), _s())
    return ids, mask, modality


def interleaved_rope(modality, sid, img_cos, img_sin, sizes, txt_cos, txt_sin):
    """Rotary rows and image-count indices of packed rows (udm_interleaved_rope): modality / sid int64 [B, L] -> (cos, sin fp32 [B, L, half], count_idx int64 [B, L])."""
    import ctypes

    _chk(modality, torch.int64, "interleaved_rope modality"), _chk(sid, torch.int64, "interleaved_rope sid")
    for t, nm in ((img_cos, "img_cos"), (img_sin, "img_sin"), (txt_cos, "txt_cos"), (txt_sin, "txt_sin")):
        _chk(t, F32, f"interleaved_rope {nm}")
    B, L = modality.shape
    half = img_cos.shape[1]
    dev = modality.device
    cos = torch.empty((B, L, half), dtype=F32, device=dev)
    sin = torch.empty((B, L, half), dtype=F32, device=dev)
    cidx = torch.empty((B, L), dtype=torch.int64, device=dev)
    scratch = torch.empty(B * 5 * L, dtype=torch.int32, device=dev)
    hs = (ctypes.c_int32 * len(sizes))(*[int(n) for n in sizes])
    _lib.call("udm_interleaved_rope", _p(modality.contiguous()), _p(sid.contiguous()), _p(img_cos.contiguous()), _p(img_sin.contiguous()), hs, len(sizes),
              _p(txt_cos.contiguous()), _p(txt_sin.contiguous()), txt_cos.shape[0], B, L, half, _p(cos), _p(sin), _p(cidx), _p(scratch), _s())
    return cos, sin, cidx


def interleaved_block_lottery(modality, sid, r, mask_prob):
    """Whole-block masking of packed rows after its draws (udm_interleaved_block_lottery): -> (accum bool [B, L], rows_hit bool [B], n_cand int64 [1] on the device)."""
    _chk(modality, torch.int64, "interleaved_block_lottery modality"), _chk(sid, torch.int64, "interleaved_block_lottery sid")
    B, L = modality.shape
    dev = modality.device
    accum = torch.empty((B, L), dtype=torch.bool, device=dev)
    rows_hit = torch.empty((B,), dtype=torch.bool, device=dev)
    n_cand = torch.empty(1, dtype=torch.int64, device=dev)
    scratch = torch.empty(B * 4 * L, dtype=torch.int32, device=dev)
    row_cands = torch.empty(B, dtype=torch.int32, device=dev)
    r = r.reshape(-1).float().contiguous()
    _lib.call("udm_interleaved_block_lottery", _p(modality.contiguous()), _p(sid.contiguous()), _p(r), r.numel(), torch.tensor(mask_prob, dtype=F32).item(), B, L,
              _p(accum), _p(rows_hit), _p(n_cand), _p(scratch), _p(row_cands), _s())
    return accum, rows_hit, n_cand


def rowgroup_sum(x, group, out):
    """out[g] += sum of the rows of x [M, d] fp32 whose group index (int64 [M]) is g; rows with an index outside [0, out.shape[0]) are skipped."""
    _chk(x, F32, "rowgroup_sum x"), _chk(group, torch.int64, "rowgroup_sum group"), _chk(out, F32, "rowgroup_sum out")
    _lib.call("udm_rowgroup_sum_f32", _p(x.contiguous()), _p(group.contiguous()), _p(out), x.shape[0], x.shape[1], out.shape[0], _s())
    return out


def sample_t_noise(u, *, antithetic, sampling_eps, noise_eps):
    """(t, sigma, dsigma, move_chance), fp32 [n] each, from the uniform draws u [n]: `_sample_t` + the log-linear schedule in one launch, rounded like the statements."""
    _chk(u, F32, "sample_t_noise u")
    n = u.numel()
    out = torch.empty((4, n), dtype=F32, device=u.device)
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
    _lib.call("udm_sample_t_noise", _p(u.contiguous()), n, 1 if antithetic else 0, f32(1 - sampling_eps), f32(sampling_eps), f32(1 - noise_eps), _p(out[0]), _p(out[1]),
              _p(out[2]), _p(out[3]), _s())
    return out[0], out[1], out[2], out[3]


def qxt_absorbing(x, r_move, move_chance, mask_id, *, r_txt=None, r_img=None, p_txt=0.0, p_img=0.0, modality_mask=None):
    """q_xt after its random draws, multimodal non-interleaved batches: (xt int64 [B, L], move_indices bool [B, L], should_mask_txt, should_mask_img, ignore: bool [B, 1]
    or None without whole-modality draws).  x int64 [B, L]; r_move fp32 [B, L]; move_chance fp32 [B] / [B, 1]; r_txt / r_img fp32 [B, 1]; modality_mask bool [B, L, 2]."""
    B, L = x.shape
    _chk(r_move, F32, "qxt_absorbing r_move"), _chk(move_chance, F32, "qxt_absorbing move_chance")
    if x.dtype != torch.int64 or not x.is_contiguous() or not r_move.is_contiguous() or move_chance.numel() != B:
        raise TypeError("qxt_absorbing: x must be contiguous int64 [B, L], r_move contiguous fp32 [B, L], move_chance one value per row")
    xt = torch.empty_like(x)
    move = torch.empty((B, L), dtype=torch.bool, device=x.device)
    rows = None
    if r_txt is not None or r_img is not None:
        if modality_mask is None or modality_mask.dtype != torch.bool or not modality_mask.is_contiguous() or modality_mask.shape != (B, L, 2):
            raise TypeError("qxt_absorbing: whole-modality masking needs a contiguous bool modality_mask [B, L, 2]")
        rows = torch.empty((3, B, 1), dtype=torch.bool, device=x.device)
    thr = lambda p: float(torch.tensor(p, dtype=torch.float32))   # `tensor < python_float` compares with the scalar rounded to the tensor's dtype
    _lib.call("udm_qxt_absorbing", _p(x), _p(r_move), _p(move_chance.contiguous()), _p(r_txt.contiguous() if r_txt is not None else None),
              _p(r_img.contiguous() if r_img is not None else None), thr(p_txt), thr(p_img), _p(modality_mask), B, L, int(mask_id), _p(xt), _p(move),
              _p(rows[0]) if rows is not None else None, _p(rows[1]) if rows is not None else None, _p(rows[2]) if rows is not None else None, _s())
    if rows is None:
        return xt, move, None, None, None
    return xt, move, rows[0], rows[1], rows[2]


def categorical_sample_rows(logits, V, Vt, mask_id, *, modality=None, restrict=False, u=None, seed=0, given=None, logits_u=None, w=None):
    """(token, log p(token)) per row of `logits` [rows, ld] bf16 under the SUBS distribution: the `maskgit` predictor's multinomial draw + confidence.
    `given` int64 [rows]: take these tokens instead of drawing (replay)."""
    _chk(logits, BF16, "categorical_sample_rows logits")
    M = logits.shape[0]
    if (logits_u is None) != (w is None):
        raise ValueError("categorical_sample_rows: guidance needs both logits_u and w")
    if logits_u is not None and (logits_u.shape != logits.shape or logits_u.stride(0) != logits.stride(0) or w.numel() != M):
        raise ValueError("categorical_sample_rows: logits_u must match logits and w must have one weight per row")
    if given is not None and (given.dtype != torch.int64 or given.numel() != M):
        raise TypeError("categorical_sample_rows: given must be int64 [rows]")
    tok = torch.empty(M, dtype=torch.int64, device=logits.device)
    logp = torch.empty(M, dtype=torch.float32, device=logits.device)
    _lib.call("udm_categorical_sample_rows", _p(logits), _p(logits_u), _p(w), logits.stride(0), _p(modality), _p(u), u.stride(0) if u is not None else 0,
              int(seed), _p(given), _p(tok), _p(logp), M, V, Vt, mask_id, 1 if restrict else 0, _s())
    return tok, logp


def sumsq(x, out):
    """out[0] = sum(x**2) (fp32, 1-D contiguous x); two-phase reduction through the scratch buffer."""
    _chk(x, F32, "sumsq x"), _chk(out, F32, "sumsq out")
    ws = _scratch(1024, x.device)
    _lib.call("udm_sumsq_f32", _p(x), x.numel(), _p(out), _p(ws), ws.numel(), _s())
    return out


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq=None, max_grad_norm=None, ema=None, ema_decay=0.0):
    """In-place AdamW update of a flat fp32 tensor (torch.optim.AdamW arithmetic); optional device-side global-norm clipping; optional parameter
    EMA (ema <- ema - (1 - ema_decay) (ema - p_new)) in the same pass."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, F32, f"adamw_step {n}")
    if ema is not None:
        _chk(ema, F32, "adamw_step ema")
    _lib.call("udm_adamw_step_ema", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
              _p(grad_norm_sq), float(max_grad_norm or 0.0), _p(ema), float(ema_decay), _s())


def adamw_step_shadow(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq, max_grad_norm, w16, w16t, ema=None, ema_decay=0.0):
    """AdamW update of a 2-D GEMM weight [R, C] that also writes its bf16 shadow w16 [>=R, C] and transposed shadow w16t [C, >=R] (+ EMA)."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, F32, f"adamw_step_shadow {n}")
    if ema is not None:
        _chk(ema, F32, "adamw_step_shadow ema")
    R, C = p.shape
    _lib.call("udm_adamw_step_shadow_ema", _p(p), _p(g), _p(m), _p(v), R, C, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
              _p(grad_norm_sq), float(max_grad_norm or 0.0), _p(w16), w16.stride(0) if w16 is not None else 0, _p(w16t),
              w16t.stride(0) if w16t is not None else 0, _p(ema), float(ema_decay), _s())


SMALL_BATCH_LINEAR_MAX_B, SMALL_BATCH_LINEAR_MAX_IN = 64, 128


def small_batch_linear_bwd_tiles(out):
    """number of partial tiles small_batch_linear_bwd(..., dx_parts=...) writes for `out` output features"""
    return int(_lib.load().udm_small_batch_linear_bwd_blocks(int(out)))


def small_batch_linear_bwd(dy, x, w16, dw, db, dx=None, dx_parts=None):
    """nn.Linear backward for a small batch in one launch (adaLN_modulation): dy fp32 [B, out] (rounded to bf16 on load), x bf16 [B, in], w16 bf16 [out, in]:
    dw[out, in] = dy^T x (overwritten), db[out] += colsum(dy) (None: skipped); the input gradient dy w16 is added into dx[B, in] (atomics) or, with `dx_parts`
    fp32 [small_batch_linear_bwd_tiles(out), B, in], written as partial tiles for the caller to sum.  B <= 64, in <= 128 and a multiple of 8; x / w16 rows 16-byte aligned."""
    _chk(dy, F32, "small_batch_linear_bwd dy"), _chk(x, BF16, "small_batch_linear_bwd x"), _chk(w16, BF16, "small_batch_linear_bwd w16")
    _chk(dw, F32, "small_batch_linear_bwd dw")
    B, out = dy.shape
    inp = x.shape[1]
    if not (dw.is_contiguous() and tuple(dw.shape) == (out, inp) and w16.shape[0] >= out and w16.shape[1] == inp and x.shape[0] == B):
        raise ValueError("small_batch_linear_bwd: inconsistent shapes")
    if dx_parts is not None:
        _chk(dx_parts, F32, "small_batch_linear_bwd dx_parts")
        if not (dx_parts.is_contiguous() and tuple(dx_parts.shape) == (small_batch_linear_bwd_tiles(out), B, inp)):
            raise ValueError("small_batch_linear_bwd: dx_parts must be contiguous [small_batch_linear_bwd_tiles(out), B, in]")
    else:
        _chk(dx, F32, "small_batch_linear_bwd dx")
        if tuple(dx.shape) != (B, inp):
            raise ValueError("small_batch_linear_bwd: dx must be [B, in]")
    _lib.call("udm_small_batch_linear_bwd", _p(dy), dy.stride(0), _p(x), x.stride(0), _p(w16), w16.stride(0), _p(dw), _p(db), _p(dx), dx.stride(0) if dx is not None else 0,
              _p(dx_parts), B, out, inp, _s())


def colsum(x, out):
    """out[c] += sum_r x[r, c] (bias gradient) without writing a transpose."""
    _chk(x, BF16, "colsum")
    _lib.call("udm_transpose_bf16", _p(x), None, x.shape[0], x.shape[1], x.stride(0), 0, _p(out), _s())
    return out


def debug_set(key: str, value: int):
    """Diagnostics / A-B switches of the library (udm_debug_set; keys in include/unidisc_hip.h)."""
    _lib.call("udm_debug_set", key.encode(), int(value))


def gemm_set_persist(enable: int):
    debug_set("gemm_persist", int(enable))


def gemm_set_quad(mode: int):
    """Diagnostics: one-wave-per-SIMD GEMM kernels (gemm_quad.hip): 0 = off, 1 = auto, 2 = wherever the shape fits."""
    debug_set("gemm_quad", int(mode))


def gemm_set_cus(cus: int):
    """The GEMMs plan for `cus` CUs (multiple of 8; 0 = all 256): persistent grids are capped, split-K cuts for `cus` blocks, single-round GEMMs of more tiles are
    split by rows (`_row_split`) - leaves CUs to RCCL's kernels in data-parallel runs."""
    _lib.call("udm_gemm_set_cus", int(cus))
    _CUS[0] = int(cus)


def gemm_set_tile(tile: int):
    """Diagnostics: force the GEMM tile family (-1 auto, 0 small kernel, 192/256/320)."""
    debug_set("gemm_tile", tile)


def transpose(x, out=None, colsum=None, R=None, C=None):
    """out[C,R] = x[R,C]^T (bf16); colsum[c] += sum_r x[r,c] when given."""
    _chk(x, BF16, "transpose")
    R = x.shape[0] if R is None else R
    C = x.shape[1] if C is None else C
    if out is None:
        out = torch.empty((C, R), dtype=BF16, device=x.device)
    _lib.call("udm_transpose_bf16", _p(x), _p(out), R, C, x.stride(0), out.stride(0), _p(colsum), _s())
    return out


def cast_transpose(w, out, out_t):
    """fp32 [R,C] -> bf16 [R,C] (out) and bf16 [C,R] (out_t); either may be None."""
    _chk(w, F32, "cast_transpose")
    R, C = w.shape
    _lib.call("udm_cast_transpose_f32_bf16", _p(w), _p(out), _p(out_t), R, C, w.stride(0), out.stride(0) if out is not None else 0,
              out_t.stride(0) if out_t is not None else 0, _s())


def cast_transpose_jobs(items, device):
    """Device job table for `cast_transpose_multi`: items = [(w fp32 [R, C], out bf16 [>=R, C] or None, out_t bf16 [C, >=R] or None), ...]."""
    import struct

    buf, tile0 = bytearray(), 0
    for w, out, out_t in items:
        _chk(w, F32, "cast_transpose_jobs")
        R, C = w.shape
        tiles_c = (C + 63) // 64
        buf += struct.pack("<QQQqqqiiii", _p(w), _p(out) or 0, _p(out_t) or 0, w.stride(0), out.stride(0) if out is not None else 0,
                           out_t.stride(0) if out_t is not None else 0, R, C, tile0, tiles_c)
        tile0 += ((R + 63) // 64) * tiles_c
    table = torch.frombuffer(buf, dtype=torch.uint8).to(device)   # (a bytearray is writable: no non-writable-buffer warning)
    return table, len(items), tile0


def adamw_jobs(items, device):
    """Device job table for `adamw_step_multi`: items = [(p, g, m, v, ema or None)] fp32 contiguous tensors of equal numel per item."""
    import struct

    buf, chunk0 = bytearray(), 0
    for p, g, m, v, e in items:
        n = p.numel()
        for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (e, "ema")):
            _chk_packed_f32(t, f"adamw_jobs {nm}", n)
        buf += struct.pack("<QQQQQqq", _p(p), _p(g), _p(m), _p(v), _p(e) or 0, n, chunk0)
        chunk0 += (n + 1023) // 1024
    return torch.frombuffer(buf, dtype=torch.uint8).to(device), len(items), chunk0


def adamw_step_multi(jobs, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq=None, max_grad_norm=None, ema_decay=0.0):
    """AdamW (+ clipping coefficient from the device scalar, + EMA) of every tensor of a job table (see `adamw_jobs`) in one launch."""
    table, n, chunks = jobs
    _lib.call("udm_adamw_step_multi", _p(table), n, chunks, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
              _p(grad_norm_sq) if grad_norm_sq is not None else None, float(max_grad_norm if max_grad_norm is not None else 0.0), float(ema_decay), _s())


def adamw_shadow_jobs(items, device):
    """Device job table for `adamw_step_shadow_multi`: items = [(p [R, C], g, m, v, ema or None, w16 or None, w16t or None)]."""
    import struct

    buf, tile0 = bytearray(), 0
    for p, g, m, v, e, w16, w16t in items:
        R, C = p.shape
        for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (e, "ema")):
            _chk_packed_f32(t, f"adamw_shadow_jobs {nm}", R * C)
        for t, nm, shp in ((w16, "w16", (R, C)), (w16t, "w16t", (C, R))):   # (shadows may be padded to whole GEMM tiles: at least the parameter's extent)
            if t is not None and (t.dtype != BF16 or t.dim() != 2 or t.shape[0] < shp[0] or t.shape[1] < shp[1] or t.stride(1) != 1):
                raise TypeError(f"adamw_shadow_jobs {nm}: expected bf16 of at least {shp} with unit inner stride, got {t.dtype} {tuple(t.shape)} stride {t.stride()}")
        tiles_c = (C + 63) // 64
        buf += struct.pack("<QQQQQQQqqiiii", _p(p), _p(g), _p(m), _p(v), _p(e) or 0, _p(w16) or 0, _p(w16t) or 0, w16.stride(0) if w16 is not None else 0,
                           w16t.stride(0) if w16t is not None else 0, R, C, tile0, tiles_c)
        tile0 += ((R + 63) // 64) * tiles_c
    return torch.frombuffer(buf, dtype=torch.uint8).to(device), len(items), tile0


def adamw_step_shadow_multi(jobs, lr, beta1, beta2, eps, weight_decay, step, grad_norm_sq=None, max_grad_norm=None, ema_decay=0.0):
    table, n, tiles = jobs
    _lib.call("udm_adamw_step_shadow_multi", _p(table), n, tiles, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
              _p(grad_norm_sq) if grad_norm_sq is not None else None, float(max_grad_norm if max_grad_norm is not None else 0.0), float(ema_decay), _s())


def cast_transpose_multi(jobs):
    """fp32 -> bf16 (and transposed bf16) for every matrix of a job table (see `cast_transpose_jobs`) in one launch."""
    table, n, tiles = jobs
    _lib.call("udm_cast_transpose_multi_f32_bf16", _p(table), n, tiles, _s())


def cast_f32_bf16(x, y, scale=1.0):
    _lib.call("udm_cast_f32_bf16", _p(x), _p(y), x.numel(), float(scale), _s())
    return y


def cast_bf16_f32(x, y, scale=1.0):
    _lib.call("udm_cast_bf16_f32", _p(x), _p(y), x.numel(), float(scale), _s())
    return y


# ------------------------------------------------------------------------------------------------ norms
def _mod_ptrs(mod, idx, d):
    """adaLN slices: `mod` is the bf16 (or fp32 gradient) [Bp, n*d] adaLN tensor, idx picks d-wide column chunks."""
    if mod is None:
        return [None] * len(idx), 0
    es = mod.element_size()
    return [mod.data_ptr() + es * k * d for k in idx], mod.stride(0)


def norm_fwd(x, w, norm_type, L, *, mod=None, mod_idx=(0, 1), modality=None, any_img=None):
    """y = modulate(norm(x) * w) -> bf16.  mod_idx = (shift chunk, scale chunk) of the adaLN output."""
    M, d = x.shape
    y = torch.empty((M, d), dtype=BF16, device=x.device)
    rstd = torch.empty(M, dtype=F32, device=x.device)
    mean = torch.empty(M, dtype=F32, device=x.device) if norm_type == NORM_LN else None
    (shift, scale), ms = _mod_ptrs(mod, mod_idx, d)
    _lib.call("udm_norm_fwd", _p(x), _p(y), _p(rstd), _p(mean), _p(w), shift, scale, ms, _p(modality) if mod is not None else None,
              _p(any_img) if mod is not None else None, M, d, L, norm_type, 1e-6 if norm_type == NORM_RMS else 1e-5, _s())
    return y, rstd, mean


def norm_bwd(dy, x, rstd, mean, w, norm_type, L, dx, dw, *, accumulate=True, mod=None, dmod=None, mod_idx=(0, 1), modality=None, any_img=None):
    """dx (+)= d norm / dx, dw += ..., dmod[:, shift/scale chunks] += ... (fp32 atomics)."""
    M, d = x.shape
    (shift, scale), ms = _mod_ptrs(mod, mod_idx, d)
    (dshift, dscale), _ = _mod_ptrs(dmod, mod_idx, d)
    ws = _scratch(1024 * d, x.device) if M >= 2048 else None
    _lib.call("udm_norm_bwd", _p(dy), _p(x), _p(rstd), _p(mean), _p(w), shift, scale, ms, _p(modality) if mod is not None else None,
              _p(any_img) if mod is not None else None, _p(dx), _p(dw), dshift, dscale, M, d, L, norm_type, 1 if accumulate else 0,
              _p(ws), ws.numel() if ws is not None else 0, _s())


def residual_fwd(x_in, branch, L, *, w_b=None, norm_type=NORM_RMS, mod=None, gate_idx=None, modality=None, p_drop=0.0, seed=0, next_w=None, next_mod=None,
                 next_mod_idx=(0, 1), next_modality=None, next_any_img=None):
    """x_out = x_in + gate * dropout(sandwich_norm(branch)).  gate = chunk gate_idx of `mod` (None: no gate).
    next_w: weight of the norm that consumes x_out next -- fused; returns (x_out, rstd, mean, (h, rstd_n, mean_n)).  next_mod: that norm is modulated by chunks
    next_mod_idx = (shift, scale) of this adaLN tensor (image rows only under next_modality / next_any_img, as norm_fwd)."""
    M, d = x_in.shape
    x_out = torch.empty_like(x_in)
    rstd = torch.empty(M, dtype=F32, device=x_in.device) if w_b is not None else None
    mean = torch.empty(M, dtype=F32, device=x_in.device) if (w_b is not None and norm_type == NORM_LN) else None
    (gate,), ms = _mod_ptrs(mod if gate_idx is not None else None, (gate_idx,), d)
    eps = 1e-6 if norm_type == NORM_RMS else 1e-5
    if next_w is None:
        _lib.call("udm_residual_fwd", _p(x_in), _p(branch), _p(x_out), _p(w_b), _p(rstd), _p(mean), gate, ms, _p(modality), M, d, L, norm_type,
                  eps, float(p_drop), int(seed), _s())
        return x_out, rstd, mean
    h = torch.empty((M, d), dtype=BF16, device=x_in.device)
    rstd_n = torch.empty(M, dtype=F32, device=x_in.device)
    mean_n = torch.empty(M, dtype=F32, device=x_in.device) if norm_type == NORM_LN else None
    if next_mod is not None:
        (n_shift, n_scale), n_ms = _mod_ptrs(next_mod, next_mod_idx, d)
        _lib.call("udm_residual_norm_fwd_ada", _p(x_in), _p(branch), _p(x_out), _p(w_b), _p(rstd), _p(mean), gate, ms, _p(modality), M, d, L, norm_type,
                  eps, float(p_drop), int(seed), _p(next_w), _p(h), _p(rstd_n), _p(mean_n), n_shift, n_scale, n_ms, _p(next_modality), _p(next_any_img), _s())
        return x_out, rstd, mean, (h, rstd_n, mean_n)
    _lib.call("udm_residual_norm_fwd", _p(x_in), _p(branch), _p(x_out), _p(w_b), _p(rstd), _p(mean), gate, ms, _p(modality), M, d, L, norm_type,
              eps, float(p_drop), int(seed), _p(next_w), _p(h), _p(rstd_n), _p(mean_n), _s())
    return x_out, rstd, mean, (h, rstd_n, mean_n)


def residual_bwd(dx, branch, L, *, w_b=None, rstd=None, mean=None, norm_type=NORM_RMS, mod=None, dmod=None, gate_idx=None, modality=None, dw_b=None,
                 p_drop=0.0, seed=0):
    M, d = dx.shape
    dbranch = torch.empty((M, d), dtype=BF16, device=dx.device)
    use = gate_idx is not None and mod is not None
    (gate,), ms = _mod_ptrs(mod if use else None, (gate_idx,), d)
    (dgate,), _ = _mod_ptrs(dmod if use else None, (gate_idx,), d)
    ws = _scratch(1536 * d, dx.device) if w_b is not None else None
    _lib.call("udm_residual_bwd", _p(dx), _p(branch), _p(dbranch), _p(w_b), _p(rstd), _p(mean), gate, ms, _p(modality), _p(dw_b), dgate, M, d, L,
              norm_type, float(p_drop), int(seed), _p(ws), ws.numel() if ws is not None else 0, _s())
    return dbranch


def norm_residual_bwd(dy, x, rstd, mean, w, norm_type, L, dx, dw, branch, *, accumulate=True, w_b=None, rstd_b=None, mean_b=None, dw_b=None, p_drop=0.0, seed=0,
                      dbias=None):
    """Unmodulated norm backward (dx (+)= ..., dw += ...) followed by the residual-branch backward on the updated dx (returns d branch, bf16; dw_b += ...;
    dbias += column sums of d branch when given).  One fused pass per row at d = 2048 / 4096 (block per row) and at d < 2048 (wave per row), the separate
    kernels otherwise."""
    M, d = x.shape
    if (d in (2048, 4096) or (d < 2048 and d % 8 == 0 and d >= 64)) and x.is_cuda:
        dbranch = torch.empty((M, d), dtype=BF16, device=x.device)
        ws = _scratch(min(M, 1536) * 3 * d, x.device)
        _lib.call("udm_norm_residual_bwd", _p(dy), _p(x), _p(rstd), _p(mean), _p(w), _p(dx), _p(dw), 1 if accumulate else 0, _p(branch), _p(dbranch), _p(w_b),
                  _p(rstd_b), _p(mean_b), _p(dw_b), _p(dbias), M, d, norm_type, float(p_drop), int(seed), _p(ws), ws.numel(), _s())
        return dbranch
    norm_bwd(dy, x, rstd, mean, w, norm_type, L, dx, dw, accumulate=accumulate)
    out = residual_bwd(dx, branch, L, w_b=w_b, rstd=rstd_b, mean=mean_b, norm_type=norm_type, dw_b=dw_b, p_drop=p_drop, seed=seed)
    if dbias is not None:
        colsum(out, dbias)
    return out


_ADA_GRID_ROWS = 768     # partial-sum rows of udm_norm_residual_bwd_ada's workspace (rowops.hip: grid = B * max(1, min(768 / B, L)))


def norm_residual_bwd_ada_ok(M, d, L):
    """Does the fused adaLN form (norm_residual_bwd_ada) cover this shape?  (the block-per-row kernel: d = 2048 / 4096, whole batch elements)"""
    return d in (2048, 4096) and L > 0 and M % L == 0 and M // L <= _ADA_GRID_ROWS   # (the launch is B x min(768 / B, L) blocks over a 768 x 6 x d workspace)


def norm_residual_bwd_ada(dy, x, rstd, mean, w, norm_type, L, dx, dw, branch, *, accumulate=True, w_b=None, rstd_b=None, mean_b=None, dw_b=None, p_drop=0.0, seed=0,
                          dbias=None, mod_n=None, dmod_n=None, mod_idx=(0, 1), modality=None, any_img=None, mod_r=None, dmod_r=None, gate_idx=None, modality_r=None):
    """norm_residual_bwd with adaLN-Zero: the norm modulated by chunks `mod_idx` = (shift, scale) of `mod_n` (gradients into the same chunks of `dmod_n`; image rows
    only under `modality` / `any_img` as in norm_bwd), the residual branch gated by chunk `gate_idx` of `mod_r` (gradient into `dmod_r`; gate and dropout on rows
    with modality_r == 1 only when given, as in residual_bwd).  Shapes must pass norm_residual_bwd_ada_ok."""
    M, d = x.shape
    (shift, scale), ms_n = _mod_ptrs(mod_n, mod_idx, d)
    (dshift, dscale), _ = _mod_ptrs(dmod_n, mod_idx, d)
    use_gate = gate_idx is not None and mod_r is not None
    (gate,), ms_r = _mod_ptrs(mod_r if use_gate else None, (gate_idx,), d)
    (dgate,), _ = _mod_ptrs(dmod_r if use_gate else None, (gate_idx,), d)
    if mod_n is not None and use_gate and ms_n != ms_r:
        raise ValueError("norm_residual_bwd_ada: the two adaLN tensors must share their row stride")
    for m_, dm_ in ((mod_n, dmod_n), (mod_r if use_gate else None, dmod_r)):
        if m_ is not None and (dm_ is None or dm_.stride(0) != m_.stride(0)):
            raise ValueError("norm_residual_bwd_ada: an adaLN tensor and its gradient must share their row stride")
    dbranch = torch.empty((M, d), dtype=BF16, device=x.device)
    ws = _scratch(_ADA_GRID_ROWS * 6 * d, x.device)
    _lib.call("udm_norm_residual_bwd_ada", _p(dy), _p(x), _p(rstd), _p(mean), _p(w), _p(dx), _p(dw), 1 if accumulate else 0, _p(branch), _p(dbranch), _p(w_b), _p(rstd_b),
              _p(mean_b), _p(dw_b), _p(dbias), shift, scale, dshift, dscale, gate, dgate, ms_n or ms_r, _p(modality) if mod_n is not None else None,
              _p(any_img) if mod_n is not None else None, _p(modality_r), M, d, L, norm_type, float(p_drop), int(seed), _p(ws), ws.numel(), _s())
    return dbranch


# ------------------------------------------------------------------------------------------------ attention
def attention_q_scale(D):
    """log2(e) / sqrt(D): folded into the stored q by qknorm_rope_fwd(q_scale=...) on the engine's path, so that the attention kernels' scores are the base-2
    exponents as they leave the matrix pipe (attention_fwd / attention_bwd with q_prescaled=True)."""
    return 1.4426950408889634 / math.sqrt(D)


def qknorm_rope_fwd(qkv, cos, sin, L, D, *, gq=None, bq=None, gk=None, bk=None, q_scale=1.0):
    """qkv bf16 [M, 3d] -> qkr bf16 [M, 2d] (normalised + rotated q | k), LayerNorm statistics.  q_scale != 1: the q half holds bf16(q * q_scale) (one rounding)."""
    M, d3 = qkv.shape
    d = d3 // 3
    qkr = torch.empty((M, 2 * d), dtype=BF16, device=qkv.device)
    stats = torch.empty((M, 4), dtype=F32, device=qkv.device) if gq is not None else None
    per_sample = 1 if cos.dim() == 3 else 0
    _lib.call("udm_qknorm_rope_fwd", _p(qkv), _p(qkr), _p(gq), _p(bq), _p(gk), _p(bk), _p(stats), _p(cos), _p(sin), per_sample, M, d, L, D, 1e-5, float(q_scale), _s())
    return qkr, stats


def qknorm_rope_bwd(dqkr, qkv, dqkv, cos, sin, L, D, *, gq=None, gk=None, stats=None, dgq=None, dbq=None, dgk=None, dbk=None, q_scale=1.0):
    M, d3 = qkv.shape
    d = d3 // 3
    per_sample = 1 if cos.dim() == 3 else 0
    contig = gq is not None and dbq.data_ptr() == dgq.data_ptr() + 4 * d and dgk.data_ptr() == dgq.data_ptr() + 8 * d and dbk.data_ptr() == dgq.data_ptr() + 12 * d
    ws = _scratch(4096 * d, qkv.device) if contig else None
    _lib.call("udm_qknorm_rope_bwd", _p(dqkr), _p(qkv), _p(dqkv), _p(gq), _p(gk), _p(stats), _p(cos), _p(sin), per_sample, _p(dgq), _p(dbq), _p(dgk),
              _p(dbk), M, d, L, D, float(q_scale), _p(ws), ws.numel() if ws is not None else 0, _s())


def modality_mask_codes(txt_drop, img_drop, txt_length, L):
    """Attention mask codes [B, L] int64 (csrc/attention_common.h) of the reference's modality attention dropout (`_attn_mask`, model_utils.py:721-731):
    in samples with `txt_drop` text queries see text keys only, with `img_drop` image queries see image keys only; positions < txt_length are text.
    Sample id 0 everywhere (one document per row); key class 1 = text, 2 = image; query mask = the classes a position may attend to."""
    B = txt_drop.shape[0]
    dev = txt_drop.device
    is_img = (torch.arange(L, device=dev) >= int(txt_length))[None].expand(B, L)
    kbit = torch.where(is_img, 2, 1).to(torch.int64)
    qm_txt = torch.where(txt_drop.reshape(B, 1).bool(), 1, 3)
    qm_img = torch.where(img_drop.reshape(B, 1).bool(), 2, 3)
    qmask = torch.where(is_img, qm_img, qm_txt).to(torch.int64)
    return ((kbit << 32) | (qmask << 40)).contiguous()


def attention_doc_ranges(sample_ids):
    """int32 [B, ceil(L/64), 8] = {lo, hi, idmin, idmax, exact, 0, 0, 0} per 64-row tile: the span of positions that can share a sample id with it
    (tile skipping for packed samples), the tile's id interval (idmin = -1 if it holds padding; idmin == idmax: one document, no per-element id
    test needed) and whether that document's rows are exactly [lo, hi) (then its key blocks take the wave-specialised dK/dV kernel)."""
    B, L = sample_ids.shape
    r = torch.empty((B, (L + 63) // 64, 8), dtype=torch.int32, device=sample_ids.device)
    _lib.call("udm_attention_doc_ranges", _p(sample_ids), B, L, _p(r), _s())
    return r


def attention_fwd(qkr, qkv, B, L, H, D, sample_ids=None, doc_ranges=None, q_prescaled=False):
    """q, k from qkr [M,2d] (normalised+rotated), v from qkv [M,3d] columns [2d,3d).  q_prescaled: q holds q * attention_q_scale(D)."""
    d = H * D
    M = B * L
    o = torch.empty((M, d), dtype=BF16, device=qkr.device)
    lse = torch.empty((B, H, L), dtype=F32, device=qkr.device)
    q_ptr, k_ptr, v_ptr = qkr.data_ptr(), qkr.data_ptr() + 2 * d, qkv.data_ptr() + 4 * d
    _lib.call("udm_attention_fwd", q_ptr, k_ptr, v_ptr, _p(o), _p(lse), _p(sample_ids), _p(doc_ranges), B, H, L, D, 2 * d, 2 * d, 3 * d, d, 1 if q_prescaled else 0, _s())
    return o, lse


def attention_bwd(qkr, qkv, o, do, lse, dqkr, dqkv, B, L, H, D, sample_ids=None, doc_ranges=None, q_prescaled=False):
    """Writes dq|dk (wrt the stored rotated q, k) into dqkr [M,2d] and dv into dqkv[:, 2d:3d]."""
    d = H * D
    delta = torch.empty((3, B, H, L), dtype=F32, device=qkr.device)   # delta | -lse | -delta (include/unidisc_hip.h)
    q_ptr, k_ptr, v_ptr = qkr.data_ptr(), qkr.data_ptr() + 2 * d, qkv.data_ptr() + 4 * d
    dq_ptr, dk_ptr, dv_ptr = dqkr.data_ptr(), dqkr.data_ptr() + 2 * d, dqkv.data_ptr() + 4 * d
    _lib.call("udm_attention_bwd", q_ptr, k_ptr, v_ptr, _p(o), _p(do), _p(lse), _p(delta), dq_ptr, dk_ptr, dv_ptr, _p(sample_ids), _p(doc_ranges), B, H, L, D, 2 * d,
              2 * d, 3 * d, d, do.stride(0), 2 * d, 2 * d, 3 * d, 1 if q_prescaled else 0, _s())


def attention_fwd_generic(q, k, v, B, L, H, D, sample_ids=None, doc_ranges=None, q_prescaled=False):
    """q, k, v: separate contiguous bf16 [B*L, H*D] (unit tests)."""
    d = H * D
    o = torch.empty((B * L, d), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, L), dtype=F32, device=q.device)
    _lib.call("udm_attention_fwd", _p(q), _p(k), _p(v), _p(o), _p(lse), _p(sample_ids), _p(doc_ranges), B, H, L, D, d, d, d, d, 1 if q_prescaled else 0, _s())
    return o, lse


def attention_bwd_generic(q, k, v, o, do, lse, B, L, H, D, sample_ids=None, doc_ranges=None, q_prescaled=False):
    d = H * D
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty((3, B, H, L), dtype=F32, device=q.device)
    _lib.call("udm_attention_bwd", _p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(sample_ids), _p(doc_ranges), B, H, L, D, d, d, d, d,
              d, d, d, d, 1 if q_prescaled else 0, _s())
    return dq, dk, dv


def set_tr_read(enable: bool):
    debug_set("attention_tr_read", 1 if enable else 0)


def set_attention_fwd64(enable):
    """A/B switch: the 64-queries-per-wave forward (csrc/attention_fwd64.hip; head dim 128, no mask, L % 256 == 0) on / off; 2 = on, without the balanced walk
    (whole 256-query blocks only)."""
    debug_set("attention_fwd64", int(enable))


def set_attention_dq64(enable):
    """A/B switch: the 64-queries-per-wave dQ pass (csrc/attention_dq64.hip; head dim 128, no mask, L % 256 == 0, q pre-scaled) on / off; 2 = on, without the balanced
    walk.  Off = attn_bwd_dq_kernel (8 waves, two workgroups per CU)."""
    debug_set("attention_dq64", int(enable))


def set_attention_dkv64(enable):
    """A/B switch: the 64-keys-per-wave dK / dV pass (csrc/attention_dkv64.hip; head dim 128, no mask, L % 256 == 0, q pre-scaled) on / off; 2 = on, without the
    balanced walk (whole 256-key blocks only).  Off = the wave-specialised 8-wave kernel of attention_dkv_ws.hip."""
    debug_set("attention_dkv64", int(enable))


# ------------------------------------------------------------------------------------------------ embedding / CE / adaLN helpers
def embedding_fwd(ids, E, modality=None, Em=None):
    M = ids.numel()
    V, d = E.shape
    x = torch.empty((M, d), dtype=F32, device=E.device)
    _lib.call("udm_embedding_fwd", _p(ids), _p(E), _p(modality), _p(Em), _p(x), M, d, V, _s())
    return x


def embedding_bwd(ids, dx, dE, hot_id, modality=None, dEm=None):
    M = ids.numel()
    V, d = dE.shape
    _lib.call("udm_embedding_bwd", _p(ids), _p(modality), _p(dx), _p(dE), _p(dEm), M, d, V, hot_id, _s())


def subs_ce_fwd(logits, x0, xt, modality, V, Vt, mask_id, restrict):
    M, ld = logits.shape[0], logits.stride(0)
    log_p = torch.empty(M, dtype=F32, device=logits.device)
    lse = torch.empty(M, dtype=F32, device=logits.device)
    _lib.call("udm_subs_ce_fwd", _p(logits), ld, _p(x0), _p(xt), _p(modality), _p(log_p), _p(lse), M, V, Vt, mask_id, 1 if restrict else 0, _s())
    return log_p, lse


def subs_ce_bwd(logits, x0, xt, modality, lse, g, V, Vt, mask_id, restrict, narrow_txt_rows=-1):
    """d logits in place.  narrow_txt_rows >= 0 (with restrict; dit.py split_head): the first narrow_txt_rows rows are the text group of a per-modality head, the
    rest its image group - only the columns the group's GEMMs read are written (text group [0, ceil64(Vt)), image group [floor8(Vt), ld))."""
    M, ld = logits.shape[0], logits.stride(0)
    _lib.call("udm_subs_ce_bwd", _p(logits), ld, _p(x0), _p(xt), _p(modality), _p(lse), _p(g), M, V, Vt, mask_id, 1 if restrict else 0,
              int(narrow_txt_rows) if restrict else -1, _s())


def subs_logprobs(logits, xt, modality, V, Vt, mask_id, restrict, out_dtype=BF16):
    M, ld = logits.shape[0], logits.stride(0)
    out = torch.empty((M, V), dtype=out_dtype, device=logits.device)
    _lib.call("udm_subs_logprobs", _p(logits), ld, _p(xt), _p(modality), _p(out), V, 1 if out_dtype == F32 else 0, M, V, Vt, mask_id, 1 if restrict else 0,
              _s())
    return out


def diffusion_loss(log_p, w_loss, w_std, attention_mask, modality_mask, *, weighted, full_mask=False, text_w=1.0, img_w=1.0, ratio=None):
    """The loss arithmetic of compute_loss in one launch: (nlls [B, L], coef [B, L] = d loss / d log_p, scalars [8] = loss, txt_loss, img_loss, txt_frac,
    img_frac, valid_frac, txt_count, img_count).  log_p fp32 [B, L]; w_loss / w_std fp32 [B] (weights of the optimised loss and of the reported NLLs);
    attention_mask bool [B, L]; modality_mask bool [B, L, 2] (text, image) or None."""
    _chk(log_p, F32, "diffusion_loss log_p"), _chk(w_loss, F32, "diffusion_loss w_loss"), _chk(w_std, F32, "diffusion_loss w_std")
    B, L = log_p.shape
    att = attention_mask.contiguous()
    mm = modality_mask.contiguous() if modality_mask is not None else None
    if att.dtype != torch.bool or (mm is not None and mm.dtype != torch.bool):
        raise TypeError("diffusion_loss: masks must be bool tensors")
    nlls = torch.empty((B, L), dtype=F32, device=log_p.device)
    coef = torch.empty((B, L), dtype=F32, device=log_p.device)
    scalars = torch.empty(8, dtype=F32, device=log_p.device)
    _lib.call("udm_diffusion_loss", _p(log_p.contiguous()), _p(w_loss.contiguous()), _p(w_std.contiguous()), _p(att), _p(mm), _p(nlls), _p(coef), _p(scalars), B, L,
              1 if weighted else 0, 1 if full_mask else 0, float(text_w), float(img_w), -1.0 if ratio is None else float(ratio), _s())
    return nlls, coef, scalars


def timestep_embedding(sigma, out, B, dim=256):
    _lib.call("udm_timestep_embedding", _p(sigma), _p(out), B, dim, _s())


def silu_fwd(x, n=None):
    y = torch.empty_like(x)
    _lib.call("udm_silu_fwd", _p(x), _p(y), x.numel() if n is None else n, _s())
    return y


def silu_bwd(x, dy):
    dx = torch.empty_like(x)
    _lib.call("udm_silu_bwd", _p(x), _p(dy), _p(dx), x.numel(), _s())
    return dx
