"""Rotary tables (host-side, built once at module construction)."""
from __future__ import annotations

import torch


def rotary_table_1d(seq_len: int, dim: int, base: float = 10000.0):
    """cos/sin fp32 [seq_len, dim/2]: reference ``Rotary`` (models/dit.py:307-330) sliced as in :1226-1239."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(seq_len).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return freqs.cos(), freqs.sin()


def lumina_rope_2d(embed_dim, len_h, len_w, linear_factor=1.0, ntk_factor=1.0):
    """Lumina 2-D RoPE table, complex64 [len_h, len_w, embed_dim/2], last dim interleaved [h0, w0, h1, w1, ...].

    Used only when ``diffusers`` (the reference's source of this table, models/dit.py:12,1052) is not installed.
    Restates diffusers 0.32.2 ``get_2d_rotary_pos_embed_lumina``; PARITY UNPINNED (third-party, un-vendored) —
    kernels take the tables as data, so only end-to-end parity with pretrained checkpoints depends on it.
    """
    assert embed_dim % 4 == 0
    half = embed_dim // 2
    theta = 10000.0 * ntk_factor
    freqs = 1.0 / (theta ** (torch.arange(0, half, 2, dtype=torch.float32)[: half // 2] / half)) / linear_factor
    fh = torch.outer(torch.arange(len_h, dtype=torch.float32), freqs)
    fw = torch.outer(torch.arange(len_w, dtype=torch.float32), freqs)
    eh = torch.polar(torch.ones_like(fh), fh).view(len_h, 1, half // 2, 1).repeat(1, len_w, 1, 1)
    ew = torch.polar(torch.ones_like(fw), fw).view(1, len_w, half // 2, 1).repeat(len_h, 1, 1, 1)
    return torch.cat([eh, ew], dim=-1).flatten(2)
