"""Weight-only 4-bit (W4A16) tensors in the MFMA-operand-native layout.

``pack_w4`` turns integer weights ``q[N, K]`` (0..15), zero points ``z[N, K/G]``
and scales ``s[N, K/G]`` into

* ``qw  [N/16, K/128, 64, 4] int32`` - tile ``(nt, kb)`` is 1 KiB; lane
  ``c + 16 g`` holds row ``16 nt + c``, columns ``128 kb + 32 g + 8 w + e`` in word
  ``w``, element ``e`` in nibble ``(e >> 1) + 4 (e & 1)``;
* ``sz  [N/16, K/G, 16, 2]`` in the activation dtype - ``(scale, OFF + zero)``
  with ``OFF`` = 128 (bf16) or 1024 (fp16), the value whose exponent bits the
  kernel ORs onto a nibble to make it a float.

The kernels in ``csrc/quant_gemm.hip`` consume this layout; the functions here
are the layout's definition and the reference semantics (``unpack_w4`` +
``F.linear``).  AWQ checkpoints (the reference's documented model is AWQ,
``.env.server:11``) are converted with ``awq_unpack`` + ``pack_w4`` after the
tensor-parallel shards have been cut.
"""
from __future__ import annotations

from typing import Tuple

import torch

_AWQ_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]
_NIBBLE_OF_ELEM = [(e >> 1) + 4 * (e & 1) for e in range(8)]


def w4_offset(dtype: torch.dtype) -> float:
    if dtype in (torch.bfloat16, torch.float32):      # fp32 only on the reference path
        return 128.0
    if dtype == torch.float16:
        return 1024.0
    raise TypeError(f"W4A16 needs bf16/fp16 activations, got {dtype}")


def awq_unpack(packed: torch.Tensor) -> torch.Tensor:
    """AWQ ``[R, C/8] int32`` -> ``[R, C] uint8`` (nibble order 0,4,1,5,2,6,3,7)."""
    shifts = torch.tensor([4 * _AWQ_ORDER.index(i) for i in range(8)], dtype=torch.int32,
                          device=packed.device)
    u = (packed.unsqueeze(-1) >> shifts) & 0xF
    return u.reshape(packed.shape[0], -1).to(torch.uint8)


def awq_pack(vals: torch.Tensor) -> torch.Tensor:
    """Inverse of ``awq_unpack``: ``[R, C]`` ints in 0..15 -> ``[R, C/8] int32``."""
    r, c = vals.shape
    v = vals.to(torch.int64).reshape(r, c // 8, 8)
    out = torch.zeros((r, c // 8), dtype=torch.int64, device=vals.device)
    for i in range(8):
        out |= v[..., i] << (4 * _AWQ_ORDER.index(i))
    out = torch.where(out >= (1 << 31), out - (1 << 32), out)
    return out.to(torch.int32)


def gptq_to_awq(name: str, w: torch.Tensor, zero_offset: int = 1) -> torch.Tensor:
    """One tensor of a 4-bit GPTQ linear (``qweight [in/8, out]``: eight input rows per word, low nibble
    first; ``qzeros [in/g, out/8]``: eight output columns per word in natural order, stored minus
    ``zero_offset`` by the original format; ``scales [in/g, out]``) in the AWQ arrangement the
    W4A16 path consumes (``qweight [in, out/8]``, ``qzeros [in/g, out/8]``, AWQ nibble order).
    The two formats hold the same integers: only the packing differs."""
    shifts = torch.arange(0, 32, 4, dtype=torch.int32, device=w.device)
    if name.endswith(".qweight"):
        rows = ((w.unsqueeze(1) >> shifts.view(1, 8, 1)) & 0xF)           # [in/8, 8, out]
        return awq_pack(rows.reshape(-1, w.shape[1]))
    if name.endswith(".qzeros"):
        z = ((w.unsqueeze(-1) >> shifts) & 0xF).reshape(w.shape[0], -1) + zero_offset
        if int(z.max()) > 15:
            raise NotImplementedError(f"{name}: a GPTQ zero point of 16 does not fit the 4-bit zero points "
                                      "of the W4A16 kernels (load with --quantization awq-dequant)")
        return awq_pack(z)
    return w


def pack_w4(q: torch.Tensor, z: torch.Tensor, s: torch.Tensor, dtype: torch.dtype,
            ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``q [N, K]``, ``z [N, K/G]``, ``s [N, K/G]`` -> ``(qw, sz)`` as described above."""
    n, k = q.shape
    ng = s.shape[1]
    assert n % 16 == 0 and k % 128 == 0 and k % ng == 0 and (k // ng) % 128 == 0, (n, k, ng)
    assert z.shape == s.shape == (n, ng)
    v = q.to(torch.int64).reshape(n // 16, 16, k // 128, 4, 4, 8)        # nt c kb g w e
    v = v.permute(0, 2, 3, 1, 4, 5).reshape(n // 16, k // 128, 64, 4, 8)  # nt kb (g c) w e
    word = torch.zeros(v.shape[:-1], dtype=torch.int64, device=q.device)
    for e in range(8):
        word |= v[..., e] << (4 * _NIBBLE_OF_ELEM[e])
    word = torch.where(word >= (1 << 31), word - (1 << 32), word)
    qw = word.to(torch.int32).contiguous()
    off = w4_offset(dtype)
    sc = s.to(torch.float32).reshape(n // 16, 16, ng).permute(0, 2, 1)
    zc = z.to(torch.float32).reshape(n // 16, 16, ng).permute(0, 2, 1) + off
    sz = torch.stack([sc, zc], dim=-1).to(dtype).contiguous()
    return qw, sz


def unpack_w4(qw: torch.Tensor, sz: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Inverse of ``pack_w4``: ``(q [N,K] uint8, z [N,K/G] f32, s [N,K/G] f32)``."""
    nt, kb = qw.shape[0], qw.shape[1]
    w = qw.to(torch.int64) & 0xFFFFFFFF
    elems = [(w >> (4 * _NIBBLE_OF_ELEM[e])) & 0xF for e in range(8)]
    v = torch.stack(elems, dim=-1).reshape(nt, kb, 4, 16, 4, 8)          # nt kb g c w e
    q = v.permute(0, 3, 1, 2, 4, 5).reshape(nt * 16, kb * 128).to(torch.uint8)
    ng = sz.shape[1]
    s = sz[..., 0].to(torch.float32).permute(0, 2, 1).reshape(nt * 16, ng)
    z = sz[..., 1].to(torch.float32).permute(0, 2, 1).reshape(nt * 16, ng) - w4_offset(sz.dtype)
    return q, z, s


def w4_dequant(qw: torch.Tensor, sz: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Dense ``[N, K]`` weights ``(q - z) * s`` (reference path)."""
    q, z, s = unpack_w4(qw, sz)
    g = q.shape[1] // s.shape[1]
    w = (q.to(torch.float32) - z.repeat_interleave(g, dim=1)) * s.repeat_interleave(g, dim=1)
    return w.to(dtype)


def w4a16_gemm(x: torch.Tensor, qw: torch.Tensor, sz: torch.Tensor) -> torch.Tensor:
    """Reference ``x [M, K] -> [M, N]``: fp32 accumulation over exact ``(q-z)*s``."""
    return (x.to(torch.float32) @ w4_dequant(qw, sz).t()).to(x.dtype)


def quantize_w4(w: torch.Tensor, group: int = 128) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Asymmetric round-to-nearest 4-bit quantisation of dense ``w [N, K]`` per
    ``group`` input columns: ``(q, z, s)`` with ``w ~ (q - z) * s``."""
    n, k = w.shape
    wg = w.to(torch.float32).reshape(n, k // group, group)
    lo, hi = wg.amin(-1), wg.amax(-1)
    s = ((hi - lo) / 15.0).clamp_min(1e-8)
    z = (-lo / s).round().clamp(0, 15)
    q = (wg / s.unsqueeze(-1) + z.unsqueeze(-1)).round().clamp(0, 15)
    return q.reshape(n, k).to(torch.uint8), z, s


# ------------------------------------------------------------------ NVFP4
# E2M1 code -> value; code bit 3 is the sign (``Dockerfile:5-6`` of the
# reference installs the fp4 kernels its documented deployments rely on).
FP4_VALUES = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]


def fp4_lut(device=None) -> torch.Tensor:
    v = torch.tensor(FP4_VALUES, dtype=torch.float32, device=device)
    return torch.cat([v, -v])


def pack_fp4(packed: torch.Tensor, block_scale: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """NVFP4 checkpoint tensors -> kernel layout.  ``packed [N, K/2] uint8``
    holds element ``2i`` in the low nibble of byte ``i``; ``block_scale
    [N, K/16]`` is ``float8_e4m3fn`` (one scale per 16 input columns).  Returns
    ``qw [N/16, K/128, 64, 4] int32`` (same tiles as ``pack_w4``, element ``e`` in
    nibble ``e``) and ``fs [N/16, K/128, 64] int32`` - lane ``c + 16 g`` holds the
    scales of row ``c``, blocks ``8 kb + 2 g`` (low half) and ``8 kb + 2 g + 1``
    (high half), as bf16 bit patterns (every e4m3 value is exact in bf16)."""
    n, kh = packed.shape
    k = kh * 2
    assert packed.dtype == torch.uint8 and n % 16 == 0 and k % 128 == 0, (packed.dtype, n, k)
    assert block_scale.shape == (n, k // 16), (block_scale.shape, n, k)
    words = packed.contiguous().view(torch.int32).reshape(n // 16, 16, k // 128, 4, 4)      # nt c kb g w
    qw = words.permute(0, 2, 3, 1, 4).reshape(n // 16, k // 128, 64, 4).contiguous()
    sb = block_scale.to(torch.float32).to(torch.bfloat16).reshape(n // 16, 16, k // 128, 4, 2)   # nt c kb g j
    fs = sb.permute(0, 2, 3, 1, 4).reshape(n // 16, k // 128, 64, 2).contiguous().view(torch.int32)
    return qw, fs.reshape(n // 16, k // 128, 64)


def unpack_fp4(qw: torch.Tensor, fs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inverse of ``pack_fp4``: ``(codes [N, K] uint8 in 0..15, block_scale [N, K/16] f32)``."""
    nt, kb = qw.shape[0], qw.shape[1]
    w = qw.reshape(nt, kb, 4, 16, 4).permute(0, 3, 1, 2, 4).contiguous()               # nt c kb g w
    by = w.view(torch.uint8).reshape(nt * 16, kb * 64)                                # [N, K/2]
    codes = torch.stack([by & 0xF, by >> 4], dim=-1).reshape(nt * 16, kb * 128)
    sb = fs.contiguous().view(torch.bfloat16).reshape(nt, kb, 4, 16, 2).permute(0, 3, 1, 2, 4)
    return codes, sb.reshape(nt * 16, kb * 8).to(torch.float32)


def fp4_dequant(qw: torch.Tensor, fs: torch.Tensor, alpha: torch.Tensor = None,
                dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Dense ``[N, K]`` weights ``e2m1 * block_scale * alpha[row]`` (reference path)."""
    codes, sc = unpack_fp4(qw, fs)
    w = fp4_lut(qw.device)[codes.long()] * sc.repeat_interleave(16, dim=1)
    if alpha is not None:
        w = w * alpha.to(torch.float32).reshape(-1, 1)
    return w.to(dtype)


def fp4a16_gemm(x: torch.Tensor, qw: torch.Tensor, fs: torch.Tensor, alpha: torch.Tensor = None) -> torch.Tensor:
    return (x.to(torch.float32) @ fp4_dequant(qw, fs, alpha).t()).to(x.dtype)


def quantize_fp4(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Round-to-nearest NVFP4 quantisation of dense ``w [N, K]``:
    ``(packed [N, K/2] uint8, block_scale [N, K/16] float8_e4m3fn, global_scale [1] f32)``
    with ``w ~ e2m1 * block_scale / global_scale`` - the tensors (and the
    divide-by-global-scale convention) of a compressed-tensors
    ``nvfp4-pack-quantized`` checkpoint."""
    n, k = w.shape
    wf = w.to(torch.float32)
    amax = wf.abs().max().clamp_min(1e-12)
    gscale = (448.0 * 6.0 / amax).reshape(1)
    blocks = wf.reshape(n, k // 16, 16)
    bs = (blocks.abs().amax(-1) / 6.0 * gscale).clamp(max=448.0).to(torch.float8_e4m3fn)
    eff = (bs.to(torch.float32) / gscale).clamp_min(1e-30).unsqueeze(-1)
    lut = torch.tensor(FP4_VALUES, dtype=torch.float32, device=w.device)
    mag = (blocks.abs() / eff).clamp(max=6.0)
    code = (mag.unsqueeze(-1) - lut).abs().argmin(-1)
    code = (code + 8 * (blocks < 0)).to(torch.uint8).reshape(n, k // 2, 2)
    packed = code[..., 0] | (code[..., 1] << 4)
    return packed.contiguous(), bs, gscale
