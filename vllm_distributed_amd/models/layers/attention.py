"""Attention layer over the paged KV cache, plus the per-step forward context.

The model runner publishes one ``AttentionMetadata`` per step; every layer's
``Attention.forward`` reads it.  A batch is ordered decode-first: the first
``num_decodes`` sequences contribute one query token each and go to the
split-KV decode kernel; the rest (prompt chunks, possibly continuing a cached
prefix) go to the varlen prefill kernel.  Both read K/V through block tables.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from ... import ops


@dataclass
class AttentionMetadata:
    num_decodes: int                 # sequences with q_len == 1, at the front
    num_decode_tokens: int           # == num_decodes
    num_prefills: int
    num_prefill_tokens: int
    slot_mapping: torch.Tensor       # [T] int32 (cache slot per token, -1 = skip)
    block_table: torch.Tensor        # [B, max_blocks] int32
    seq_lens: torch.Tensor           # [B] int32 (context length incl. this step)
    query_start_loc: torch.Tensor    # [B+1] int32
    max_query_len: int
    max_seq_len: int
    # prefill-only views (sequence index rebased to the prefill part)
    prefill_query_start_loc: Optional[torch.Tensor] = None
    max_prefill_query_len: int = 0
    max_decode_seq_len: int = 0
    decode_workspace: Optional[torch.Tensor] = None
    # every prefill sequence of the step starts at position 0 (no cached context, no earlier chunk)
    prefill_fresh: bool = False

    def token_rows(self, positions: torch.Tensor):
        """Per token of the step: the block-table row it reads (None = its own index: a pure decode batch)
        and the number of keys it sees (its position + 1).  For kernels whose rows are tokens
        (``LatentAttention``); computed once per step and kept."""
        if self.num_prefills == 0:
            return None, self.seq_lens
        cached = getattr(self, "_token_rows", None)
        if cached is None:
            t = positions.shape[0]
            idx = torch.arange(t, dtype=self.query_start_loc.dtype, device=positions.device)
            rows = torch.bucketize(idx, self.query_start_loc[1:], right=True).to(torch.int32)
            cached = self._token_rows = (rows, (positions + 1).to(torch.int32))
        return cached


class ForwardContext:
    attn_metadata: Optional[AttentionMetadata] = None
    positions: Optional[torch.Tensor] = None


_CTX = ForwardContext()


def set_forward_context(attn_metadata: Optional[AttentionMetadata],
                        positions: Optional[torch.Tensor] = None) -> None:
    _CTX.attn_metadata = attn_metadata
    _CTX.positions = positions


def get_forward_context() -> ForwardContext:
    return _CTX


class Attention(nn.Module):
    """Owns one layer's slice of the KV cache (bound by the model runner)."""

    KERNEL_HEAD_DIMS = (64, 96, 128, 256)

    def __init__(self, num_heads: int, head_dim: int, num_kv_heads: int, scale: float,
                 layer_idx: int, sliding_window: Optional[int] = None, pad_head_dim: bool = False,
                 softcap: Optional[float] = None):
        super().__init__()
        # ``pad_head_dim`` (callers of ``forward`` only): a head size the attention kernels are not
        # built for (Phi-2's 80) is zero-padded to the next one they are; ``head_dim`` is then the
        # size q / k / v have in the cache and the kernels, ``model_head_dim`` the model's own.
        # Zero columns change neither the scores nor (sliced off again) the output.
        self.model_head_dim = head_dim
        if pad_head_dim and head_dim not in self.KERNEL_HEAD_DIMS and head_dim < self.KERNEL_HEAD_DIMS[-1]:
            head_dim = next(d for d in self.KERNEL_HEAD_DIMS if d > head_dim)
        self.num_heads, self.head_dim, self.num_kv_heads = num_heads, head_dim, num_kv_heads
        self.scale = scale
        # a query sees its last `window` keys only (0 = the whole context)
        self.window = int(sliding_window or 0)
        # scores squashed to softcap * tanh(score / softcap) before the softmax (Gemma-2); 0 = off
        self.softcap = float(softcap or 0.0)
        self.layer_idx = layer_idx
        self.k_cache: Optional[torch.Tensor] = None
        self.v_cache: Optional[torch.Tensor] = None
        # e4m3 cache only: (k_scale[Hkv], v_scale[Hkv]) fp32; stored = value / scale.  None = 1.0
        self.kv_scales: Optional[tuple] = None
        # fp32 [num_heads] each, or None: an extra softmax column per head that is dropped after the softmax
        # (attention sinks: gpt-oss), and the slopes of a linear position bias (ALiBi: Bloom, MPT, Falcon-RW)
        self.sinks: Optional[torch.Tensor] = None
        self.alibi_slopes: Optional[torch.Tensor] = None

    def bind_kv_cache(self, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_scales: Optional[tuple] = None) -> None:
        self.k_cache, self.v_cache, self.kv_scales = k_cache, v_cache, kv_scales

    def kv_cache_shapes(self, num_blocks: int, block_size: int):
        """Shapes of this layer's two cache tensors."""
        dims = (num_blocks, self.num_kv_heads, self.head_dim, block_size)
        return ops.k_cache_shape(*dims), ops.v_cache_shape(*dims)

    def kv_values_per_token(self) -> int:
        return 2 * self.num_kv_heads * self.head_dim

    def set_kv_scale(self, which: str, value: torch.Tensor) -> None:
        """Checkpoint ``k_scale`` / ``v_scale`` (a scalar, or one value per local KV head)."""
        ks, vs = self.kv_scales or (None, None)
        t = value.detach().float().reshape(-1)
        t = (t.expand(self.num_kv_heads) if t.numel() == 1 else t).contiguous().clone()
        ones = torch.ones(self.num_kv_heads, dtype=torch.float32, device=t.device)
        ks, vs = (t, vs if vs is not None else ones) if which == "k" else (ks if ks is not None else ones, t)
        self.kv_scales = (ks, vs)

    def forward_cached(self, q: torch.Tensor) -> torch.Tensor:
        """``q[T, Hq*D]`` (already rotated; this step's K/V already in the cache)."""
        md = _CTX.attn_metadata
        t = q.shape[0]
        qv = q.view(t, self.num_heads, self.head_dim)
        if md is None:               # memory-profiling pass: no cache bound
            return q
        out = torch.empty((t, self.num_heads, self.head_dim), dtype=q.dtype, device=q.device)
        nd = md.num_decode_tokens
        if nd > 0:
            ops.paged_attention_decode(
                qv[:nd], self.k_cache, self.v_cache, md.block_table[:md.num_decodes],
                md.seq_lens[:md.num_decodes], self.scale, md.max_decode_seq_len,
                out=out[:nd], workspace=md.decode_workspace, kv_scales=self.kv_scales, window=self.window,
                softcap=self.softcap, sinks=self.sinks, alibi_slopes=self.alibi_slopes)
        if md.num_prefills > 0:
            ops.paged_attention(
                qv[nd:], self.k_cache, self.v_cache, md.block_table[md.num_decodes:],
                md.prefill_query_start_loc, md.seq_lens[md.num_decodes:], self.scale,
                md.max_prefill_query_len, md.max_seq_len, out=out[nd:], kv_scales=self.kv_scales,
                window=self.window, softcap=self.softcap, sinks=self.sinks, alibi_slopes=self.alibi_slopes)
        return out.view(t, self.num_heads * self.head_dim)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """Unfused path: write k/v (``[T, Hkv*D]``) to the cache, then attend."""
        md = _CTX.attn_metadata
        if self.head_dim != self.model_head_dim:
            t, extra = q.shape[0], self.head_dim - self.model_head_dim
            pad = torch.nn.functional.pad
            if md is None:
                return q
            k = pad(k.reshape(t, self.num_kv_heads, self.model_head_dim), (0, extra))
            v = pad(v.reshape(t, self.num_kv_heads, self.model_head_dim), (0, extra))
            ops.reshape_and_cache(k, v, self.k_cache, self.v_cache, md.slot_mapping, self.kv_scales)
            qp = pad(q.reshape(t, self.num_heads, self.model_head_dim), (0, extra))
            out = self.forward_cached(qp.view(t, self.num_heads * self.head_dim))
            return out.view(t, self.num_heads, self.head_dim)[..., :self.model_head_dim].reshape(
                t, self.num_heads * self.model_head_dim)
        if md is not None:
            t = k.shape[0]
            ops.reshape_and_cache(k.reshape(t, self.num_kv_heads, self.head_dim),
                                  v.reshape(t, self.num_kv_heads, self.head_dim),
                                  self.k_cache, self.v_cache, md.slot_mapping, self.kv_scales)
        return self.forward_cached(q)


class LatentAttention(Attention):
    """Multi-head latent attention with the up-projections absorbed into the query and the output: every head
    attends over ONE shared key ``[normalised c_kv | rotated k_rope]`` (``rank + rope`` wide) whose first ``rank``
    values are also the value.  The cache holds that row twice - as a key row and in the V^T fragment layout - so
    ``rank + rope + rank`` values per token where per-head keys and values would take ``heads * 2 * head_dim``."""
    latent = True

    def __init__(self, num_heads: int, rank: int, rope: int, scale: float, layer_idx: int):
        super().__init__(num_heads, rank + rope, 1, scale, layer_idx)
        self.v_dim = rank

    def kv_cache_shapes(self, num_blocks: int, block_size: int):
        return (ops.k_cache_shape(num_blocks, 1, self.head_dim, block_size),
                ops.v_cache_shape(num_blocks, 1, self.v_dim, block_size))

    def kv_values_per_token(self) -> int:
        return self.head_dim + self.v_dim

    def forward_latent(self, q: torch.Tensor, latent: torch.Tensor) -> torch.Tensor:
        """``q[T, H, rank + rope]`` (absorbed, rotated), ``latent[T, rank + rope]`` of this step's tokens:
        writes the cache, returns ``[T, H, rank]``."""
        md = _CTX.attn_metadata
        if md is None:                          # memory-profiling pass: no cache bound
            return q[..., :self.v_dim].contiguous()
        ops.mla_cache_write(latent, self.k_cache, self.v_cache, md.slot_mapping)
        return self.attend(q)

    def attend(self, q: torch.Tensor) -> torch.Tensor:
        """``q[T, H, rank + rope]`` against the cache (this step's rows already written) -> ``[T, H, rank]``."""
        md = _CTX.attn_metadata
        rows, lens = md.token_rows(_CTX.positions)
        return ops.mla_attention(q, self.k_cache, self.v_cache, md.block_table, rows, lens, self.scale, md.max_seq_len)


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float, dtype: torch.dtype):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)
        self.eps = eps

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        """``x`` may be a value its producer left unfinished - a projection's K-split partial sums
        (``layers.linear.PartialSums``) or the expert rows of a sparse block before their weighted sum
        (``layers.moe.ExpertRows``): with a residual, the kernel that adds and normalises finishes it."""
        if not isinstance(x, torch.Tensor):
            if residual is not None:
                return x.add_rms_norm(residual, self.weight, self.eps)
            x = x.materialize()
        if residual is None:
            return ops.rms_norm(x, self.weight, self.eps)
        return ops.fused_add_rms_norm(x, residual, self.weight, self.eps)


class LayerNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float, dtype: torch.dtype, bias: bool = True):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(hidden_size, dtype=dtype), requires_grad=False) if bias else None
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.layer_norm(x, self.weight, self.bias, self.eps)
