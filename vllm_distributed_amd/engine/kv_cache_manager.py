"""Paged KV-cache block manager with hash-based prefix caching.

The GPU KV cache is ``num_blocks`` fixed-size blocks per layer; this manager
owns the block ids.  A full block's identity is the hash chain of the tokens
it and its predecessors hold, so two requests with a common prefix share
blocks (ref-counted).  Freed blocks keep their hash and stay reusable until
the allocator evicts them in LRU order.  (The reference obtains this from the
vLLM engine it embeds - reference ``src/launch.py:407-410``.)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

from ..utils import cdiv
from .request import Request


class KVCacheBlock:
    __slots__ = ("block_id", "ref_cnt", "block_hash")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_cnt = 0
        self.block_hash: Optional[int] = None


def hash_block_tokens(parent_hash: Optional[int], token_ids) -> int:
    return hash((parent_hash, tuple(token_ids)))


class BlockPool:
    def __init__(self, num_blocks: int, enable_caching: bool):
        self.num_blocks = num_blocks
        self.enable_caching = enable_caching
        self.blocks = [KVCacheBlock(i) for i in range(num_blocks)]
        # LRU order: head = evict first
        self.free: "OrderedDict[int, None]" = OrderedDict((i, None) for i in range(num_blocks))
        self.cached: Dict[int, int] = {}   # block_hash -> block_id

    @property
    def num_free_blocks(self) -> int:
        return len(self.free)

    def get_cached_block(self, block_hash: int) -> Optional[KVCacheBlock]:
        bid = self.cached.get(block_hash)
        return None if bid is None else self.blocks[bid]

    def touch(self, block: KVCacheBlock) -> None:
        """Take a reference; a free-but-cached block leaves the free list."""
        if block.ref_cnt == 0:
            self.free.pop(block.block_id, None)
        block.ref_cnt += 1

    def allocate(self, n: int) -> List[KVCacheBlock]:
        if n > len(self.free):
            raise RuntimeError("out of KV blocks")
        out = []
        for _ in range(n):
            bid, _ = self.free.popitem(last=False)
            blk = self.blocks[bid]
            if blk.block_hash is not None:       # evict its cached identity
                if self.cached.get(blk.block_hash) == bid:
                    del self.cached[blk.block_hash]
                blk.block_hash = None
            blk.ref_cnt = 1
            out.append(blk)
        return out

    def release(self, blocks: List[KVCacheBlock]) -> None:
        """Drop one reference each; ``blocks`` should be ordered so the ones to
        evict first (sequence tails) come first."""
        for blk in blocks:
            blk.ref_cnt -= 1
            assert blk.ref_cnt >= 0
            if blk.ref_cnt == 0:
                self.free[blk.block_id] = None

    def set_hash(self, block: KVCacheBlock, block_hash: int) -> None:
        if not self.enable_caching or block.block_hash is not None:
            return
        block.block_hash = block_hash
        self.cached.setdefault(block_hash, block.block_id)

    def reset_prefix_cache(self) -> bool:
        if len(self.free) != self.num_blocks:
            return False
        self.cached.clear()
        for b in self.blocks:
            b.block_hash = None
        return True


class KVCacheManager:
    # The window is treated as this much wider when blocks are released: the attention kernels fetch
    # whole 64-key tiles (the one the window starts in is read and masked), a verification step may
    # hand positions back (rejected speculative drafts, at most 16), and one step is in flight under
    # async scheduling.
    WINDOW_MARGIN = 128

    def __init__(self, num_blocks: int, block_size: int, max_model_len: int,
                 enable_caching: bool = True, sliding_window: Optional[int] = None):
        # every attention layer sees at most this many most recent keys (None: some layer sees all)
        self.window = int(sliding_window) if sliding_window else None
        self.num_released: Dict[str, int] = {}       # leading blocks already given back, per request
        self.block_size = block_size
        self.max_model_len = max_model_len
        self.enable_caching = enable_caching
        self.pool = BlockPool(num_blocks, enable_caching)
        self.req_blocks: Dict[str, List[KVCacheBlock]] = {}
        self.num_cached_block: Dict[str, int] = {}   # full blocks already hashed per request
        self.prefix_queries = 0
        self.prefix_hits = 0

    # ----------------------------------------------------------------- stats
    @property
    def num_free_blocks(self) -> int:
        return self.pool.num_free_blocks

    @property
    def usage(self) -> float:
        return 1.0 - self.pool.num_free_blocks / max(1, self.pool.num_blocks)

    # ---------------------------------------------------------- prefix cache
    def _extend_hashes(self, request: Request, num_tokens: int) -> None:
        """Make ``request.block_hashes`` cover every full block in the first
        ``num_tokens`` tokens."""
        bs = self.block_size
        toks = request.all_token_ids
        n_full = min(num_tokens, len(toks)) // bs
        hashes = request.block_hashes
        # KV produced under an adapter is only reusable under the same adapter
        lora = getattr(request, "lora_request", None)
        parent = hashes[-1] if hashes else (None if lora is None else hash(("lora", lora.lora_int_id)))
        mm = getattr(request, "mm_inputs", None)
        for i in range(len(hashes), n_full):
            if mm:
                # image placeholders read the same for every image: a block over one also hashes its pixels
                from ..multimodal import block_mm_salt
                salt = block_mm_salt(mm, i * bs, (i + 1) * bs)
                if salt is not None:
                    parent = hash((parent, "mm") + salt)
            parent = hash_block_tokens(parent, toks[i * bs:(i + 1) * bs])
            hashes.append(parent)

    def get_computed_blocks(self, request: Request) -> Tuple[List[KVCacheBlock], int]:
        """Longest cached prefix of the request's tokens.  Never covers the whole
        sequence: at least one token must be computed to produce logits."""
        if (not self.enable_caching or request.sampling_params.prompt_logprobs is not None
                or request.sampling_params.pooling == "mean"):
            return [], 0          # prompt logprobs need every prompt position computed
        self._extend_hashes(request, request.num_tokens)
        max_hit_blocks = (request.num_tokens - 1) // self.block_size
        blocks: List[KVCacheBlock] = []
        for h in request.block_hashes[:max_hit_blocks]:
            blk = self.pool.get_cached_block(h)
            if blk is None:
                break
            blocks.append(blk)
        self.prefix_queries += request.num_tokens
        self.prefix_hits += len(blocks) * self.block_size
        return blocks, len(blocks) * self.block_size

    # ------------------------------------------------------------ allocation
    def allocate_slots(self, request: Request, num_new_tokens: int,
                       new_computed_blocks: Optional[List[KVCacheBlock]] = None,
                       ) -> Optional[List[int]]:
        """Ensure the request owns blocks for ``num_computed_tokens +
        num_new_tokens`` tokens.  Returns the ids of newly attached blocks
        (cached hits included on first allocation) or None when the pool
        cannot satisfy it."""
        new_computed_blocks = new_computed_blocks or []
        held = self.req_blocks.get(request.request_id, [])
        num_computed = request.num_computed_tokens + len(new_computed_blocks) * self.block_size \
            if not held else request.num_computed_tokens
        need_total = cdiv(num_computed + num_new_tokens, self.block_size)
        n_new = need_total - len(held) - len(new_computed_blocks)
        # cached blocks sitting in the free list will be claimed by touch()
        n_free_cached = sum(1 for b in new_computed_blocks if b.ref_cnt == 0)
        if n_new > self.pool.num_free_blocks - n_free_cached:
            return None
        for b in new_computed_blocks:
            self.pool.touch(b)
        fresh = self.pool.allocate(n_new) if n_new > 0 else []
        if not held:
            self.req_blocks[request.request_id] = held = []
            self.num_cached_block[request.request_id] = len(new_computed_blocks)
        attached = new_computed_blocks + fresh
        held.extend(attached)
        return [b.block_id for b in attached]

    def cache_blocks(self, request: Request, num_computed_tokens: int) -> None:
        """Publish hashes for blocks that became full (call after the tokens
        up to ``num_computed_tokens`` are known)."""
        if not self.enable_caching:
            return
        held = self.req_blocks.get(request.request_id)
        if not held:
            return
        n_full = min(num_computed_tokens, request.num_tokens) // self.block_size
        start = self.num_cached_block.get(request.request_id, 0)
        if n_full <= start:
            return
        self._extend_hashes(request, n_full * self.block_size)
        for i in range(start, min(n_full, len(held))):
            if held[i] is not None:
                self.pool.set_hash(held[i], request.block_hashes[i])
        self.num_cached_block[request.request_id] = n_full

    def release_out_of_window(self, request: Request) -> int:
        """Give back the leading blocks of a running request that no attention layer will read again:
        with every layer windowed, the next query (position ``num_computed_tokens``) and all later ones
        see keys from ``num_computed_tokens - window + 1`` on.  The request keeps its place in the
        block table (workers hold the stale ids and never dereference them: the kernels skip whole
        key tiles below the window); a released block that carries a prefix hash stays reusable
        until it is evicted.  Returns the number of blocks released."""
        if self.window is None:
            return 0
        held = self.req_blocks.get(request.request_id)
        if not held:
            return 0
        dead = (request.num_computed_tokens - self.window - self.WINDOW_MARGIN) // self.block_size
        done = self.num_released.get(request.request_id, 0)
        dead = min(dead, len(held))
        if dead <= done:
            return 0
        gone = [b for b in held[done:dead] if b is not None]
        for i in range(done, dead):
            held[i] = None
        self.num_released[request.request_id] = dead
        self.pool.release(gone)          # oldest first: evicted first
        return len(gone)

    def free(self, request: Request) -> None:
        held = self.req_blocks.pop(request.request_id, None)
        self.num_cached_block.pop(request.request_id, None)
        self.num_released.pop(request.request_id, None)
        if held:
            self.pool.release([b for b in reversed(held) if b is not None])

    def get_block_ids(self, request_id: str) -> List[int]:
        """Block ids in sequence order; blocks released behind the window read as block 0 (a valid
        index nobody dereferences)."""
        return [0 if b is None else b.block_id for b in self.req_blocks.get(request_id, [])]

    def reset_prefix_cache(self) -> bool:
        return self.pool.reset_prefix_cache()
