"""Constrained decoding support: the byte-level JSON automaton of
``csrc/core/json_grammar.cpp`` (host library ``_C/libvdamd_core.so``) and the
per-vocabulary token byte table it walks.

vLLM serves ``response_format={"type": "json_object"}`` through its guided-decoding
backends; the reference inherits that with the rest of vLLM's OpenAI app
(reference ``src/launch.py:413-457``).  Here the grammar is an explicit automaton
and the mask is computed per step on the host, next to the other logit edits.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import numpy as np

_LIB: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.environ.get("VDAMD_CORE_LIB") or os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_C", "libvdamd_core.so")
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: run `python -m vllm_distributed_amd.build`")
        lib = ctypes.CDLL(path)
        P, L = ctypes.c_void_p, ctypes.c_long
        lib.vdamd_json_state_bytes.restype = ctypes.c_int
        lib.vdamd_json_reset.argtypes = [P]
        lib.vdamd_json_reset.restype = None
        lib.vdamd_json_advance.argtypes = [P, ctypes.c_char_p, L]
        lib.vdamd_json_advance.restype = ctypes.c_int
        lib.vdamd_json_done.argtypes = [P]
        lib.vdamd_json_done.restype = ctypes.c_int
        lib.vdamd_json_mask.argtypes = [P, P, P, L, P]
        lib.vdamd_json_mask.restype = L
        _LIB = lib
    return _LIB


class TokenTable:
    """Every vocabulary entry's bytes, concatenated, with offsets (the layout the mask walk reads)."""

    def __init__(self, token_bytes: List[bytes], vocab_size: int):
        toks = list(token_bytes[:vocab_size]) + [b""] * max(0, vocab_size - len(token_bytes))
        self.n = len(toks)
        self.tokens = toks
        self.offsets = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum([len(t) for t in toks], out=self.offsets[1:])
        self.blob = np.frombuffer(b"".join(toks) or b"\0", dtype=np.uint8).copy()
        self.mask_cache: dict = {}
        self._by_bytes: Optional[dict] = None

    @property
    def by_bytes(self) -> dict:
        """bytes -> token ids spelled exactly so (built on first use)."""
        if self._by_bytes is None:
            d: dict = {}
            for i, t in enumerate(self.tokens):
                if t:
                    d.setdefault(t, []).append(i)
            self._by_bytes = d
        return self._by_bytes


class JsonGrammar:
    """One request's position in the JSON-object grammar."""

    def __init__(self) -> None:
        self.lib = load()
        self.state = ctypes.create_string_buffer(self.lib.vdamd_json_state_bytes())
        self.lib.vdamd_json_reset(self.state)

    def advance(self, data: bytes) -> bool:
        """Consume ``data``; False (and no change) when it does not continue a JSON object."""
        return self.lib.vdamd_json_advance(self.state, data, len(data)) == 0

    @property
    def done(self) -> bool:
        return bool(self.lib.vdamd_json_done(self.state))

    def mask(self, table: TokenTable) -> np.ndarray:
        """uint8 [vocab]: 1 for tokens that may come next.  Masks are remembered per grammar
        position on the table (inside a string the position does not change from token to
        token, so most steps of a request are a dictionary hit); treat the result as read-only."""
        key = self.state.raw
        out = table.mask_cache.get(key)
        if out is None:
            out = np.empty(table.n, dtype=np.uint8)
            self.lib.vdamd_json_mask(self.state, table.blob.ctypes.data, table.offsets.ctypes.data, table.n,
                                     out.ctypes.data)
            if len(table.mask_cache) >= 512:
                table.mask_cache.clear()
            table.mask_cache[key] = out
        return out


class ChoiceGrammar:
    """Output restricted to one of a few fixed strings (vLLM's ``guided_choice``).  The
    position is the text produced so far; a token may come next when its bytes are a
    prefix of what is left of some still-matching choice, found by looking every such
    prefix up in the vocabulary (no walk over the vocabulary)."""

    def __init__(self, choices: List[str]) -> None:
        self.choices = [c.encode("utf-8") for c in choices if c]
        if not self.choices:
            raise ValueError("guided_choice needs at least one non-empty string")
        self.text = b""

    def _live(self) -> List[bytes]:
        return [c for c in self.choices if c.startswith(self.text)]

    def advance(self, data: bytes) -> bool:
        new = self.text + data
        if not any(c.startswith(new) for c in self.choices):
            return False
        self.text = new
        return True

    @property
    def may_stop(self) -> bool:
        """The text is one of the choices (EOS is allowed next)."""
        return self.text in self.choices

    @property
    def done(self) -> bool:
        """A complete choice that no other choice extends: only EOS may follow."""
        return self.may_stop and not any(len(c) > len(self.text) for c in self._live())

    def mask(self, table: TokenTable) -> np.ndarray:
        out = np.zeros(table.n, dtype=np.uint8)
        lookup = table.by_bytes
        for c in self._live():
            rest = c[len(self.text):]
            for n in range(1, len(rest) + 1):
                ids = lookup.get(rest[:n])
                if ids:
                    out[ids] = 1
        return out
