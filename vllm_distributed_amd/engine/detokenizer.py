"""Incremental detokenisation with stop-string handling."""
from __future__ import annotations

from typing import List, Optional

from ..sampling_params import SamplingParams


class IncrementalDetokenizer:
    """Decodes a growing token sequence without re-decoding all of it: keeps a
    short window of recent tokens (``prefix_offset..read_offset``) so that
    multi-token characters and leading-space rules come out right, and only
    emits text once it no longer ends in an incomplete UTF-8 sequence."""

    def __init__(self, tokenizer, prompt_token_ids: List[int], sp: SamplingParams):
        self.tok = tokenizer
        self.sp = sp
        self.token_ids: List[int] = list(prompt_token_ids[-6:])
        self.prefix_offset = 0
        self.read_offset = len(self.token_ids)
        self.text = ""
        self._emitted = 0
        self.stop = list(sp.stop or [])
        self._hold = max((len(s) for s in self.stop), default=1) - 1

    def append(self, new_token_ids: List[int]) -> Optional[str]:
        """Feed tokens; returns the stop string that fired, if any."""
        if not self.sp.detokenize:
            return None
        for t in new_token_ids:
            self.token_ids.append(t)
            prefix = self.tok.decode(self.token_ids[self.prefix_offset:self.read_offset],
                                     skip_special_tokens=self.sp.skip_special_tokens)
            full = self.tok.decode(self.token_ids[self.prefix_offset:],
                                   skip_special_tokens=self.sp.skip_special_tokens)
            if len(full) > len(prefix) and not full.endswith("�"):
                self.text += full[len(prefix):]
                self.prefix_offset = self.read_offset
                self.read_offset = len(self.token_ids)
            if self.stop:
                hit = self._check_stop()
                if hit is not None:
                    return hit
        return None

    def _check_stop(self) -> Optional[str]:
        for s in self.stop:
            # only the tail that could contain a new match
            start = max(0, self._emitted - len(s))
            idx = self.text.find(s, start)
            if idx != -1:
                end = idx + len(s) if self.sp.include_stop_str_in_output else idx
                self.text = self.text[:end]
                return s
        return None

    def take_delta(self, finished: bool) -> str:
        """Text not yet handed out; holds back a tail that might still become
        part of a stop string."""
        end = len(self.text) if finished or not self.stop else max(self._emitted, len(self.text) - self._hold)
        delta = self.text[self._emitted:end]
        self._emitted = end
        return delta
