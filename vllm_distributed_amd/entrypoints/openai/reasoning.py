"""Reasoning parsers: split a chat completion into ``reasoning_content`` (the
model's thinking between ``<think>`` and ``</think>``) and ``content``.

vLLM's ``--reasoning-parser`` option, which the reference exposes with the rest
of vLLM's ``serve`` flags (reference ``src/launch.py:465-491``).  The splitter is
incremental so streaming responses can route each delta to the right field:
text that might still turn out to be (part of) a tag is held back until it is
decided.
"""
from __future__ import annotations

from typing import Dict, Tuple, Type


class ReasoningSplitter:
    start_tag = "<think>"
    end_tag = "</think>"
    # True: the output is reasoning until the end tag even without a start tag (templates that
    # open the think block in the prompt); False: reasoning only when the output opens the block
    starts_in_reasoning = False

    def __init__(self) -> None:
        self.state = "start"
        self.buf = ""

    def feed(self, text: str) -> Tuple[str, str]:
        """Returns (reasoning delta, content delta) decided by this piece of text."""
        self.buf += text
        reasoning, content = "", ""
        while True:
            if self.state == "start":
                stripped = self.buf.lstrip()
                if stripped.startswith(self.start_tag):
                    self.state, self.buf = "reasoning", stripped[len(self.start_tag):]
                elif self.start_tag.startswith(stripped):
                    break                                   # nothing yet, or a tag still being spelled
                else:
                    self.state = "reasoning" if self.starts_in_reasoning else "content"
            elif self.state == "reasoning":
                i = self.buf.find(self.end_tag)
                if i >= 0:
                    reasoning += self.buf[:i]
                    self.state, self.buf = "content", self.buf[i + len(self.end_tag):]
                    continue
                keep = 0                                    # a tail that may be the start of the end tag stays held
                for k in range(min(len(self.end_tag) - 1, len(self.buf)), 0, -1):
                    if self.end_tag.startswith(self.buf[-k:]):
                        keep = k
                        break
                reasoning += self.buf[:len(self.buf) - keep]
                self.buf = self.buf[len(self.buf) - keep:]
                break
            else:
                content += self.buf
                self.buf = ""
                break
        return reasoning, content

    def flush(self) -> Tuple[str, str]:
        """What was still held when the generation ended."""
        held, self.buf = self.buf, ""
        if self.state == "reasoning" or (self.state == "start" and self.starts_in_reasoning):
            return held, ""
        return "", held

    @classmethod
    def split(cls, text: str) -> Tuple[str, str]:
        s = cls()
        r, c = s.feed(text)
        r2, c2 = s.flush()
        return r + r2, c + c2


class Qwen3ReasoningSplitter(ReasoningSplitter):
    """Reasoning only when the output itself opens ``<think>``."""


class DeepSeekR1ReasoningSplitter(ReasoningSplitter):
    """Everything up to ``</think>`` is reasoning, with or without the opening tag."""
    starts_in_reasoning = True


class ReasoningParserManager:
    parsers: Dict[str, Type[ReasoningSplitter]] = {
        "qwen3": Qwen3ReasoningSplitter, "deepseek_r1": DeepSeekR1ReasoningSplitter}

    @classmethod
    def get(cls, name: str) -> Type[ReasoningSplitter]:
        if name not in cls.parsers:
            raise KeyError(f"unknown reasoning parser {name!r}; available: {sorted(cls.parsers)}")
        return cls.parsers[name]
