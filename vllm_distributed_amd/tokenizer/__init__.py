"""Tokenizers: a HuggingFace tokenizer when its files are available locally,
otherwise a self-contained byte-level tokenizer (the GPU box has no network,
so benchmark/test configurations must not depend on the hub)."""
from __future__ import annotations

import json
import re
import os
from typing import Any, Dict, List, Optional, Sequence

from ..config import ModelConfig, resolve_model_path
from ..logger import init_logger

logger = init_logger(__name__)

DEFAULT_CHAT_TEMPLATE = (
    "{% for message in messages %}"
    "<|{{ message['role'] }}|>\n{{ message['content'] }}\n"
    "{% endfor %}"
    "{% if add_generation_prompt %}<|assistant|>\n{% endif %}")


class ByteTokenizer:
    """ids 0..255 = bytes, then specials; ids beyond decode to nothing, so any
    model vocabulary can be driven with synthetic prompts."""

    def __init__(self, vocab_size: int, bos_token_id: Optional[int] = None,
                 eos_token_id: Optional[int] = None):
        self.vocab_size = vocab_size
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.chat_template = DEFAULT_CHAT_TEMPLATE
        self.is_fast = True
        self.all_special_ids = [i for i in (bos_token_id, eos_token_id) if i is not None]

    _SPELLED = re.compile(r"<(\d{3,7})>")

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        # ids past the bytes read back from the spelling convert_ids_to_tokens gives them ("<511>"): the way a
        # synthetic prompt names a special token (an image placeholder) in text
        ids: List[int] = []
        at = 0
        for m in self._SPELLED.finditer(text):
            if 256 <= int(m.group(1)) < self.vocab_size:
                ids += list(text[at:m.start()].encode("utf-8")) + [int(m.group(1))]
                at = m.end()
        ids += list(text[at:].encode("utf-8"))
        ids = [b % self.vocab_size for b in ids]
        if add_special_tokens and self.bos_token_id is not None:
            ids = [self.bos_token_id] + ids
        return ids

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        special = set(self.all_special_ids) if skip_special_tokens else set()
        bs = bytes(i for i in ids if i < 256 and i not in special)
        return bs.decode("utf-8", errors="replace")

    def convert_ids_to_tokens(self, ids: Sequence[int]) -> List[str]:
        return [chr(i) if i < 256 else f"<{i}>" for i in ids]

    def token_bytes(self) -> List[bytes]:
        """The bytes each id contributes to the text (b"" for ids that contribute none)."""
        special = set(self.all_special_ids)
        return [bytes([i]) if i < 256 and i not in special else b"" for i in range(self.vocab_size)]

    def apply_chat_template(self, messages: List[Dict[str, Any]], tokenize: bool = False,
                            add_generation_prompt: bool = True, tools=None, **kw) -> Any:
        text = render_chat_template(self.chat_template, messages, add_generation_prompt, tools)
        return self.encode(text) if tokenize else text

    def __len__(self) -> int:
        return self.vocab_size


def render_chat_template(template: str, messages, add_generation_prompt: bool = True,
                         tools=None, **extra) -> str:
    # templates come from model repositories and (when the operator allows it) from requests:
    # always rendered in the immutable sandbox, where attribute walks to Python internals,
    # mutation of passed objects and unsafe callables raise SecurityError
    from jinja2.sandbox import ImmutableSandboxedEnvironment
    env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
    env.globals["raise_exception"] = lambda m: (_ for _ in ()).throw(ValueError(m))
    env.filters["tojson"] = lambda v, **k: json.dumps(v, ensure_ascii=False)
    return env.from_string(template).render(messages=messages, tools=tools,
                                            add_generation_prompt=add_generation_prompt, **extra)


def _byte_level_decoder() -> Dict[str, int]:
    """Inverse of the byte-level BPE alphabet: printable bytes map to themselves, the rest to
    code points from U+0100 upwards in byte order."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, n = {}, 0
    for b in range(256):
        if b in keep:
            table[chr(b)] = b
        else:
            table[chr(256 + n)] = b
            n += 1
    return table


class HFTokenizer:
    def __init__(self, path: str, trust_remote_code: bool = False):
        from transformers import AutoTokenizer
        self.tok = AutoTokenizer.from_pretrained(path, trust_remote_code=trust_remote_code,
                                                 local_files_only=True)
        self.vocab_size = len(self.tok)
        self.bos_token_id = self.tok.bos_token_id
        self.eos_token_id = self.tok.eos_token_id
        self.chat_template = getattr(self.tok, "chat_template", None) or DEFAULT_CHAT_TEMPLATE
        self.all_special_ids = list(self.tok.all_special_ids)

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        return self.tok.encode(text, add_special_tokens=add_special_tokens)

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        return self.tok.decode(list(ids), skip_special_tokens=skip_special_tokens)

    def convert_ids_to_tokens(self, ids: Sequence[int]) -> List[str]:
        return self.tok.convert_ids_to_tokens(list(ids))

    def token_bytes(self) -> List[bytes]:
        """The bytes each id contributes to the text.  Byte-level BPE vocabularies (GPT-2 / Llama-3 /
        Qwen) spell bytes through a fixed printable alphabet; SentencePiece ones use U+2581 for a
        space and ``<0xNN>`` for raw bytes.  Special and added tokens get b"" (never produced by
        constrained decoding)."""
        toks = self.tok.convert_ids_to_tokens(list(range(self.vocab_size)))
        special = set(self.all_special_ids) | set(getattr(self.tok, "added_tokens_decoder", {}) or {})
        byte_level = _byte_level_decoder()
        is_bl = sum(1 for t in toks[:4096] if t and "\u0120" in t) > 8          # the byte-level alphabet's space
        out: List[bytes] = []
        for i, t in enumerate(toks):
            if t is None or i in special:
                out.append(b"")
            elif is_bl:
                try:
                    out.append(bytes(byte_level[ch] for ch in t))
                except KeyError:
                    out.append(b"")
            elif len(t) == 6 and t.startswith("<0x") and t.endswith(">"):
                out.append(bytes([int(t[3:5], 16)]))
            else:
                out.append(t.replace("\u2581", " ").encode("utf-8"))
        return out

    def apply_chat_template(self, messages, tokenize: bool = False, add_generation_prompt: bool = True,
                            tools=None, **kw) -> Any:
        if getattr(self.tok, "chat_template", None):
            return self.tok.apply_chat_template(messages, tokenize=tokenize, tools=tools,
                                                add_generation_prompt=add_generation_prompt, **kw)
        text = render_chat_template(self.chat_template, messages, add_generation_prompt, tools)
        return self.encode(text) if tokenize else text

    def __len__(self) -> int:
        return self.vocab_size


def get_tokenizer(mc: ModelConfig):
    path = resolve_model_path(mc.tokenizer or mc.model)
    if path is not None and any(os.path.isfile(os.path.join(path, f)) for f in
                                ("tokenizer.json", "tokenizer.model", "vocab.json", "tokenizer_config.json")):
        try:
            return HFTokenizer(path, mc.trust_remote_code)
        except Exception as e:  # noqa: BLE001
            logger.warning("could not load tokenizer from %s (%r); using the byte tokenizer", path, e)
    c = mc.hf_config
    eos = c.get("eos_token_id")
    if isinstance(eos, list):
        eos = eos[0]
    return ByteTokenizer(mc.vocab_size, c.get("bos_token_id"), eos)
