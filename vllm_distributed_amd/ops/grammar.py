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


# --------------------------------------------------------------------------- regular expressions
_DFA_CACHE: dict = {}
_DFA_MAX_STATES = 60000


class CompiledRegex:
    """A regular expression determinised by ``csrc/core/regex_dfa.cpp``; shared by every request
    that uses the same pattern."""

    def __init__(self, pattern: str):
        lib = load()
        P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
        if not getattr(lib, "_dfa_bound", False):
            lib.vdamd_dfa_compile.argtypes = [ctypes.c_char_p, L, I, ctypes.c_char_p, I]
            lib.vdamd_dfa_compile.restype = P
            lib.vdamd_dfa_free.argtypes = [P]
            lib.vdamd_dfa_free.restype = None
            lib.vdamd_dfa_num_states.argtypes = [P]
            lib.vdamd_dfa_num_states.restype = I
            lib.vdamd_dfa_step.argtypes = [P, I, ctypes.c_char_p, L]
            lib.vdamd_dfa_step.restype = I
            lib.vdamd_dfa_flags.argtypes = [P, I]
            lib.vdamd_dfa_flags.restype = I
            lib.vdamd_dfa_mask.argtypes = [P, I, P, P, L, P]
            lib.vdamd_dfa_mask.restype = L
            lib._dfa_bound = True
        self.lib = lib
        raw = pattern.encode("utf-8")
        err = ctypes.create_string_buffer(256)
        self.handle = lib.vdamd_dfa_compile(raw, len(raw), _DFA_MAX_STATES, err, len(err))
        if not self.handle:
            raise ValueError(f"cannot compile guided-decoding pattern: {err.value.decode('utf-8', 'replace')}")
        self.pattern = pattern
        self.num_states = lib.vdamd_dfa_num_states(self.handle)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.vdamd_dfa_free(h)


def compile_regex(pattern: str) -> CompiledRegex:
    c = _DFA_CACHE.get(pattern)
    if c is None:
        c = CompiledRegex(pattern)
        if len(_DFA_CACHE) >= 64:
            _DFA_CACHE.pop(next(iter(_DFA_CACHE)))
        _DFA_CACHE[pattern] = c
    return c


class RegexGrammar:
    """One request's position in a compiled pattern (vLLM's ``guided_regex``; ``guided_json`` and
    ``response_format: json_schema`` arrive here lowered by ``json_schema_to_regex``)."""

    def __init__(self, pattern: str) -> None:
        self.rx = compile_regex(pattern)
        self.state = 0

    def advance(self, data: bytes) -> bool:
        s = self.rx.lib.vdamd_dfa_step(self.rx.handle, self.state, data, len(data))
        if s < 0:
            return False
        self.state = s
        return True

    @property
    def may_stop(self) -> bool:
        return bool(self.rx.lib.vdamd_dfa_flags(self.rx.handle, self.state) & 1)

    @property
    def done(self) -> bool:
        return self.rx.lib.vdamd_dfa_flags(self.rx.handle, self.state) == 1

    def mask(self, table: TokenTable) -> np.ndarray:
        key = (id(self.rx), self.state)
        out = table.mask_cache.get(key)
        if out is None:
            out = np.empty(table.n, dtype=np.uint8)
            self.rx.lib.vdamd_dfa_mask(self.rx.handle, self.state, table.blob.ctypes.data,
                                       table.offsets.ctypes.data, table.n, out.ctypes.data)
            if len(table.mask_cache) >= 512:
                table.mask_cache.clear()
            table.mask_cache[key] = out
        return out


# --------------------------------------------------------------------------- context-free grammars
_CFG_CACHE: dict = {}


class CompiledGrammar:
    """An EBNF grammar (GBNF spelling, start rule ``root``) compiled by ``csrc/core/cfg_grammar.cpp``."""

    def __init__(self, text: str):
        lib = load()
        P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
        if not getattr(lib, "_cfg_bound", False):
            lib.vdamd_cfg_compile.argtypes = [ctypes.c_char_p, L, ctypes.c_char_p, I]
            lib.vdamd_cfg_compile.restype = P
            lib.vdamd_cfg_free.argtypes = [P]
            lib.vdamd_cfg_free.restype = None
            lib.vdamd_cfg_state_new.argtypes = [P]
            lib.vdamd_cfg_state_new.restype = P
            lib.vdamd_cfg_state_free.argtypes = [P]
            lib.vdamd_cfg_state_free.restype = None
            lib.vdamd_cfg_advance.argtypes = [P, ctypes.c_char_p, L]
            lib.vdamd_cfg_advance.restype = I
            lib.vdamd_cfg_flags.argtypes = [P]
            lib.vdamd_cfg_flags.restype = I
            lib.vdamd_cfg_state_key.argtypes = [P, ctypes.c_char_p, L]
            lib.vdamd_cfg_state_key.restype = L
            lib.vdamd_trie_build.argtypes = [P, P, L]
            lib.vdamd_trie_build.restype = P
            lib.vdamd_trie_free.argtypes = [P]
            lib.vdamd_trie_free.restype = None
            lib.vdamd_cfg_mask.argtypes = [P, P, P]
            lib.vdamd_cfg_mask.restype = L
            lib._cfg_bound = True
        self.lib = lib
        raw = text.encode("utf-8")
        err = ctypes.create_string_buffer(256)
        self.handle = lib.vdamd_cfg_compile(raw, len(raw), err, len(err))
        if not self.handle:
            raise ValueError(f"cannot compile guided-decoding grammar: {err.value.decode('utf-8', 'replace')}")
        self.text = text

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.vdamd_cfg_free(h)


def compile_grammar(text: str) -> CompiledGrammar:
    c = _CFG_CACHE.get(text)
    if c is None:
        c = CompiledGrammar(text)
        if len(_CFG_CACHE) >= 64:
            _CFG_CACHE.pop(next(iter(_CFG_CACHE)))
        _CFG_CACHE[text] = c
    return c


class _Trie:
    """The vocabulary's byte strings as a trie (built once per token table, walked by every mask)."""

    def __init__(self, lib, table: "TokenTable"):
        self.lib = lib
        self.handle = lib.vdamd_trie_build(table.blob.ctypes.data, table.offsets.ctypes.data, table.n)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.vdamd_trie_free(h)


class CfgGrammar:
    """One request's position in a context-free grammar (vLLM's ``guided_grammar``): the set of parse
    stacks the text so far leaves open."""

    def __init__(self, text: str) -> None:
        self.cg = compile_grammar(text)
        self.state = self.cg.lib.vdamd_cfg_state_new(self.cg.handle)

    def __del__(self):
        s, self.state = getattr(self, "state", None), None
        if s:
            self.cg.lib.vdamd_cfg_state_free(s)

    def advance(self, data: bytes) -> bool:
        return self.cg.lib.vdamd_cfg_advance(self.state, data, len(data)) == 0

    @property
    def may_stop(self) -> bool:
        return bool(self.cg.lib.vdamd_cfg_flags(self.state) & 1)

    @property
    def done(self) -> bool:
        return self.cg.lib.vdamd_cfg_flags(self.state) == 1

    def _key(self) -> bytes:
        lib = self.cg.lib
        n = lib.vdamd_cfg_state_key(self.state, None, 0)
        buf = ctypes.create_string_buffer(max(1, n))
        lib.vdamd_cfg_state_key(self.state, buf, n)
        return buf.raw[:n]

    def mask(self, table: TokenTable) -> np.ndarray:
        key = (id(self.cg), self._key())
        out = table.mask_cache.get(key)
        if out is None:
            trie = getattr(table, "_trie", None)
            if trie is None:
                trie = table._trie = _Trie(self.cg.lib, table)
            out = np.empty(table.n, dtype=np.uint8)
            self.cg.lib.vdamd_cfg_mask(self.state, trie.handle, out.ctypes.data)
            if len(table.mask_cache) >= 512:
                table.mask_cache.clear()
            table.mask_cache[key] = out
        return out


# --------------------------------------------------------------------------- JSON schema -> regex
_WS = "[ ]?"
_STR_CHAR = r'([^"\\\x00-\x1f]|\\["\\/bfnrt]|\\u[0-9a-fA-F]{4})'
_STRING = '"' + _STR_CHAR + '*"'
_INTEGER = r"-?(0|[1-9][0-9]{0,17})"
_NUMBER = _INTEGER + r"(\.[0-9]{1,17})?([eE][+-]?[0-9]{1,3})?"
_BOOLEAN = "(true|false)"
_NULL = "null"
_FORMATS = {
    "date-time": r'"[0-9]{4}-[0-9]{2}-[0-9]{2}T[0-9]{2}:[0-9]{2}:[0-9]{2}(\.[0-9]{1,6})?(Z|[+-][0-9]{2}:[0-9]{2})"',
    "date": r'"[0-9]{4}-[0-9]{2}-[0-9]{2}"',
    "time": r'"[0-9]{2}:[0-9]{2}:[0-9]{2}(\.[0-9]{1,6})?(Z|[+-][0-9]{2}:[0-9]{2})?"',
    "uuid": r'"[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{12}"',
}
_MAX_REGEX = 400_000


def _rx_escape(text: str) -> str:
    out = []
    for ch in text:
        if ch in r"\.[]{}()*+?|^$-/":
            out.append("\\" + ch)
        elif ch == "\n":
            out.append(r"\n")
        elif ch == "\t":
            out.append(r"\t")
        elif ch == "\r":
            out.append(r"\r")
        elif ord(ch) < 0x20:
            out.append("\\x%02x" % ord(ch))
        else:
            out.append(ch)
    return "".join(out)


def _json_literal(value) -> str:
    import json
    return _rx_escape(json.dumps(value, ensure_ascii=False, separators=(",", ":")))


def _any_json(depth: int) -> str:
    scalar = f"({_STRING}|{_NUMBER}|{_BOOLEAN}|{_NULL})"
    if depth <= 0:
        return scalar
    inner = _any_json(depth - 1)
    arr = rf"\[{_WS}({inner}({_WS},{_WS}{inner})*)?{_WS}\]"
    obj = rf"\{{{_WS}({_STRING}{_WS}:{_WS}{inner}({_WS},{_WS}{_STRING}{_WS}:{_WS}{inner})*)?{_WS}\}}"
    return f"({_STRING}|{_NUMBER}|{_BOOLEAN}|{_NULL}|{arr}|{obj})"


def json_schema_to_regex(schema, root=None, depth: int = 0) -> str:
    """A regular expression for the compact JSON texts that satisfy ``schema``.

    Objects list their ``properties`` in schema order; those not named in ``required`` may be
    left out.  Supported: ``type`` (also a list of types), ``enum``, ``const``, ``anyOf`` /
    ``oneOf``, ``allOf`` with one entry, local ``$ref``, string ``minLength`` / ``maxLength`` /
    ``pattern`` / ``format`` (date-time, date, time, uuid), integer / number / boolean / null,
    arrays with ``items`` / ``minItems`` / ``maxItems``, nested objects.  A schema without a type
    admits any JSON value of limited nesting.  Recursive references are unrolled three levels."""
    import json
    if isinstance(schema, str):
        schema = json.loads(schema)
    if schema is True or schema == {}:
        return _any_json(2)
    if not isinstance(schema, dict):
        raise ValueError("a JSON schema must be an object")
    root = schema if root is None else root
    if depth > 12:
        raise ValueError("JSON schema nests (or recurses) too deeply for guided decoding")
    if "$ref" in schema:
        ref = schema["$ref"]
        if not ref.startswith("#/"):
            raise ValueError(f"only local $ref is supported, got {ref!r}")
        node = root
        for part in ref[2:].split("/"):
            part = part.replace("~1", "/").replace("~0", "~")
            if not isinstance(node, dict) or part not in node:
                raise ValueError(f"unresolvable $ref {ref!r}")
            node = node[part]
        return json_schema_to_regex(node, root, depth + 4)
    if "const" in schema:
        return _json_literal(schema["const"])
    if "enum" in schema:
        if not schema["enum"]:
            raise ValueError("empty enum")
        return "(" + "|".join(_json_literal(v) for v in schema["enum"]) + ")"
    for key in ("anyOf", "oneOf"):
        if key in schema:
            return "(" + "|".join(json_schema_to_regex(s, root, depth + 1) for s in schema[key]) + ")"
    if "allOf" in schema:
        if len(schema["allOf"]) != 1:
            raise ValueError("allOf with more than one entry is not supported")
        return json_schema_to_regex(schema["allOf"][0], root, depth + 1)
    t = schema.get("type")
    if isinstance(t, list):
        return "(" + "|".join(json_schema_to_regex({**schema, "type": one}, root, depth + 1) for one in t) + ")"
    if t is None:
        if "properties" in schema:
            t = "object"
        elif "items" in schema:
            t = "array"
        else:
            return _any_json(2)
    if t == "string":
        if "format" in schema and schema["format"] in _FORMATS:
            return _FORMATS[schema["format"]]
        if "pattern" in schema:
            pat = schema["pattern"]
            pat = pat[1:] if pat.startswith("^") else pat
            pat = pat[:-1] if pat.endswith("$") and not pat.endswith("\\$") else pat
            return '"(' + pat + ')"'
        lo, hi = int(schema.get("minLength", 0)), schema.get("maxLength")
        if lo == 0 and hi is None:
            return _STRING
        return '"' + _STR_CHAR + "{" + str(lo) + "," + ("" if hi is None else str(int(hi))) + '}"'
    if t == "integer":
        return _INTEGER
    if t == "number":
        return _NUMBER
    if t == "boolean":
        return _BOOLEAN
    if t == "null":
        return _NULL
    if t == "array":
        item = json_schema_to_regex(schema.get("items", {}), root, depth + 1)
        lo, hi = int(schema.get("minItems", 0)), schema.get("maxItems")
        if hi is not None and int(hi) == 0:
            return rf"\[{_WS}\]"
        more = "{" + str(max(lo - 1, 0)) + "," + ("" if hi is None else str(int(hi) - 1)) + "}"
        body = f"{item}({_WS},{_WS}{item}){more}"
        if lo == 0:
            body = f"({body})?"
        return rf"\[{_WS}{body}{_WS}\]"
    if t == "object":
        props = schema.get("properties") or {}
        if not props:
            return _any_json(1) if schema.get("additionalProperties", True) is not False else rf"\{{{_WS}\}}"
        required = set(schema.get("required") or [])
        names = list(props)
        pieces = [f'"{_rx_escape(json.dumps(n, ensure_ascii=False)[1:-1])}"{_WS}:{_WS}'
                  + json_schema_to_regex(props[n], root, depth + 1) for n in names]

        def seq(i: int, need_comma: bool) -> str:
            """members i.. given whether one was already written"""
            if i == len(names):
                return ""
            lead = f"{_WS},{_WS}" if need_comma else ""
            here = lead + pieces[i] + seq(i + 1, True)
            if names[i] in required:
                return here
            rest = seq(i + 1, need_comma)
            if len(here) + len(rest) > _MAX_REGEX:
                raise ValueError("JSON schema has too many optional properties for guided decoding")
            return f"({here}|{rest})" if rest else f"({here})?"

        return rf"\{{{_WS}{seq(0, False)}{_WS}\}}"
    raise ValueError(f"unsupported JSON schema type {t!r}")
