"""Tool-call parsers: turn model text into OpenAI ``tool_calls``.

``--tool-call-parser <name>`` selects one; ``--tool-parser-plugin <file.py>``
imports a user module that registers more (reference ``src/launch.py:38,
417-418``; the documented deployment uses ``qwen3_coder``, ``.env.server:11``).
"""
from __future__ import annotations

import importlib.util
import json
import os
import re
import uuid
from typing import Any, Callable, Dict, List, Optional, Tuple, Type


def _call_id() -> str:
    return "call_" + uuid.uuid4().hex[:24]


class ToolParser:
    """``extract(text, tools) -> (content, tool_calls)``; ``tool_calls`` are
    OpenAI-shaped dicts.  ``content`` is the text outside the calls."""

    def __init__(self, tokenizer=None):
        self.tokenizer = tokenizer

    def extract(self, text: str, tools: Optional[List[dict]] = None) -> Tuple[Optional[str], List[dict]]:
        raise NotImplementedError

    @staticmethod
    def make_call(name: str, arguments: Any) -> dict:
        if not isinstance(arguments, str):
            arguments = json.dumps(arguments, ensure_ascii=False)
        return {"id": _call_id(), "type": "function", "function": {"name": name, "arguments": arguments}}


class ToolParserManager:
    tool_parsers: Dict[str, Type[ToolParser]] = {}

    @classmethod
    def register_module(cls, name: str) -> Callable[[Type[ToolParser]], Type[ToolParser]]:
        def deco(klass: Type[ToolParser]) -> Type[ToolParser]:
            cls.tool_parsers[name] = klass
            return klass
        return deco

    @classmethod
    def get_tool_parser(cls, name: str) -> Type[ToolParser]:
        if name not in cls.tool_parsers:
            raise KeyError(f"unknown tool parser {name!r}; available: {sorted(cls.tool_parsers)}")
        return cls.tool_parsers[name]

    @classmethod
    def import_tool_parser(cls, plugin_path: str) -> None:
        name = os.path.splitext(os.path.basename(plugin_path))[0]
        spec = importlib.util.spec_from_file_location(f"vdamd_tool_plugin_{name}", plugin_path)
        if spec is None or spec.loader is None:
            raise ImportError(f"cannot import tool parser plugin {plugin_path}")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)


def _param_types(tools: Optional[List[dict]], fn_name: str) -> Dict[str, str]:
    for t in tools or []:
        f = t.get("function", t)
        if f.get("name") == fn_name:
            props = (f.get("parameters") or {}).get("properties") or {}
            return {k: (v.get("type") if isinstance(v, dict) else None) for k, v in props.items()}
    return {}


def _coerce(value: str, typ: Optional[str]) -> Any:
    v = value.strip("\n")
    if typ in (None, "string"):
        if typ is None:
            try:
                return json.loads(v)
            except (ValueError, TypeError):
                return v
        return v
    try:
        if typ == "integer":
            return int(v.strip())
        if typ == "number":
            f = float(v.strip())
            return int(f) if f.is_integer() and "." not in v else f
        if typ == "boolean":
            return v.strip().lower() in ("true", "1", "yes")
        return json.loads(v)
    except (ValueError, TypeError):
        return v


@ToolParserManager.register_module("qwen3_coder")
class Qwen3CoderToolParser(ToolParser):
    """``<tool_call><function=NAME><parameter=KEY>VALUE</parameter>...</function></tool_call>``"""

    _call = re.compile(r"<tool_call>(.*?)</tool_call>|<tool_call>(.*)$", re.S)
    _func = re.compile(r"<function=([^>\s]+)>(.*?)(?:</function>|$)", re.S)
    _param = re.compile(r"<parameter=([^>\s]+)>(.*?)(?:</parameter>|(?=<parameter=)|$)", re.S)

    def extract(self, text, tools=None):
        calls: List[dict] = []
        for m in self._call.finditer(text):
            body = m.group(1) if m.group(1) is not None else m.group(2)
            for fm in self._func.finditer(body or ""):
                name = fm.group(1)
                types = _param_types(tools, name)
                args = {pm.group(1): _coerce(pm.group(2), types.get(pm.group(1)))
                        for pm in self._param.finditer(fm.group(2))}
                calls.append(self.make_call(name, args))
        if not calls:
            return text, []
        content = self._call.sub("", text).strip()
        return (content or None), calls


@ToolParserManager.register_module("hermes")
class HermesToolParser(ToolParser):
    """``<tool_call>{"name": ..., "arguments": {...}}</tool_call>``"""

    _call = re.compile(r"<tool_call>\s*(.*?)\s*</tool_call>|<tool_call>\s*(.*)$", re.S)

    def extract(self, text, tools=None):
        calls = []
        for m in self._call.finditer(text):
            raw = m.group(1) if m.group(1) is not None else m.group(2)
            try:
                obj = json.loads(raw)
                calls.append(self.make_call(obj["name"], obj.get("arguments", obj.get("parameters", {}))))
            except (ValueError, KeyError, TypeError):
                continue
        if not calls:
            return text, []
        return (self._call.sub("", text).strip() or None), calls


@ToolParserManager.register_module("llama3_json")
class Llama3JsonToolParser(ToolParser):
    """A bare JSON object (or ``;``-separated objects) ``{"name":..,"parameters":..}``."""

    def extract(self, text, tools=None):
        s = text.strip()
        if s.startswith("<|python_tag|>"):
            s = s[len("<|python_tag|>"):].strip()
        if not s.startswith("{"):
            return text, []
        calls = []
        dec = json.JSONDecoder()
        i = 0
        try:
            while i < len(s):
                obj, end = dec.raw_decode(s, i)
                calls.append(self.make_call(obj["name"], obj.get("parameters", obj.get("arguments", {}))))
                i = end
                while i < len(s) and s[i] in " ;\n\t":
                    i += 1
        except (ValueError, KeyError, TypeError):
            return text, []
        return None, calls


def _json_call_list(parser: ToolParser, raw: str) -> List[dict]:
    """``[{"name": .., "arguments" | "parameters": ..}, ...]`` (or one such object) -> calls; [] if it is not that."""
    try:
        obj = json.loads(raw)
    except (ValueError, TypeError):
        # the list may be followed by ordinary text: take the longest JSON prefix
        try:
            obj, _ = json.JSONDecoder().raw_decode(raw.lstrip())
        except (ValueError, TypeError):
            return []
    items = obj if isinstance(obj, list) else [obj]
    calls = []
    for it in items:
        if not isinstance(it, dict) or "name" not in it:
            return []
        calls.append(parser.make_call(it["name"], it.get("arguments", it.get("parameters", {}))))
    return calls


@ToolParserManager.register_module("mistral")
class MistralToolParser(ToolParser):
    """``[TOOL_CALLS][{"name": .., "arguments": {..}}, ...]``, or (newer tokenizers) one
    ``[TOOL_CALLS]NAME{json arguments}`` per call."""

    TOKEN = "[TOOL_CALLS]"
    _named = re.compile(r"\s*([A-Za-z_][\w.\-]*)\s*(?:\[ARGS\])?\s*(\{.*)$", re.S)

    def extract(self, text, tools=None):
        if self.TOKEN not in text:
            return text, []
        content, _, rest = text.partition(self.TOKEN)
        calls: List[dict] = []
        for part in rest.split(self.TOKEN):
            part = part.strip()
            if part.startswith(("[", "{")):
                calls.extend(_json_call_list(self, part))
                continue
            m = self._named.match(part)
            if m:
                try:
                    args, _ = json.JSONDecoder().raw_decode(m.group(2))
                    calls.append(self.make_call(m.group(1), args))
                except (ValueError, TypeError):
                    pass
        if not calls:
            return text, []
        return (content.strip() or None), calls


@ToolParserManager.register_module("deepseek_v3")
class DeepSeekV3ToolParser(ToolParser):
    """``<｜tool▁calls▁begin｜><｜tool▁call▁begin｜>function<｜tool▁sep｜>NAME\n```json\n{..}\n```<｜tool▁call▁end｜>...<｜tool▁calls▁end｜>``"""

    BEGIN = "<\uff5ctool\u2581calls\u2581begin\uff5c>"
    _call = re.compile("<\uff5ctool\u2581call\u2581begin\uff5c>(?:[^<]*?)<\uff5ctool\u2581sep\uff5c>([^\n<]+)\\s*"
                       "(?:```json)?\\s*(.*?)\\s*(?:```)?\\s*<\uff5ctool\u2581call\u2581end\uff5c>", re.S)

    def extract(self, text, tools=None):
        if self.BEGIN not in text:
            return text, []
        content = text[:text.index(self.BEGIN)]
        calls = []
        for m in self._call.finditer(text):
            try:
                calls.append(self.make_call(m.group(1).strip(), json.loads(m.group(2))))
            except (ValueError, TypeError):
                continue
        if not calls:
            return text, []
        return (content.strip() or None), calls


@ToolParserManager.register_module("glm45")
class Glm45ToolParser(ToolParser):
    """``<tool_call>NAME\n<arg_key>K</arg_key>\n<arg_value>V</arg_value>...</tool_call>``; values are typed by the
    tool's parameter schema."""

    _call = re.compile(r"<tool_call>\s*([^\n<]+)\s*(.*?)</tool_call>", re.S)
    _arg = re.compile(r"<arg_key>(.*?)</arg_key>\s*<arg_value>(.*?)</arg_value>", re.S)

    def extract(self, text, tools=None):
        calls = []
        for m in self._call.finditer(text):
            name = m.group(1).strip()
            types = _param_types(tools, name)
            calls.append(self.make_call(name, {k.strip(): _coerce(v, types.get(k.strip()))
                                               for k, v in self._arg.findall(m.group(2))}))
        if not calls:
            return text, []
        return (self._call.sub("", text).strip() or None), calls


@ToolParserManager.register_module("granite")
class GraniteToolParser(ToolParser):
    """``<|tool_call|>[{"name": .., "arguments": {..}}, ...]`` (Granite 3.0) or ``<tool_call>[...]`` (3.1)."""

    def extract(self, text, tools=None):
        s = text.strip()
        for tok in ("<|tool_call|>", "<tool_call>"):
            if s.startswith(tok):
                calls = _json_call_list(self, s[len(tok):].strip())
                return (None, calls) if calls else (text, [])
        return text, []


@ToolParserManager.register_module("phi4_mini_json")
class Phi4MiniJsonToolParser(ToolParser):
    """``functools[{"name": .., "arguments": {..}}, ...]``"""

    _call = re.compile(r"functools\s*(\[.*\])", re.S)

    def extract(self, text, tools=None):
        m = self._call.search(text)
        calls = _json_call_list(self, m.group(1)) if m else []
        if not calls:
            return text, []
        return (text[:m.start()].strip() or None), calls


@ToolParserManager.register_module("pythonic")
class PythonicToolParser(ToolParser):
    """``[get_weather(city="Paris", days=3), other()]``: a Python list of calls with literal keyword
    arguments (Llama-3.2 / Llama-4 "pythonic" prompts).  Parsed with ``ast``; nothing is evaluated
    beyond literals."""

    def extract(self, text, tools=None):
        import ast
        s = text.strip()
        if s.startswith("<|python_start|>"):
            s = s[len("<|python_start|>"):].split("<|python_end|>")[0].strip()
        if not (s.startswith("[") and s.endswith("]")):
            return text, []
        try:
            tree = ast.parse(s, mode="eval").body
            if not isinstance(tree, ast.List) or not tree.elts:
                return text, []
            calls = []
            for node in tree.elts:
                if not isinstance(node, ast.Call) or node.args:
                    return text, []
                fn = node.func
                parts = []
                while isinstance(fn, ast.Attribute):
                    parts.append(fn.attr)
                    fn = fn.value
                if not isinstance(fn, ast.Name):
                    return text, []
                name = ".".join([fn.id] + parts[::-1])
                if any(kw.arg is None for kw in node.keywords):      # f(**{...}): not a literal keyword
                    return text, []
                calls.append(self.make_call(name, {kw.arg: ast.literal_eval(kw.value) for kw in node.keywords}))
        except (SyntaxError, ValueError, TypeError, RecursionError, MemoryError):
            # model output is untrusted text: whatever the parser cannot take stays plain content
            return text, []
        return None, calls


__all__ = ["ToolParser", "ToolParserManager", "Qwen3CoderToolParser", "HermesToolParser",
           "Llama3JsonToolParser", "MistralToolParser", "DeepSeekV3ToolParser", "Glm45ToolParser",
           "GraniteToolParser", "Phi4MiniJsonToolParser", "PythonicToolParser"]
