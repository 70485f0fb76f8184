"""``chat`` and ``complete``: terminal clients for a running server.

The reference's head CLI registers the sub-commands of vLLM's ``openai`` CLI
module next to ``serve`` (reference ``src/launch.py:465-471``): an interactive
chat and a completion prompt that talk to ``/v1/chat/completions`` and
``/v1/completions`` of a server started elsewhere.  Streaming responses are
printed token by token as the SSE events arrive.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Dict, Iterator, List, Optional

import httpx


def add_client_args(p: argparse.ArgumentParser) -> None:
    p.add_argument("--url", type=str, default="http://localhost:8000/v1", help="base URL of the running server")
    p.add_argument("--model-name", type=str, default=None, help="served model name (default: first of /v1/models)")
    p.add_argument("--api-key", type=str, default=None, help="bearer token (default: $OPENAI_API_KEY)")
    p.add_argument("--max-tokens", type=int, default=256)
    p.add_argument("--temperature", type=float, default=0.7)
    p.add_argument("-q", "--quick", type=str, default=None, metavar="TEXT",
                   help="send one message / prompt, print the answer and exit")


class ServerClient:
    def __init__(self, url: str, api_key: Optional[str] = None, model: Optional[str] = None,
                 transport: Optional[httpx.BaseTransport] = None, timeout: float = 600.0):
        key = api_key or os.environ.get("OPENAI_API_KEY")
        headers = {"Authorization": f"Bearer {key}"} if key else {}
        self.http = httpx.Client(base_url=url.rstrip("/"), headers=headers, transport=transport, timeout=timeout)
        self.model = model or self._first_model()

    def _first_model(self) -> str:
        r = self.http.get("/models")
        r.raise_for_status()
        return r.json()["data"][0]["id"]

    def _stream(self, path: str, body: Dict) -> Iterator[Dict]:
        with self.http.stream("POST", path, json=dict(body, model=self.model, stream=True)) as r:
            if r.status_code != 200:
                r.read()
                raise RuntimeError(f"{path}: HTTP {r.status_code}: {r.text}")
            for line in r.iter_lines():
                if not line.startswith("data:"):
                    continue
                data = line[len("data:"):].strip()
                if data == "[DONE]":
                    return
                yield json.loads(data)

    def complete(self, prompt: str, **params) -> Iterator[str]:
        for ev in self._stream("/completions", dict(params, prompt=prompt)):
            for c in ev.get("choices", []):
                if c.get("text"):
                    yield c["text"]

    def chat(self, messages: List[Dict], **params) -> Iterator[str]:
        for ev in self._stream("/chat/completions", dict(params, messages=messages)):
            for c in ev.get("choices", []):
                piece = (c.get("delta") or {}).get("content")
                if piece:
                    yield piece

    def close(self) -> None:
        self.http.close()


def _params(args: argparse.Namespace) -> Dict:
    return {"max_tokens": args.max_tokens, "temperature": args.temperature}


def _emit(pieces: Iterator[str], out) -> str:
    text = []
    for p in pieces:
        text.append(p)
        out.write(p)
        out.flush()
    out.write("\n")
    return "".join(text)


def cmd_complete(args: argparse.Namespace, transport: Optional[httpx.BaseTransport] = None, out=None) -> None:
    out = out or sys.stdout
    client = ServerClient(args.url, args.api_key, args.model_name, transport)
    try:
        if args.quick is not None:
            _emit(client.complete(args.quick, **_params(args)), out)
            return
        out.write(f"Using model: {client.model}. Enter a prompt (Ctrl-D to quit).\n")
        while True:
            try:
                prompt = input("> ")
            except EOFError:
                return
            _emit(client.complete(prompt, **_params(args)), out)
    finally:
        client.close()


def cmd_chat(args: argparse.Namespace, transport: Optional[httpx.BaseTransport] = None, out=None) -> None:
    out = out or sys.stdout
    client = ServerClient(args.url, args.api_key, args.model_name, transport)
    messages: List[Dict] = []
    if getattr(args, "system_prompt", None):
        messages.append({"role": "system", "content": args.system_prompt})
    try:
        if args.quick is not None:
            messages.append({"role": "user", "content": args.quick})
            _emit(client.chat(messages, **_params(args)), out)
            return
        out.write(f"Using model: {client.model}. Type a message (Ctrl-D to quit).\n")
        while True:
            try:
                msg = input("> ")
            except EOFError:
                return
            messages.append({"role": "user", "content": msg})
            answer = _emit(client.chat(messages, **_params(args)), out)
            messages.append({"role": "assistant", "content": answer})
    finally:
        client.close()
