"""``/v1/responses``: the OpenAI Responses API on top of the chat pipeline.

vLLM registers this route in the app the reference serves (reference ``src/launch.py:429``).
A response is one chat turn: ``instructions`` become the system message, ``input`` is a string or
a list of messages (``input_text`` / ``output_text`` parts are flattened), tools keep their
function form.  Stored responses can be read back (``GET``) and cancelled while they stream.
"""
from __future__ import annotations

import json
import time
import uuid
from typing import Any, AsyncIterator, Dict, List, Optional, Union

from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, StreamingResponse
from pydantic import BaseModel, ConfigDict


class ResponsesRequest(BaseModel):
    model_config = ConfigDict(extra="allow")
    model: Optional[str] = None
    input: Union[str, List[Dict[str, Any]]]
    instructions: Optional[str] = None
    max_output_tokens: Optional[int] = None
    temperature: Optional[float] = None
    top_p: Optional[float] = None
    stream: bool = False
    store: bool = True
    tools: Optional[List[Dict[str, Any]]] = None
    tool_choice: Union[None, str, Dict[str, Any]] = "auto"
    text: Optional[Dict[str, Any]] = None            # {"format": {"type": "json_schema" | "json_object" | "text", ...}}
    metadata: Optional[Dict[str, Any]] = None
    previous_response_id: Optional[str] = None
    user: Optional[str] = None


def _flatten(content) -> str:
    if isinstance(content, str):
        return content
    return "".join(p.get("text", "") for p in content or [] if isinstance(p, dict))


def _messages(req: ResponsesRequest, store: Dict[str, dict]) -> List[dict]:
    msgs: List[dict] = []
    if req.previous_response_id:
        prev = store.get(req.previous_response_id)
        if prev is None:
            raise ValueError(f"previous_response_id {req.previous_response_id!r} is unknown (or was not stored)")
        msgs.extend(prev["_messages"])
    if req.instructions:
        msgs = [m for m in msgs if m["role"] != "system"]
        msgs.insert(0, {"role": "system", "content": req.instructions})
    if isinstance(req.input, str):
        msgs.append({"role": "user", "content": req.input})
    else:
        for item in req.input:
            kind = item.get("type", "message")
            if kind == "message":
                role = item.get("role", "user")
                msgs.append({"role": "system" if role == "developer" else role, "content": _flatten(item.get("content"))})
            elif kind == "function_call":
                msgs.append({"role": "assistant", "content": None, "tool_calls": [{
                    "id": item.get("call_id"), "type": "function",
                    "function": {"name": item.get("name"), "arguments": item.get("arguments", "{}")}}]})
            elif kind == "function_call_output":
                msgs.append({"role": "tool", "tool_call_id": item.get("call_id"), "content": str(item.get("output", ""))})
            else:
                raise ValueError(f"input item type {kind!r} is not supported")
    return msgs


def _chat_body(req: ResponsesRequest, msgs: List[dict]) -> dict:
    body: Dict[str, Any] = {"model": req.model, "messages": msgs, "stream": req.stream}
    if req.max_output_tokens is not None:
        body["max_completion_tokens"] = req.max_output_tokens
    for k in ("temperature", "top_p"):
        if getattr(req, k) is not None:
            body[k] = getattr(req, k)
    if req.tools:
        tools = []
        for t in req.tools:
            if t.get("type") != "function":
                raise ValueError(f"tool type {t.get('type')!r} is not supported (function tools only)")
            fn = t.get("function") or {k: t[k] for k in ("name", "description", "parameters") if k in t}
            tools.append({"type": "function", "function": fn})
        body["tools"] = tools
        tc = req.tool_choice
        if isinstance(tc, dict) and "function" not in tc and tc.get("name"):
            tc = {"type": "function", "function": {"name": tc["name"]}}
        body["tool_choice"] = tc
    fmt = (req.text or {}).get("format") or {}
    if fmt.get("type") == "json_schema":
        body["response_format"] = {"type": "json_schema", "json_schema": {
            "name": fmt.get("name", "response"), "schema": fmt.get("schema")}}
    elif fmt.get("type") == "json_object":
        body["response_format"] = {"type": "json_object"}
    for k, v in (req.model_extra or {}).items():        # sampling extras (top_k, seed, logit_bias, ...)
        body.setdefault(k, v)
    if req.stream:
        body["stream_options"] = {"include_usage": True}
    return body


def _output_items(message: dict) -> List[dict]:
    items: List[dict] = []
    if message.get("reasoning_content"):
        items.append({"type": "reasoning", "id": "rs_" + uuid.uuid4().hex, "summary": [],
                      "content": [{"type": "reasoning_text", "text": message["reasoning_content"]}]})
    if message.get("content"):
        items.append({"type": "message", "id": "msg_" + uuid.uuid4().hex, "role": "assistant", "status": "completed",
                      "content": [{"type": "output_text", "text": message["content"], "annotations": []}]})
    for tc in message.get("tool_calls") or []:
        items.append({"type": "function_call", "id": "fc_" + uuid.uuid4().hex, "call_id": tc.get("id"),
                      "name": tc["function"]["name"], "arguments": tc["function"]["arguments"],
                      "status": "completed"})
    return items


def _response_object(rid: str, req: ResponsesRequest, model: str, created: int, status: str,
                     output: List[dict], usage: Optional[dict], incomplete: Optional[str] = None) -> dict:
    u = None
    if usage:
        u = {"input_tokens": usage.get("prompt_tokens", 0), "output_tokens": usage.get("completion_tokens", 0),
             "total_tokens": usage.get("total_tokens", 0)}
    return {"id": rid, "object": "response", "created_at": created, "status": status, "model": model,
            "output": output, "usage": u, "instructions": req.instructions, "metadata": req.metadata or {},
            "max_output_tokens": req.max_output_tokens, "temperature": req.temperature, "top_p": req.top_p,
            "tools": req.tools or [], "tool_choice": req.tool_choice, "text": req.text or {"format": {"type": "text"}},
            "previous_response_id": req.previous_response_id, "parallel_tool_calls": True, "error": None,
            "incomplete_details": {"reason": incomplete} if incomplete else None,
            "output_text": "".join(c["text"] for it in output if it["type"] == "message" for c in it["content"])}


MAX_STORED = 512


def register(app: FastAPI, chat_handler, chat_request_cls, error) -> None:
    store: Dict[str, dict] = {}
    app.state.responses_store = store

    def remember(resp: dict, msgs: List[dict], assistant: dict) -> None:
        if len(store) >= MAX_STORED:
            store.pop(next(iter(store)))
        kept = dict(resp)
        kept["_messages"] = msgs + [assistant]
        store[resp["id"]] = kept

    @app.post("/v1/responses")
    async def create_response(req: ResponsesRequest, request: Request):
        st = request.app.state
        rid = "resp_" + uuid.uuid4().hex
        created = int(time.time())
        try:
            msgs = _messages(req, store)
            chat_req = chat_request_cls(**_chat_body(req, msgs))
        except ValueError as e:
            return error(str(e))
        model = req.model or st.served_names[0]
        result = await chat_handler(chat_req, request)
        if isinstance(result, JSONResponse):
            return result                                   # an error of the chat pipeline, as it is
        if not req.stream:
            choice = result["choices"][0]
            message = choice["message"]
            status = "incomplete" if choice["finish_reason"] == "length" else "completed"
            resp = _response_object(rid, req, model, created, status, _output_items(message), result.get("usage"),
                                    "max_output_tokens" if status == "incomplete" else None)
            if req.store:
                remember(resp, msgs, {k: message.get(k) for k in ("role", "content", "tool_calls") if message.get(k)
                                      or k == "role"})
            return resp

        async def events() -> AsyncIterator[str]:
            seq = 0

            def ev(kind: str, **fields) -> str:
                nonlocal seq
                payload = dict(type=kind, sequence_number=seq, **fields)
                seq += 1
                return f"event: {kind}\ndata: {json.dumps(payload, ensure_ascii=False)}\n\n"
            base = _response_object(rid, req, model, created, "in_progress", [], None)
            yield ev("response.created", response=base)
            yield ev("response.in_progress", response=base)
            item_id = "msg_" + uuid.uuid4().hex
            text, opened, usage, finish = "", False, None, None
            calls: Dict[int, dict] = {}
            async for raw in result.body_iterator:
                raw = raw.decode() if isinstance(raw, bytes) else raw
                for line in raw.split("\n\n"):
                    if not line.startswith("data: ") or line == "data: [DONE]":
                        continue
                    chunk = json.loads(line[6:])
                    if "error" in chunk:
                        yield ev("error", message=chunk["error"].get("message"), code=None)
                        return
                    if chunk.get("usage"):
                        usage = chunk["usage"]
                    for ch in chunk.get("choices") or []:
                        delta = ch.get("delta") or {}
                        piece = delta.get("content")
                        if piece:
                            if not opened:
                                opened = True
                                yield ev("response.output_item.added", output_index=0, item={
                                    "type": "message", "id": item_id, "role": "assistant", "status": "in_progress",
                                    "content": []})
                                yield ev("response.content_part.added", item_id=item_id, output_index=0,
                                         content_index=0, part={"type": "output_text", "text": "", "annotations": []})
                            text += piece
                            yield ev("response.output_text.delta", item_id=item_id, output_index=0, content_index=0,
                                     delta=piece)
                        for tc in delta.get("tool_calls") or []:
                            slot = calls.setdefault(tc.get("index", 0), {"id": None, "name": None, "arguments": ""})
                            slot["id"] = tc.get("id") or slot["id"]
                            fn = tc.get("function") or {}
                            slot["name"] = fn.get("name") or slot["name"]
                            slot["arguments"] += fn.get("arguments") or ""
                        finish = ch.get("finish_reason") or finish
            message: Dict[str, Any] = {"role": "assistant", "content": text or None}
            if calls:
                message["tool_calls"] = [{"id": c["id"], "type": "function",
                                          "function": {"name": c["name"], "arguments": c["arguments"]}}
                                         for _, c in sorted(calls.items())]
            output = _output_items(message)
            if opened:
                output[0]["id"] = item_id
                yield ev("response.output_text.done", item_id=item_id, output_index=0, content_index=0, text=text)
                yield ev("response.content_part.done", item_id=item_id, output_index=0, content_index=0,
                         part=output[0]["content"][0])
                yield ev("response.output_item.done", output_index=0, item=output[0])
            for k, item in enumerate(output[1 if opened else 0:], start=1 if opened else 0):
                yield ev("response.output_item.added", output_index=k, item=item)
                yield ev("response.output_item.done", output_index=k, item=item)
            status = "incomplete" if finish == "length" else "completed"
            final = _response_object(rid, req, model, created, status, output, usage,
                                     "max_output_tokens" if status == "incomplete" else None)
            if req.store:
                remember(final, msgs, {k: v for k, v in message.items() if v or k == "role"})
            yield ev("response.incomplete" if status == "incomplete" else "response.completed", response=final)
        return StreamingResponse(events(), media_type="text/event-stream")

    @app.get("/v1/responses/{response_id}")
    async def get_response(response_id: str):
        resp = store.get(response_id)
        if resp is None:
            return error(f"response {response_id!r} not found", 404, "NotFoundError")
        return {k: v for k, v in resp.items() if not k.startswith("_")}

    @app.delete("/v1/responses/{response_id}")
    async def delete_response(response_id: str):
        if store.pop(response_id, None) is None:
            return error(f"response {response_id!r} not found", 404, "NotFoundError")
        return {"id": response_id, "object": "response", "deleted": True}
