"""Embedding, pooling and scoring routes.

The reference serves whatever routes vLLM's ``build_app`` registers (reference
``src/launch.py:429``); for a generative checkpoint those include ``/v1/embeddings``,
``/pooling`` and ``/score``.  Here a causal LM embeds a text as the final hidden state of its
last token (or the mean over the prompt), computed by an ordinary prefill whose request
carries ``SamplingParams.pooling``: such requests batch and chunk with everything else.
"""
from __future__ import annotations

import asyncio
import base64
import struct
import time
import uuid
from typing import Any, Dict, List, Optional, Union

from fastapi import FastAPI, Request
from pydantic import BaseModel, ConfigDict

from ...sampling_params import SamplingParams


class EmbeddingRequest(BaseModel):
    model_config = ConfigDict(extra="allow")
    model: Optional[str] = None
    input: Union[str, List[str], List[int], List[List[int]]]
    encoding_format: str = "float"
    dimensions: Optional[int] = None
    user: Optional[str] = None
    truncate_prompt_tokens: Optional[int] = None
    priority: int = 0
    normalize: Optional[bool] = None
    pooling_type: Optional[str] = None          # "last" (default) | "mean"


class ScoreRequest(BaseModel):
    model_config = ConfigDict(extra="allow")
    model: Optional[str] = None
    text_1: Union[str, List[str]]
    text_2: Union[str, List[str]]
    truncate_prompt_tokens: Optional[int] = None
    priority: int = 0


def _as_prompts(inp) -> List[Any]:
    if isinstance(inp, str):
        return [inp]
    if not inp:
        raise ValueError("empty input")
    if isinstance(inp[0], int):
        return [{"prompt_token_ids": list(inp)}]
    if isinstance(inp[0], list):
        return [{"prompt_token_ids": list(t)} for t in inp]
    return list(inp)


def _norm(v: List[float]) -> List[float]:
    n = sum(x * x for x in v) ** 0.5
    return [x / n for x in v] if n > 0 else v


async def _pool(st, prompts: List[Any], pooling: str, truncate: Optional[int], priority: int, model: Optional[str],
                rid: str):
    """(vectors, prompt token counts) of ``prompts``, all in flight together."""
    sp = SamplingParams(pooling=pooling, temperature=0.0, detokenize=False, truncate_prompt_tokens=truncate)
    lora = st.loras.get(model) if model else None

    async def one(i: int, p):
        final = None
        async for out in st.engine.generate(p, sp, f"{rid}-{i}", priority, lora):
            final = out
        if final is None or final.pooled is None:
            raise RuntimeError("the engine returned no pooled state")
        return final.pooled, len(final.prompt_token_ids)
    got = await asyncio.gather(*[one(i, p) for i, p in enumerate(prompts)])
    return [g[0] for g in got], [g[1] for g in got]


def register(app: FastAPI, error, check_model) -> None:
    async def _embed(req: EmbeddingRequest, request: Request, raw: bool):
        st = request.app.state
        err = check_model(st, req.model)
        if err:
            return err
        pooling = req.pooling_type or "last"
        if pooling not in ("last", "mean"):
            return error("pooling_type must be 'last' or 'mean'")
        if req.encoding_format not in ("float", "base64"):
            return error("encoding_format must be 'float' or 'base64'")
        try:
            prompts = _as_prompts(req.input)
        except ValueError as e:
            return error(str(e))
        rid = ("pool-" if raw else "embd-") + uuid.uuid4().hex
        try:
            vecs, counts = await _pool(st, prompts, pooling, req.truncate_prompt_tokens, req.priority, req.model, rid)
        except ValueError as e:
            return error(str(e))
        normalize = (not raw) if req.normalize is None else req.normalize
        data = []
        for i, v in enumerate(vecs):
            if req.dimensions:
                if req.dimensions < 1 or req.dimensions > len(v):
                    return error(f"dimensions must be between 1 and {len(v)}")
                v = v[:req.dimensions]             # matryoshka-style truncation, then re-normalised
            if normalize:
                v = _norm(v)
            payload: Any = v
            if req.encoding_format == "base64":
                payload = base64.b64encode(struct.pack(f"<{len(v)}f", *v)).decode("ascii")
            data.append({"object": "pooling", "index": i, "data": payload} if raw else
                        {"object": "embedding", "index": i, "embedding": payload})
        n = sum(counts)
        return {"id": rid, "object": "list", "created": int(time.time()),
                "model": req.model or st.served_names[0], "data": data,
                "usage": {"prompt_tokens": n, "total_tokens": n, "completion_tokens": 0}}

    @app.post("/v1/embeddings")
    async def embeddings(req: EmbeddingRequest, request: Request):
        return await _embed(req, request, raw=False)

    @app.post("/pooling")
    async def pooling(req: EmbeddingRequest, request: Request):
        return await _embed(req, request, raw=True)

    async def _score(req: ScoreRequest, request: Request):
        st = request.app.state
        err = check_model(st, req.model)
        if err:
            return err
        a = [req.text_1] if isinstance(req.text_1, str) else list(req.text_1)
        b = [req.text_2] if isinstance(req.text_2, str) else list(req.text_2)
        if not a or not b:
            return error("text_1 and text_2 must not be empty")
        if len(a) == 1:
            a = a * len(b)
        if len(a) != len(b):
            return error("text_1 must hold one entry, or as many as text_2")
        rid = "score-" + uuid.uuid4().hex
        uniq: Dict[str, int] = {}
        for t in a + b:                           # one forward per distinct text
            uniq.setdefault(t, len(uniq))
        try:
            vecs, counts = await _pool(st, list(uniq), "last", req.truncate_prompt_tokens, req.priority, req.model,
                                       rid)
        except ValueError as e:
            return error(str(e))
        vecs = [_norm(v) for v in vecs]
        data = [{"object": "score", "index": i,
                 "score": float(sum(x * y for x, y in zip(vecs[uniq[a[i]]], vecs[uniq[b[i]]])))}
                for i in range(len(a))]
        n = sum(counts)
        return {"id": rid, "object": "list", "created": int(time.time()),
                "model": req.model or st.served_names[0], "data": data,
                "usage": {"prompt_tokens": n, "total_tokens": n, "completion_tokens": 0}}

    @app.post("/score")
    async def score(req: ScoreRequest, request: Request):
        return await _score(req, request)

    @app.post("/v1/score")
    async def score_v1(req: ScoreRequest, request: Request):
        return await _score(req, request)
