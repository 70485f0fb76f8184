"""OpenAI-compatible HTTP server.

Bring-up order follows the reference's ``run_server`` (reference
``src/launch.py:413-457``): bind the listening socket first, import a
tool-parser plugin, load the uvicorn log config, build the engine client,
build the app, initialise app state, serve; close the socket after shutdown.

Endpoints: ``/health`` ``/ping`` ``/version`` ``/v1/models``
``/v1/completions`` ``/v1/chat/completions`` (SSE streaming, tool calls)
``/tokenize`` ``/detokenize`` ``/metrics`` ``/get_tokenizer_info``
``/start_profile`` ``/stop_profile`` ``/reset_prefix_cache`` ``/sleep``
``/wake_up``.  ``--api-key`` (or ``VLLM_API_KEY``) enforces a bearer token on
``/v1/*``.
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import socket
import time
import uuid
from contextlib import asynccontextmanager
from typing import Any, AsyncIterator, Dict, List, Optional, Tuple, Union

from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, Response, StreamingResponse
from pydantic import BaseModel, ConfigDict

from ... import envs
from ...engine.arg_utils import EngineArgs
from ...engine.async_llm import AsyncLLM
from ...engine.core import EngineDeadError
from ...logger import init_logger
from ...sampling_params import SamplingParams
from ...engine.beam_search import BeamSearchParams, beam_search_async
from ...version import __version__
from .tool_parsers import ToolParserManager

logger = init_logger("entrypoints.openai.api_server")


# ------------------------------------------------------------------ protocol
class _Base(BaseModel):
    model_config = ConfigDict(extra="allow")


class StreamOptions(_Base):
    include_usage: Optional[bool] = False


class CompletionRequest(_Base):
    model: Optional[str] = None
    prompt: Union[str, List[str], List[int], List[List[int]]]
    max_tokens: Optional[int] = 16
    temperature: Optional[float] = 1.0
    top_p: Optional[float] = 1.0
    top_k: Optional[int] = -1
    min_p: Optional[float] = 0.0
    n: int = 1
    best_of: Optional[int] = None          # sample this many, return the n most probable (not streamable)
    use_beam_search: bool = False          # n = beam width; deterministic (not streamable)
    length_penalty: float = 1.0
    stream: bool = False
    stream_options: Optional[StreamOptions] = None
    stop: Union[None, str, List[str]] = None
    stop_token_ids: Optional[List[int]] = None
    seed: Optional[int] = None
    logprobs: Optional[int] = None
    echo: bool = False
    presence_penalty: Optional[float] = 0.0
    frequency_penalty: Optional[float] = 0.0
    repetition_penalty: Optional[float] = 1.0
    ignore_eos: bool = False
    min_tokens: int = 0
    skip_special_tokens: bool = True
    logit_bias: Optional[Dict[str, float]] = None
    allowed_token_ids: Optional[List[int]] = None
    bad_words: Optional[List[str]] = None
    truncate_prompt_tokens: Optional[int] = None
    prompt_logprobs: Optional[int] = None
    response_format: Optional[Dict[str, Any]] = None
    priority: int = 0


class ChatMessage(_Base):
    role: str
    content: Union[None, str, List[Dict[str, Any]]] = None
    tool_calls: Optional[List[Dict[str, Any]]] = None
    tool_call_id: Optional[str] = None
    name: Optional[str] = None


class ChatCompletionRequest(_Base):
    model: Optional[str] = None
    messages: List[ChatMessage]
    max_tokens: Optional[int] = None
    max_completion_tokens: Optional[int] = None
    temperature: Optional[float] = 1.0
    top_p: Optional[float] = 1.0
    top_k: Optional[int] = -1
    min_p: Optional[float] = 0.0
    n: int = 1
    use_beam_search: bool = False          # n = beam width; deterministic (not streamable)
    length_penalty: float = 1.0
    stream: bool = False
    stream_options: Optional[StreamOptions] = None
    stop: Union[None, str, List[str]] = None
    stop_token_ids: Optional[List[int]] = None
    seed: Optional[int] = None
    logprobs: Optional[bool] = False
    top_logprobs: Optional[int] = None
    presence_penalty: Optional[float] = 0.0
    frequency_penalty: Optional[float] = 0.0
    repetition_penalty: Optional[float] = 1.0
    tools: Optional[List[Dict[str, Any]]] = None
    tool_choice: Union[None, str, Dict[str, Any]] = "auto"
    ignore_eos: bool = False
    min_tokens: int = 0
    skip_special_tokens: bool = True
    add_generation_prompt: bool = True
    chat_template: Optional[str] = None
    chat_template_kwargs: Optional[Dict[str, Any]] = None
    logit_bias: Optional[Dict[str, float]] = None
    allowed_token_ids: Optional[List[int]] = None
    bad_words: Optional[List[str]] = None
    truncate_prompt_tokens: Optional[int] = None
    prompt_logprobs: Optional[int] = None
    response_format: Optional[Dict[str, Any]] = None
    priority: int = 0


class TokenizeRequest(_Base):
    model: Optional[str] = None
    prompt: Optional[str] = None
    messages: Optional[List[ChatMessage]] = None
    add_special_tokens: bool = True
    add_generation_prompt: bool = True


class DetokenizeRequest(_Base):
    model: Optional[str] = None
    tokens: List[int]


def _error(message: str, status: int = 400, etype: str = "BadRequestError") -> JSONResponse:
    return JSONResponse({"object": "error", "message": message, "type": etype, "param": None,
                         "code": status}, status_code=status)


# ------------------------------------------------------------- server set-up
def setup_server(args: argparse.Namespace) -> Tuple[str, socket.socket]:
    """Bind the HTTP socket before the (slow) engine start so a port conflict
    fails immediately."""
    host = args.host or "0.0.0.0"
    family = socket.AF_INET6 if ":" in host else socket.AF_INET
    sock = socket.socket(family=family, type=socket.SOCK_STREAM)
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.bind((host, args.port))
    scheme = "https" if getattr(args, "ssl_certfile", None) else "http"
    return f"{scheme}://{host}:{sock.getsockname()[1]}", sock


MAX_SAMPLES_PER_PROMPT = 128      # n / best_of: each sample is a request sharing the prompt's KV blocks
MAX_BEAM_WIDTH = 32


def _nth_params(sp: SamplingParams, j: int) -> SamplingParams:
    """Sampling parameters of the ``j``-th of ``n`` samples of one prompt."""
    out = sp.clone()
    out.n = 1
    if out.seed is not None:
        out.seed = (out.seed + j) % (2 ** 63)
    return out


def load_log_config(path: Optional[str]) -> Optional[dict]:
    if not path:
        return None
    try:
        with open(path) as f:
            return json.load(f)
    except Exception as e:  # noqa: BLE001
        logger.warning("failed to load log config %s: %r", path, e)
        return None


@asynccontextmanager
async def build_async_engine_client(args: argparse.Namespace, client_config: Optional[dict] = None,
                                    ) -> AsyncIterator[AsyncLLM]:
    """Engine client with the distributed executor injected whenever more than
    one rank is needed (reference ``src/launch.py:391-410``)."""
    engine_args = EngineArgs.from_cli_args(args)
    if engine_args.distributed_executor_backend in (None, "external_launcher") and \
            engine_args.tensor_parallel_size * engine_args.pipeline_parallel_size > 1:
        engine_args.distributed_executor_backend = "distributed"
    loop = asyncio.get_running_loop()
    # scheduler + executor in their own process unless --disable-frontend-multiprocessing
    # (reference src/launch.py:408 passes the same flag through)
    multiprocess = not getattr(args, "disable_frontend_multiprocessing", False)
    engine = await loop.run_in_executor(None, lambda: AsyncLLM.from_engine_args(engine_args, multiprocess))
    try:
        yield engine
    finally:
        engine.shutdown()


def build_app(args: argparse.Namespace) -> FastAPI:
    app = FastAPI(title="vllm-distributed-amd", version=__version__)
    api_key = getattr(args, "api_key", None) or envs.VLLM_API_KEY
    # cross-origin access for browser clients; open by default, as in vLLM's server
    from fastapi.middleware.cors import CORSMiddleware
    app.add_middleware(CORSMiddleware,
                       allow_origins=getattr(args, "allowed_origins", None) or ["*"],
                       allow_credentials=bool(getattr(args, "allow_credentials", False)),
                       allow_methods=getattr(args, "allowed_methods", None) or ["*"],
                       allow_headers=getattr(args, "allowed_headers", None) or ["*"])

    if api_key:
        import hmac
        want = ("Bearer " + api_key).encode()
        # liveness / scrape endpoints stay open; everything that generates, tokenizes or
        # administers needs the key (``/invocations`` reaches the same handlers as ``/v1``)
        public = {"/health", "/ping", "/version", "/metrics"}

        @app.middleware("http")
        async def auth(request: Request, call_next):
            if request.method != "OPTIONS" and request.url.path.rstrip("/") not in public:
                got = (request.headers.get("Authorization") or "").encode()
                if not hmac.compare_digest(got, want):
                    return JSONResponse({"error": "Unauthorized"}, status_code=401)
            return await call_next(request)

    _register_routes(app, dev_mode=bool(getattr(args, "enable_dev_endpoints", False)) or envs.VLLM_SERVER_DEV_MODE)
    return app


def maybe_register_tokenizer_info_endpoint(args: argparse.Namespace) -> None:
    """Kept for bring-up parity; the endpoint is always registered and gated at
    request time by ``--enable-tokenizer-info-endpoint``."""
    return None


def _register_lora(state, name: str, path: str):
    from ...lora import LoRARequest
    if name in state.loras or name in state.served_names:
        raise ValueError(f"model name {name!r} is already served")
    if not os.path.isfile(os.path.join(path, "adapter_config.json")):
        raise ValueError(f"{path!r} is not a LoRA adapter directory (no adapter_config.json)")
    state.loras[name] = LoRARequest(name, state.next_lora_id, path)
    state.next_lora_id += 1
    return state.loras[name]


async def init_app_state(engine: AsyncLLM, vllm_config, state, args: argparse.Namespace) -> None:
    state.engine = engine
    state.args = args
    state.served_names = vllm_config.model_config.served_model_name
    state.max_model_len = vllm_config.model_config.max_model_len
    state.max_logprobs = int(getattr(args, "max_logprobs", 20) or 20)
    # the model authors' generation_config.json fills in what a request leaves unset (--generation-config)
    state.default_sampling = vllm_config.model_config.default_sampling_params()
    if state.default_sampling:
        logger.info("default sampling parameters from generation_config.json: %s", state.default_sampling)
    state.tool_parser = None
    name = getattr(args, "tool_call_parser", None)
    if getattr(args, "enable_auto_tool_choice", False) and name:
        state.tool_parser = ToolParserManager.get_tool_parser(name)(engine.tokenizer)
    state.reasoning = None
    if getattr(args, "reasoning_parser", None):
        from .reasoning import ReasoningParserManager
        state.reasoning = ReasoningParserManager.get(args.reasoning_parser)
    state.chat_template = None
    ct = getattr(args, "chat_template", None)
    if ct:
        try:
            with open(ct) as f:
                state.chat_template = f.read()
        except OSError:
            state.chat_template = ct
    # vision-language models: image_url content parts become images of the request; each leaves the
    # tokenizer's spelling of the placeholder token in the prompt text
    hf = vllm_config.model_config.hf_config
    state.takes_images = hf.get("mm_model_type") is not None
    state.anyres_cfg = hf if hf.get("mm_model_type") == "llava_next" else None
    state.image_placeholder = "<image>"
    if state.takes_images:
        from ...multimodal import tokens_per_image
        state.tokens_per_image = tokens_per_image(hf)
        try:
            spelled = engine.tokenizer.convert_ids_to_tokens([int(hf.get("image_token_index", hf.get("image_token_id")))])
            if spelled and isinstance(spelled[0], str) and spelled[0]:
                state.image_placeholder = spelled[0]
        except Exception:  # noqa: BLE001 - a tokenizer without that table keeps the conventional spelling
            pass
    state.start_time = time.time()
    # served LoRA adapters: name -> LoRARequest (``--lora-modules name=path ...`` and /v1/load_lora_adapter)
    state.loras = {}
    state.next_lora_id = 1
    for item in getattr(args, "lora_modules", None) or []:
        name, _, path = item.partition("=")
        if not name or not path:
            raise ValueError(f"--lora-modules expects name=path, got {item!r}")
        _register_lora(state, name, path)


# -------------------------------------------------------------------- routes
def _sampling_params(req, default_max: Optional[int], max_model_len: int, logprobs: Optional[int],
                     defaults: Optional[Dict[str, Any]] = None, max_logprobs: int = 20) -> SamplingParams:
    for name, val in (("logprobs", logprobs), ("prompt_logprobs", req.prompt_logprobs)):
        if val is not None and val > max_logprobs:
            raise ValueError(f"{name}={val} exceeds this server's limit of {max_logprobs} (--max-logprobs)")
    given = req.model_fields_set
    defaults = defaults or {}

    def pick(name: str, fallback):
        v = getattr(req, name)
        if name not in given or v is None:
            return defaults.get(name, fallback)
        return v
    extra = req.model_extra or {}
    guided = _guided_options(req, extra)
    max_tokens = getattr(req, "max_completion_tokens", None) or (
        req.max_tokens if "max_tokens" in given else None) or defaults.get("max_tokens") or req.max_tokens \
        or default_max
    return SamplingParams(
        temperature=pick("temperature", 1.0), top_p=pick("top_p", 1.0), top_k=pick("top_k", -1) or -1,
        min_p=pick("min_p", 0.0) or 0.0, presence_penalty=req.presence_penalty or 0.0,
        frequency_penalty=req.frequency_penalty or 0.0, repetition_penalty=pick("repetition_penalty", 1.0) or 1.0,
        seed=req.seed, max_tokens=max_tokens, min_tokens=req.min_tokens, stop=req.stop,
        stop_token_ids=req.stop_token_ids, ignore_eos=req.ignore_eos, logprobs=logprobs,
        skip_special_tokens=req.skip_special_tokens, logit_bias=req.logit_bias,
        allowed_token_ids=req.allowed_token_ids, bad_words=req.bad_words,
        truncate_prompt_tokens=req.truncate_prompt_tokens, **guided,
        prompt_logprobs=(req.prompt_logprobs if req.prompt_logprobs is not None else
                         logprobs if getattr(req, "echo", False) and not req.stream else None))


async def _sampling_params_off_loop(req, *args) -> SamplingParams:
    """``_sampling_params`` for a route handler.  A request that carries an output constraint has its
    pattern / schema / grammar compiled to an automaton while the parameters are validated (up to
    seconds for a large one): that runs on a worker thread, so the event loop keeps serving streams."""
    extra = req.model_extra or {}
    constrained = bool(req.response_format) or bool(extra.get("structured_outputs")) or any(
        extra.get(k) is not None for k in ("guided_json", "guided_regex", "guided_choice", "guided_grammar"))
    if not constrained:
        return _sampling_params(req, *args)
    return await asyncio.get_running_loop().run_in_executor(None, _sampling_params, req, *args)


def _guided_options(req, extra: Dict[str, Any]) -> Dict[str, Any]:
    """The one output constraint a request asks for, as SamplingParams fields.  Accepted spellings:
    ``response_format`` (json_object, json_schema), vLLM's ``guided_json`` / ``guided_regex`` /
    ``guided_choice`` / ``guided_grammar`` and the ``structured_outputs`` object carrying ``json`` /
    ``regex`` / ``choice`` / ``grammar`` / ``json_object``.  Grammars are EBNF in the GBNF spelling
    (``root ::= ...``)."""
    so = extra.get("structured_outputs") or {}
    if not isinstance(so, dict):
        raise ValueError("structured_outputs must be an object")
    out: Dict[str, Any] = {}
    rf = req.response_format or {}
    kind = rf.get("type", "text")
    if kind == "json_object" or so.get("json_object") or extra.get("guided_json_object"):
        out["json_object"] = True
    elif kind == "json_schema":
        js = rf.get("json_schema") or {}
        schema = js.get("schema", js.get("json_schema"))
        if schema is None:
            raise ValueError("response_format json_schema needs json_schema.schema")
        out["guided_json"] = schema
    elif kind != "text":
        raise ValueError(f"response_format type {kind!r} is not supported (text, json_object, json_schema)")
    for field, keys in (("guided_json", ("guided_json", "json")), ("guided_regex", ("guided_regex", "regex")),
                        ("guided_choice", ("guided_choice", "choice")),
                        ("guided_grammar", ("guided_grammar", "grammar"))):
        val = extra.get(keys[0]) if extra.get(keys[0]) is not None else so.get(keys[1])
        if val is not None:
            out[field] = val
    if len(out) > 1:
        raise ValueError(f"a request can carry one output constraint, got {sorted(out)}")
    return out


def _prompt_logprobs_json(tok, plps) -> Optional[list]:
    """vLLM's response form: per prompt token ``{token id: {logprob, decoded_token}}`` (null for the first)."""
    if plps is None:
        return None
    return [None if d is None else {str(t): {"logprob": lp, "decoded_token": tok.decode([t])} for t, lp in d.items()}
            for d in plps]


def _usage(n_prompt: int, n_out: int) -> dict:
    return {"prompt_tokens": n_prompt, "completion_tokens": n_out, "total_tokens": n_prompt + n_out}


def _sse(obj: Any) -> str:
    return "data: " + (obj if isinstance(obj, str) else json.dumps(obj, ensure_ascii=False)) + "\n\n"


def _messages_to_dicts(messages: List[ChatMessage], images: Optional[list] = None,
                       image_placeholder: str = "<image>") -> List[dict]:
    """Content parts -> text.  ``image_url`` parts are collected into ``images`` (in order of appearance) and
    leave one placeholder line each at the head of their message, the form LLaVA-style prompts take."""
    out = []
    for m in messages:
        d = m.model_dump(exclude_none=True)
        c = d.get("content")
        if isinstance(c, list):
            n_img = 0
            for part in c:
                if isinstance(part, dict) and part.get("type") in ("image_url", "image", "input_image"):
                    if images is None:
                        raise ValueError("this model does not take images")
                    src = part.get("image_url") or part.get("image") or part.get("url")
                    images.append(src.get("url") if isinstance(src, dict) else src)
                    n_img += 1
            text = "".join(p.get("text", "") for p in c if isinstance(p, dict))
            d["content"] = (image_placeholder + "\n") * n_img + text
        out.append(d)
    return out


def _register_routes(app: FastAPI, dev_mode: bool = False) -> None:
    @app.get("/health")
    async def health(request: Request):
        try:
            await request.app.state.engine.check_health()
            return Response(status_code=200)
        except EngineDeadError:
            return Response(status_code=503)

    @app.get("/ping")
    @app.post("/ping")
    async def ping(request: Request):
        return await health(request)

    @app.get("/version")
    async def version():
        return {"version": __version__}

    @app.get("/v1/models")
    async def models(request: Request):
        st = request.app.state
        return {"object": "list", "data": [
            {"id": n, "object": "model", "created": int(st.start_time), "owned_by": "vllm-distributed-amd",
             "root": st.served_names[0], "max_model_len": st.max_model_len} for n in st.served_names] + [
            {"id": n, "object": "model", "created": int(st.start_time), "owned_by": "vllm-distributed-amd",
             "root": lr.lora_path, "parent": st.served_names[0], "max_model_len": st.max_model_len}
            for n, lr in st.loras.items()]}

    @app.get("/metrics")
    async def metrics(request: Request):
        return Response(request.app.state.engine.metrics.render(), media_type="text/plain; version=0.0.4")

    @app.post("/tokenize")
    async def tokenize(req: TokenizeRequest, request: Request):
        tok = request.app.state.engine.tokenizer
        if req.messages is not None:
            st = request.app.state
            text = tok.apply_chat_template(_messages_to_dicts(req.messages, [], st.image_placeholder), tokenize=False,
                                           add_generation_prompt=req.add_generation_prompt)
            ids = tok.encode(text, add_special_tokens=False)
        else:
            ids = tok.encode(req.prompt or "", add_special_tokens=req.add_special_tokens)
        return {"tokens": ids, "count": len(ids), "max_model_len": request.app.state.max_model_len}

    @app.post("/detokenize")
    async def detokenize(req: DetokenizeRequest, request: Request):
        return {"prompt": request.app.state.engine.tokenizer.decode(req.tokens, skip_special_tokens=False)}

    @app.get("/get_tokenizer_info")
    async def tokenizer_info(request: Request):
        if not getattr(request.app.state.args, "enable_tokenizer_info_endpoint", False):
            return _error("tokenizer info endpoint is disabled", 404, "NotFoundError")
        tok = request.app.state.engine.tokenizer
        return {"tokenizer_class": type(tok).__name__, "vocab_size": len(tok),
                "chat_template": getattr(tok, "chat_template", None),
                "bos_token_id": tok.bos_token_id, "eos_token_id": tok.eos_token_id}

    # profiling, sleep / wake-up and cache resets change the engine under its users' feet: only
    # registered when the operator asks for them (--enable-dev-endpoints / VLLM_SERVER_DEV_MODE=1)
    if dev_mode:
        @app.post("/start_profile")
        async def start_profile(request: Request):
            await request.app.state.engine.start_profile()
            return Response(status_code=200)

        @app.post("/stop_profile")
        async def stop_profile(request: Request):
            await request.app.state.engine.stop_profile()
            return Response(status_code=200)

        @app.post("/reset_prefix_cache")
        async def reset_prefix_cache(request: Request):
            ok = await request.app.state.engine.reset_prefix_cache()
            return Response(status_code=200 if ok else 409)

        @app.post("/sleep")
        async def sleep(request: Request, level: int = 1):
            if request.app.state.engine.engine.get_num_unfinished_requests() > 0:
                return JSONResponse({"error": "requests are in flight; retry when the server is idle"},
                                    status_code=409)
            await request.app.state.engine.sleep(level)
            request.app.state.sleeping = True
            return Response(status_code=200)

        @app.post("/wake_up")
        async def wake_up(request: Request):
            await request.app.state.engine.wake_up()
            request.app.state.sleeping = False
            return Response(status_code=200)

        @app.get("/is_sleeping")
        async def is_sleeping(request: Request):
            return {"is_sleeping": bool(getattr(request.app.state, "sleeping", False))}

    @app.post("/invocations")
    async def invocations(request: Request):
        """Single-endpoint form (SageMaker-style): the body decides between chat and text completion."""
        try:
            body = await request.json()
        except Exception:  # noqa: BLE001
            return _error("the request body is not JSON")
        try:
            if isinstance(body, dict) and "messages" in body:
                return await chat_completions(ChatCompletionRequest(**body), request)
            if isinstance(body, dict) and "prompt" in body:
                return await completions(CompletionRequest(**body), request)
        except ValueError as e:             # pydantic validation errors are ValueErrors
            return _error(str(e))
        return _error("the body needs either `messages` (chat) or `prompt` (completion)")

    def _check_model(st, name: Optional[str]) -> Optional[JSONResponse]:
        if name and name not in st.served_names and name not in st.loras:
            return _error(f"The model `{name}` does not exist.", 404, "NotFoundError")
        return None

    @app.post("/v1/load_lora_adapter")
    async def load_lora_adapter(request: Request):
        st = request.app.state
        body = await request.json()
        name, path = body.get("lora_name"), body.get("lora_path")
        if not name or not path:
            return _error("lora_name and lora_path are required")
        if getattr(st.args, "enable_lora", False) is False:
            return _error("the server was started without --enable-lora")
        try:
            lr = _register_lora(st, name, path)
            await st.engine.add_lora(lr)          # load now so a bad adapter fails here, not in a request
        except Exception as e:  # noqa: BLE001
            st.loras.pop(name, None)
            return _error(str(e))
        return Response(f"Success: LoRA adapter '{name}' added successfully.", status_code=200)

    @app.post("/v1/unload_lora_adapter")
    async def unload_lora_adapter(request: Request):
        st = request.app.state
        body = await request.json()
        lr = st.loras.get(body.get("lora_name"))
        if lr is None:
            return _error(f"The lora adapter '{body.get('lora_name')}' cannot be found.", 404, "NotFoundError")
        try:
            await st.engine.remove_lora(lr.lora_int_id)
        except Exception as e:  # noqa: BLE001
            return _error(str(e))
        del st.loras[lr.lora_name]
        return Response(f"Success: LoRA adapter '{lr.lora_name}' removed successfully.", status_code=200)

    # ------------------------------------------------------- /v1/completions
    @app.post("/v1/completions")
    async def completions(req: CompletionRequest, request: Request):
        st = request.app.state
        engine: AsyncLLM = st.engine
        err = _check_model(st, req.model)
        if err:
            return err
        n = max(1, int(req.n or 1))
        p = req.prompt
        if isinstance(p, str):
            prompts: List[Any] = [p]
        elif p and isinstance(p[0], int):
            prompts = [list(p)]
        else:
            prompts = list(p)
        if not prompts:
            return _error("empty prompt")
        model = req.model or st.served_names[0]
        rid = "cmpl-" + uuid.uuid4().hex
        created = int(time.time())
        try:
            sp = await _sampling_params_off_loop(req, 16, st.max_model_len, req.logprobs, st.default_sampling,
                                                 st.max_logprobs)
        except ValueError as e:
            return _error(str(e))
        lora = st.loras.get(req.model) if req.model else None
        best_of = int(req.best_of or n)
        if best_of < n:
            return _error("best_of must be >= n")
        if best_of > MAX_SAMPLES_PER_PROMPT or (req.use_beam_search and n > MAX_BEAM_WIDTH):
            # every sample (and every beam, every step) is a request of its own
            return _error(f"at most {MAX_SAMPLES_PER_PROMPT} samples (n / best_of) and beams of width "
                          f"{MAX_BEAM_WIDTH} per prompt")
        if (req.use_beam_search or best_of > n) and req.stream:
            return _error("beam search and best_of > n rank finished sequences: they cannot be streamed")
        if req.use_beam_search:
            tok = engine.tokenizer
            bp = BeamSearchParams(beam_width=n, max_tokens=sp.max_tokens or 16, length_penalty=req.length_penalty,
                                  ignore_eos=req.ignore_eos)
            try:
                id_lists = [list(pr) if not isinstance(pr, str) else tok.encode(pr) for pr in prompts]
                found = await asyncio.gather(*[
                    beam_search_async(engine, ids, bp, req.priority, lora, st.max_logprobs) for ids in id_lists])
            except ValueError as e:
                return _error(str(e))
            except EngineDeadError as e:
                return _error(str(e), 500, "InternalServerError")
            choices, n_out = [], 0
            for i, (pr, beams) in enumerate(zip(prompts, found)):
                for b in beams:
                    text = tok.decode(b.tokens)
                    if req.echo:
                        text = (pr if isinstance(pr, str) else tok.decode(list(pr))) + text
                    choices.append({"index": len(choices), "text": text, "logprobs": None,
                                    "cumulative_logprob": b.cum_logprob,
                                    "finish_reason": "stop" if b.finished else "length", "stop_reason": None})
                    n_out += len(b.tokens)
            return {"id": rid, "object": "text_completion", "created": created, "model": model,
                    "choices": choices, "usage": _usage(sum(len(x) for x in id_lists), n_out)}
        if best_of > n and sp.logprobs is None:
            sp.logprobs = 0                   # the ranking needs every sample's cumulative logprob
        # n (or best_of) samples per prompt = that many requests sharing the prompt (its KV blocks are shared
        # through the prefix cache); a fixed seed is advanced per sample so the samples differ reproducibly
        gens = [engine.generate(pr, _nth_params(sp, j), f"{rid}-{i}-{j}", req.priority, lora)
                for i, pr in enumerate(prompts) for j in range(best_of)]

        def lp_block(out, start: int, with_prompt: bool = False) -> Optional[dict]:
            lps = out.outputs[0].logprobs
            if req.logprobs is None or not lps:
                return None
            tok = engine.tokenizer
            ids = out.outputs[0].token_ids[start:]
            sel = lps[start:]
            if with_prompt and out.prompt_logprobs:      # echo: the prompt tokens lead, the first has no score
                ids = list(out.prompt_token_ids) + list(ids)
                sel = [d or {} for d in out.prompt_logprobs] + list(sel)
            return {"tokens": tok.convert_ids_to_tokens(ids),
                    "token_logprobs": [d.get(t) for t, d in zip(ids, sel)],
                    "top_logprobs": [{tok.decode([k]): v for k, v in d.items()} for d in sel],
                    "text_offset": []}

        if req.stream:
            async def stream() -> AsyncIterator[str]:
                n_prompt = n_out = 0
                try:
                    for i, g in enumerate(gens):
                        sent = seen = 0
                        async for out in g:
                            seen += 1
                            if seen % 16 == 0 and await request.is_disconnected():
                                return
                            c = out.outputs[0]
                            text = out.new_text
                            if req.echo and sent == 0:
                                text = (out.prompt or "") + text
                            # one event per engine output that carries tokens, even while the
                            # detokenizer holds their text back (incomplete UTF-8, stop-string tail)
                            if text or out.finished or len(c.token_ids) > sent:
                                yield _sse({"id": rid, "object": "text_completion", "created": created,
                                            "model": model, "choices": [{
                                                "index": i, "text": text, "logprobs": lp_block(out, sent),
                                                "finish_reason": c.finish_reason, "stop_reason": c.stop_reason}]})
                                sent = len(c.token_ids)
                            if out.finished:
                                if i % n == 0:
                                    n_prompt += len(out.prompt_token_ids)
                                n_out += len(c.token_ids)
                    if req.stream_options and req.stream_options.include_usage:
                        yield _sse({"id": rid, "object": "text_completion", "created": created, "model": model,
                                    "choices": [], "usage": _usage(n_prompt, n_out)})
                except Exception as e:  # noqa: BLE001
                    yield _sse({"error": {"message": str(e), "type": type(e).__name__}})
                yield _sse("[DONE]")
            return StreamingResponse(stream(), media_type="text/event-stream")

        async def collect(g):
            last = None
            async for out in g:
                last = out
            return last
        try:
            finals = await asyncio.gather(*[collect(g) for g in gens])
        except ValueError as e:
            return _error(str(e))
        except EngineDeadError as e:
            return _error(str(e), 500, "InternalServerError")
        choices = []
        n_prompt = sum(len(finals[i].prompt_token_ids) for i in range(0, len(finals), best_of))
        n_out = sum(len(out.outputs[0].token_ids) for out in finals)      # every sample was generated, kept or not
        if best_of > n:
            kept = []
            for g in range(0, len(finals), best_of):
                group = sorted(finals[g:g + best_of], key=lambda o: -(
                    o.outputs[0].cumulative_logprob if o.outputs[0].cumulative_logprob is not None
                    else float("-inf")))
                kept.extend(group[:n])
            finals = kept
        for i, out in enumerate(finals):
            c = out.outputs[0]
            text = ((out.prompt or "") if req.echo else "") + c.text
            choices.append({"index": i, "text": text, "logprobs": lp_block(out, 0, req.echo),
                            "finish_reason": c.finish_reason, "stop_reason": c.stop_reason})
            if req.prompt_logprobs is not None:
                choices[-1]["prompt_logprobs"] = _prompt_logprobs_json(engine.tokenizer, out.prompt_logprobs)
        return {"id": rid, "object": "text_completion", "created": created, "model": model,
                "choices": choices, "usage": _usage(n_prompt, n_out)}

    # -------------------------------------------------- /v1/chat/completions
    @app.post("/v1/chat/completions")
    async def chat_completions(req: ChatCompletionRequest, request: Request):
        st = request.app.state
        engine: AsyncLLM = st.engine
        err = _check_model(st, req.model)
        if err:
            return err
        n = max(1, int(req.n or 1))
        tok = engine.tokenizer
        images: Optional[list] = [] if st.takes_images else None
        try:
            messages = _messages_to_dicts(req.messages, images, st.image_placeholder)
            if images:
                from ...multimodal import load_image
                images = [load_image(src) for src in images]
        except ValueError as e:
            return _error(str(e))
        tools = req.tools if req.tool_choice != "none" else None
        # a named tool: the reply *is* that call, its arguments constrained to one JSON object
        forced_tool: Optional[str] = None
        if isinstance(req.tool_choice, dict):
            forced_tool = (req.tool_choice.get("function") or {}).get("name")
            known = [(t.get("function") or {}).get("name") for t in (req.tools or [])]
            if not forced_tool or forced_tool not in known:
                return _error(f"tool_choice names {forced_tool!r}, which is not one of the request's tools")
        elif req.tool_choice == "required":
            if not req.tools:
                return _error("tool_choice 'required' needs a non-empty tools list")
        try:
            kwargs = dict(req.chat_template_kwargs or {})
            if req.chat_template and not getattr(st.args, "trust_request_chat_template", False):
                return _error("a chat_template in the request is only honoured when the server was started "
                              "with --trust-request-chat-template")
            template = req.chat_template or st.chat_template
            if template:
                from ...tokenizer import render_chat_template
                prompt = render_chat_template(template, messages, req.add_generation_prompt, tools, **kwargs)
            else:
                prompt = tok.apply_chat_template(messages, tokenize=False, tools=tools,
                                                 add_generation_prompt=req.add_generation_prompt, **kwargs)
        except Exception as e:  # noqa: BLE001
            return _error(f"chat template error: {e}")
        prompt_ids = tok.encode(prompt, add_special_tokens=False)
        n_lp = (req.top_logprobs or 0) if req.logprobs else None
        try:
            sp = await _sampling_params_off_loop(req, None, st.max_model_len, n_lp, st.default_sampling,
                                                 st.max_logprobs)
        except ValueError as e:
            return _error(str(e))
        required_tools = req.tool_choice == "required"
        try:
            if forced_tool:
                # the arguments follow the tool's parameter schema when it has one
                fn = next((t.get("function") or {}) for t in req.tools
                          if (t.get("function") or {}).get("name") == forced_tool)
                sp = _constrain(sp, fn.get("parameters"))
            elif required_tools:
                # the reply is a JSON array of at least one {"name", "parameters"} call
                calls = [{"type": "object", "required": ["name", "parameters"], "properties": {
                    "name": {"const": (t.get("function") or {}).get("name")},
                    "parameters": (t.get("function") or {}).get("parameters") or {"type": "object"}}}
                    for t in req.tools]
                sp = _constrain(sp, {"type": "array", "minItems": 1, "items": {"anyOf": calls}})
        except ValueError as e:
            return _error(str(e))
        model = req.model or st.served_names[0]
        rid = "chatcmpl-" + uuid.uuid4().hex
        created = int(time.time())
        parser = st.tool_parser if tools and not forced_tool and not required_tools else None
        if required_tools:
            parser = _RequiredToolsParser()
        if n > MAX_SAMPLES_PER_PROMPT or (req.use_beam_search and n > MAX_BEAM_WIDTH):
            return _error(f"at most {MAX_SAMPLES_PER_PROMPT} samples (n) and beams of width {MAX_BEAM_WIDTH} per prompt")
        if req.use_beam_search:
            if req.stream:
                return _error("beam search ranks finished sequences: it cannot be streamed")
            bp = BeamSearchParams(beam_width=n, max_tokens=sp.max_tokens or max(1, st.max_model_len - len(prompt_ids)),
                                  length_penalty=req.length_penalty, ignore_eos=req.ignore_eos)
            try:
                beams = await beam_search_async(engine, prompt_ids, bp, req.priority,
                                                st.loras.get(req.model) if req.model else None, st.max_logprobs)
            except ValueError as e:
                return _error(str(e))
            except EngineDeadError as e:
                return _error(str(e), 500, "InternalServerError")
            choices = [{"index": i, "message": {"role": "assistant", "content": tok.decode(b.tokens)},
                        "logprobs": None, "cumulative_logprob": b.cum_logprob,
                        "finish_reason": "stop" if b.finished else "length", "stop_reason": None}
                       for i, b in enumerate(beams)]
            return {"id": rid, "object": "chat.completion", "created": created, "model": model, "choices": choices,
                    "usage": _usage(len(prompt_ids), sum(len(b.tokens) for b in beams))}
        from .tool_parsers import _call_id
        if images and req.use_beam_search:
            return _error("beam search over prompts with images is not supported")
        full_prompt: Dict[str, Any] = {"prompt": prompt, "prompt_token_ids": prompt_ids}
        n_prompt = len(prompt_ids)
        if images:
            full_prompt["multi_modal_data"] = {"image": images}
            # each placeholder widens to its image's embeddings (any-resolution models: as many as its aspect leaves)
            if st.anyres_cfg is not None:
                from ...multimodal import anyres_tokens, load_image
                try:
                    n_prompt += sum(anyres_tokens(st.anyres_cfg, load_image(im).size[::-1]) - 1 for im in images)
                except ValueError as e:
                    return _error(str(e))
            else:
                n_prompt += len(images) * (st.tokens_per_image - 1)
        gens = [engine.generate(full_prompt, _nth_params(sp, j),
                                rid if n == 1 else f"{rid}-{j}", req.priority,
                                st.loras.get(req.model) if req.model else None) for j in range(n)]
        gen = gens[0]

        def lp_content(out, start: int) -> Optional[dict]:
            lps = out.outputs[0].logprobs
            if not req.logprobs or not lps:
                return None
            ids = out.outputs[0].token_ids[start:]
            items = []
            for t, d in zip(ids, lps[start:]):
                items.append({"token": tok.decode([t]), "logprob": d.get(t), "bytes": None,
                              "top_logprobs": [{"token": tok.decode([k]), "logprob": v, "bytes": None}
                                               for k, v in d.items()]})
            return {"content": items}

        if req.stream:
            def chunk(index: int, delta: dict, finish=None, lp=None) -> str:
                return _sse({"id": rid, "object": "chat.completion.chunk", "created": created,
                             "model": model, "choices": [{"index": index, "delta": delta, "logprobs": lp,
                                                          "finish_reason": finish}]})

            async def choice_events(index: int, g, counts: List[int]) -> AsyncIterator[str]:
                """SSE events of one sample; ``counts[index]`` ends up holding its token count."""
                yield chunk(index, {"role": "assistant", "content": ""})
                if forced_tool:
                    yield chunk(index, {"tool_calls": [{"index": 0, "id": _call_id(), "type": "function",
                                                        "function": {"name": forced_tool, "arguments": ""}}]})
                sent = 0
                final = None
                held = ""
                seen = 0
                splitter = st.reasoning() if st.reasoning is not None else None
                async for out in g:
                    seen += 1
                    if seen % 16 == 0 and await request.is_disconnected():
                        return
                    final = out
                    text = out.new_text
                    if splitter is not None:
                        thought, text = splitter.feed(text)
                        if out.finished:
                            t2, c2 = splitter.flush()
                            thought, text = thought + t2, text + c2
                        if thought:
                            yield chunk(index, {"reasoning_content": thought}, lp=lp_content(out, sent))
                            sent = len(out.outputs[0].token_ids)
                    if parser is not None:
                        # tool-call syntax must be parsed as a whole: hold text once a call starts
                        held += text
                        if required_tools or "<tool_call>" in held or \
                                held.lstrip().startswith(("{", "<|python_tag|>")):
                            text = ""
                        else:
                            text, held = held, ""
                    if forced_tool:
                        if text:
                            yield chunk(index, {"tool_calls": [{"index": 0, "function": {"arguments": text}}]})
                    elif text or (parser is None and len(out.outputs[0].token_ids) > sent):
                        yield chunk(index, {"content": text}, lp=lp_content(out, sent))
                        sent = len(out.outputs[0].token_ids)
                c = final.outputs[0]
                counts[index] = len(c.token_ids)
                finish = "tool_calls" if forced_tool and c.finish_reason == "stop" else c.finish_reason
                if parser is not None and held:
                    content, calls = parser.extract(held, tools)
                    if calls:
                        if content:
                            yield chunk(index, {"content": content})
                        yield chunk(index, {"tool_calls": [dict(tc, index=i) for i, tc in enumerate(calls)]})
                        finish = "tool_calls"
                    else:
                        yield chunk(index, {"content": held})
                yield chunk(index, {}, finish=finish)

            async def stream() -> AsyncIterator[str]:
                counts = [0] * n
                try:
                    if n == 1:
                        async for ev in choice_events(0, gens[0], counts):
                            yield ev
                    else:
                        # the n samples advance together: interleave their events as they arrive
                        queue: asyncio.Queue = asyncio.Queue()

                        async def pump(index: int, g) -> None:
                            try:
                                async for ev in choice_events(index, g, counts):
                                    await queue.put(ev)
                            except Exception as e:  # noqa: BLE001
                                await queue.put(e)
                            finally:
                                await queue.put(None)
                        tasks = [asyncio.ensure_future(pump(k, g)) for k, g in enumerate(gens)]
                        try:
                            live = n
                            while live:
                                ev = await queue.get()
                                if ev is None:
                                    live -= 1
                                elif isinstance(ev, Exception):
                                    raise ev
                                else:
                                    yield ev
                        finally:
                            for t in tasks:
                                t.cancel()
                    if req.stream_options and req.stream_options.include_usage:
                        yield _sse({"id": rid, "object": "chat.completion.chunk", "created": created,
                                    "model": model, "choices": [],
                                    "usage": _usage(n_prompt, sum(counts))})
                except Exception as e:  # noqa: BLE001
                    yield _sse({"error": {"message": str(e), "type": type(e).__name__}})
                yield _sse("[DONE]")
            return StreamingResponse(stream(), media_type="text/event-stream")

        async def last_of(g):
            final = None
            async for out in g:
                final = out
            return final
        try:
            finals = await asyncio.gather(*[last_of(g) for g in gens])
        except ValueError as e:
            return _error(str(e))
        except EngineDeadError as e:
            return _error(str(e), 500, "InternalServerError")
        choices, n_out = [], 0
        for j, final in enumerate(finals):
            c = final.outputs[0]
            text, thought = c.text, None
            if st.reasoning is not None:
                thought, text = st.reasoning.split(c.text)
            message: Dict[str, Any] = {"role": "assistant", "content": text}
            finish = c.finish_reason
            if parser is not None:
                content, calls = parser.extract(text, tools)
                if calls:
                    message = {"role": "assistant", "content": content, "tool_calls": calls}
                    finish = "tool_calls"
            if forced_tool:
                message = {"role": "assistant", "content": None, "tool_calls": [
                    {"id": _call_id(), "type": "function", "function": {"name": forced_tool, "arguments": text}}]}
                finish = "tool_calls" if c.finish_reason == "stop" else c.finish_reason
            if st.reasoning is not None:
                message["reasoning_content"] = thought or None
                if not message["content"]:
                    message["content"] = None
            choices.append({"index": j, "message": message, "logprobs": lp_content(final, 0),
                            "finish_reason": finish, "stop_reason": c.stop_reason})
            n_out += len(c.token_ids)
        body = {"id": rid, "object": "chat.completion", "created": created, "model": model,
                "choices": choices, "usage": _usage(n_prompt, n_out)}
        if req.prompt_logprobs is not None and finals:
            body["prompt_logprobs"] = _prompt_logprobs_json(tok, finals[0].prompt_logprobs)
        return body


    from . import pooling_routes, responses_routes
    pooling_routes.register(app, _error, _check_model)
    responses_routes.register(app, chat_completions, ChatCompletionRequest, _error)


def _constrain(sp: SamplingParams, schema) -> SamplingParams:
    """``sp`` with its output held to ``schema`` (a plain JSON object when there is none)."""
    import dataclasses
    if sp.guided_regex is not None or sp.guided_choice is not None or sp.guided_grammar is not None:
        raise ValueError("tool_choice cannot be combined with another output constraint")
    if sp.json_object and schema:
        raise ValueError("tool_choice cannot be combined with response_format json_object")
    if not schema or schema == {"type": "object"}:
        return dataclasses.replace(sp, json_object=True)
    return dataclasses.replace(sp, guided_json=schema)


class _RequiredToolsParser:
    """Reads the constrained reply of ``tool_choice: "required"``: a JSON array of calls."""

    def extract(self, text: str, tools) -> tuple:
        from .tool_parsers import _call_id
        try:
            items = json.loads(text)
        except json.JSONDecodeError:
            return text, []               # cut off by max_tokens: hand back what there is
        calls = [{"id": _call_id(), "type": "function",
                  "function": {"name": it.get("name"),
                               "arguments": json.dumps(it.get("parameters", {}), ensure_ascii=False)}}
                 for it in items if isinstance(it, dict)]
        return None, calls


# ------------------------------------------------------------------- run
async def run_server(args: argparse.Namespace, client_config: Optional[dict] = None, **uvicorn_kwargs) -> None:
    from ..launcher import serve_http
    listen_address, sock = setup_server(args)
    plugin = getattr(args, "tool_parser_plugin", None)
    if plugin and len(plugin) > 3:
        ToolParserManager.import_tool_parser(plugin)
    server_index = client_config.get("client_index", 0) if client_config else 0
    log_config = load_log_config(getattr(args, "log_config_file", None))
    if log_config is not None:
        uvicorn_kwargs["log_config"] = log_config
    async with build_async_engine_client(args, client_config) as engine_client:
        maybe_register_tokenizer_info_endpoint(args)
        app = build_app(args)
        vllm_config = await engine_client.get_vllm_config()
        await init_app_state(engine_client, vllm_config, app.state, args)
        logger.info("Starting API server %d on %s", server_index, listen_address)
        shutdown_task = await serve_http(
            app, sock=sock, enable_ssl_refresh=getattr(args, "enable_ssl_refresh", False),
            host=args.host, port=args.port, log_level=getattr(args, "uvicorn_log_level", "info"),
            access_log=not getattr(args, "disable_uvicorn_access_log", False),
            timeout_keep_alive=envs.VLLM_HTTP_TIMEOUT_KEEP_ALIVE,
            ssl_keyfile=getattr(args, "ssl_keyfile", None), ssl_certfile=getattr(args, "ssl_certfile", None),
            ssl_ca_certs=getattr(args, "ssl_ca_certs", None),
            ssl_cert_reqs=getattr(args, "ssl_cert_reqs", 0), **uvicorn_kwargs)
    # the server's shutdown completes only after the engine context has been left
    # (reference src/launch.py:453-457)
    try:
        await shutdown_task
    finally:
        sock.close()
