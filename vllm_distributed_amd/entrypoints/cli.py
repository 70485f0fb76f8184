"""Head-node command line.

``serve <model>`` starts the OpenAI server on top of the distributed executor
(the only sub-command the reference effectively supports, reference
``src/launch.py:460-507``); ``chat``, ``complete``, ``bench``, ``collect-env``
and ``run-batch`` cover the other sub-commands the reference's parser
registers (``:465-471``).
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import sys
import time
from typing import List, Optional

from .. import envs
from ..engine.arg_utils import EngineArgs
from ..logger import init_logger
from ..utils import FlexibleArgumentParser
from ..version import __version__

logger = init_logger(__name__)


def add_server_args(p: argparse.ArgumentParser) -> None:
    g = p.add_argument_group("server")
    g.add_argument("--host", type=str, default=None)
    g.add_argument("--port", type=int, default=8000)
    g.add_argument("--api-key", type=str, default=None)
    g.add_argument("--uvicorn-log-level", type=str, default="info",
                   choices=["debug", "info", "warning", "error", "critical", "trace"])
    g.add_argument("--disable-uvicorn-access-log", action="store_true")
    g.add_argument("--log-config-file", type=str, default=None)
    g.add_argument("--ssl-keyfile", type=str, default=None)
    g.add_argument("--ssl-certfile", type=str, default=None)
    g.add_argument("--ssl-ca-certs", type=str, default=None)
    g.add_argument("--ssl-cert-reqs", type=int, default=0)
    g.add_argument("--enable-ssl-refresh", action="store_true")
    g.add_argument("--enable-auto-tool-choice", action="store_true")
    g.add_argument("--tool-call-parser", type=str, default=None)
    g.add_argument("--tool-parser-plugin", type=str, default="")
    g.add_argument("--max-logprobs", type=int, default=20,
                   help="largest logprobs / top_logprobs / prompt_logprobs a request may ask for")
    g.add_argument("--allowed-origins", type=json.loads, default=["*"], help="CORS: JSON list of origins")
    g.add_argument("--allowed-methods", type=json.loads, default=["*"])
    g.add_argument("--allowed-headers", type=json.loads, default=["*"])
    g.add_argument("--allow-credentials", action="store_true")
    g.add_argument("--reasoning-parser", type=str, default=None, choices=["qwen3", "deepseek_r1"],
                   help="split <think>...</think> out of chat completions into reasoning_content")
    g.add_argument("--chat-template", type=str, default=None)
    g.add_argument("--enable-tokenizer-info-endpoint", action="store_true")
    g.add_argument("--trust-request-chat-template", action="store_true",
                   help="honour a chat_template sent in a request body (rendered in a sandbox either way)")
    g.add_argument("--enable-dev-endpoints", action="store_true",
                   help="register /sleep, /wake_up, /start_profile, /stop_profile and /reset_prefix_cache")
    g.add_argument("--lora-modules", type=str, nargs="+", default=None, metavar="NAME=PATH",
                   help="LoRA adapters served under their own model names (needs --enable-lora)")
    g.add_argument("--disable-frontend-multiprocessing", action="store_true",
                   help="run the engine core on a thread of the API-server process instead of in its own process")


def _serve_parser(sub) -> argparse.ArgumentParser:
    p = sub.add_parser("serve", help="start the OpenAI-compatible server")
    p.add_argument("model_tag", type=str, nargs="?", default=None, help="model name or path")
    EngineArgs.add_cli_args(p)
    add_server_args(p)
    return p


def cmd_serve(args: argparse.Namespace) -> None:
    from ..executor.distributed import DistributedExecutor
    from .openai.api_server import run_server
    DistributedExecutor.exit_on_failure = True      # fail-stop: the supervisor restarts the container
    if args.enable_auto_tool_choice and not args.tool_call_parser:
        raise SystemExit("--enable-auto-tool-choice requires --tool-call-parser")
    try:
        import uvloop          # optional speed-up; not required
        uvloop.run(run_server(args))
    except ImportError:
        asyncio.run(run_server(args))


def cmd_collect_env(args: argparse.Namespace) -> None:
    import platform
    import torch
    from ..utils import hip_device_count_stateless
    info = {"version": __version__, "python": sys.version.split()[0], "platform": platform.platform(),
            "torch": torch.__version__, "hip": getattr(torch.version, "hip", None),
            "visible_gpus": hip_device_count_stateless(),
            "env": {k: os.environ[k] for k in envs.environment_variables if k in os.environ}}
    try:
        from ..ops import hip
        hip.load()
        info["hip_kernels"] = "loaded"
    except Exception as e:  # noqa: BLE001
        info["hip_kernels"] = f"unavailable: {e}"
    print(json.dumps(info, indent=2))


def cmd_run_batch(args: argparse.Namespace) -> None:
    """OpenAI batch file: one JSON request per line -> one JSON response per line."""
    from ..engine.llm_engine import LLMEngine
    from ..sampling_params import SamplingParams
    engine = LLMEngine.from_engine_args(EngineArgs.from_cli_args(args))
    tok = engine.tokenizer
    reqs = [json.loads(line) for line in open(args.input_file) if line.strip()]
    ids = {}
    for r in reqs:
        body = r.get("body", r)
        if "messages" in body:
            prompt = tok.apply_chat_template(body["messages"], tokenize=False, add_generation_prompt=True)
        else:
            prompt = body["prompt"]
        sp = SamplingParams(temperature=body.get("temperature", 1.0), top_p=body.get("top_p", 1.0),
                            max_tokens=body.get("max_tokens", body.get("max_completion_tokens", 16)),
                            stop=body.get("stop"), seed=body.get("seed"))
        ids[engine.add_request(None, prompt, sp)] = r
    results = {}
    while engine.has_unfinished_requests():
        for out in engine.step():
            if out.finished:
                results[out.request_id] = out
    with open(args.output_file, "w") as f:
        for rid, r in ids.items():
            out = results[rid]
            c = out.outputs[0]
            body = r.get("body", r)
            if "messages" in body:
                choice = {"index": 0, "message": {"role": "assistant", "content": c.text},
                          "finish_reason": c.finish_reason}
            else:
                choice = {"index": 0, "text": c.text, "finish_reason": c.finish_reason}
            f.write(json.dumps({"id": rid, "custom_id": r.get("custom_id"), "response": {
                "status_code": 200, "body": {"choices": [choice], "usage": {
                    "prompt_tokens": len(out.prompt_token_ids), "completion_tokens": len(c.token_ids)}}},
                "error": None}) + "\n")
    engine.shutdown()


def cmd_bench(args: argparse.Namespace) -> None:
    """Offline throughput / latency on synthetic prompts through the engine, or the HTTP client benchmark."""
    if args.bench_type == "serve":
        from . import bench_serve
        print(json.dumps(bench_serve.run_sync(args)))
        return
    import numpy as np
    from ..engine.llm_engine import LLMEngine
    from ..sampling_params import SamplingParams
    engine = LLMEngine.from_engine_args(EngineArgs.from_cli_args(args))
    rng = np.random.default_rng(0)
    vocab = engine.model_config.vocab_size
    sp = SamplingParams(temperature=0.0, max_tokens=args.output_len, ignore_eos=True, detokenize=False)
    t0 = time.perf_counter()
    for i in range(args.num_prompts):
        engine.add_request(f"bench-{i}", rng.integers(0, vocab, size=args.input_len).tolist(), sp)
    n_out, ttfts = 0, []
    while engine.has_unfinished_requests():
        for out in engine.step():
            n_out += len(out.new_token_ids)
            if out.finished and out.metrics and out.metrics.ttft is not None:
                ttfts.append(out.metrics.ttft)
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": args.bench_type, "num_prompts": args.num_prompts, "input_len": args.input_len,
                      "output_len": args.output_len, "elapsed_s": round(dt, 3),
                      "output_tokens_per_s": round(n_out / dt, 1),
                      "total_tokens_per_s": round((n_out + args.num_prompts * args.input_len) / dt, 1),
                      "ttft_p50_ms": round(1e3 * float(np.median(ttfts)), 2) if ttfts else None,
                      "step_breakdown_ms": engine.core.timer.summary()}))
    engine.shutdown()


def build_parser() -> argparse.ArgumentParser:
    parser = FlexibleArgumentParser(description="MI355X-native distributed LLM serving CLI",
                                    epilog="`launch.py remote <head_ip>` starts a worker node instead.")
    parser.add_argument("-v", "--version", action="version", version=__version__)
    sub = parser.add_subparsers(required=False, dest="subparser")
    _serve_parser(sub).set_defaults(dispatch_function=cmd_serve)
    pe = sub.add_parser("collect-env", help="print environment information")
    pe.set_defaults(dispatch_function=cmd_collect_env)
    pb = sub.add_parser("run-batch", help="run an OpenAI batch file offline")
    pb.add_argument("-i", "--input-file", required=True)
    pb.add_argument("-o", "--output-file", required=True)
    EngineArgs.add_cli_args(pb)
    pb.set_defaults(dispatch_function=cmd_run_batch)
    pbe = sub.add_parser("bench", help="offline benchmark on synthetic prompts")
    pbe.add_argument("bench_type", choices=["throughput", "latency", "serve"], nargs="?", default="throughput",
                     help="throughput / latency: offline through the engine; serve: HTTP client against --base-url")
    from . import bench_serve
    bench_serve.add_args(pbe)
    pbe.add_argument("--num-prompts", type=int, default=64)
    pbe.add_argument("--input-len", type=int, default=128)
    pbe.add_argument("--output-len", type=int, default=64)
    EngineArgs.add_cli_args(pbe)
    pbe.set_defaults(dispatch_function=cmd_bench)
    from . import client
    pc = sub.add_parser("chat", help="interactive chat with a running server")
    client.add_client_args(pc)
    pc.add_argument("--system-prompt", type=str, default=None)
    pc.set_defaults(dispatch_function=client.cmd_chat)
    pco = sub.add_parser("complete", help="text completion against a running server")
    client.add_client_args(pco)
    pco.set_defaults(dispatch_function=client.cmd_complete)
    return parser


def head_main(argv: Optional[List[str]] = None) -> None:
    # the head process is not a rank (reference src/launch.py:461-463)
    os.environ["RANK"] = "-1"
    os.environ["LOCAL_RANK"] = "-1"
    os.environ["VLLM_USE_V1"] = "1"
    parser = build_parser()
    args = parser.parse_args(argv)
    if getattr(args, "model_tag", None) is not None:
        args.model = args.model_tag
    if hasattr(args, "dispatch_function"):
        args.dispatch_function(args)
    else:
        parser.print_help()
