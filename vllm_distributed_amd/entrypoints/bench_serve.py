"""``bench serve``: HTTP serving benchmark against a running ``launch.py serve``
instance (the reference's CLI registers vLLM's benchmark sub-commands next to
``serve``, reference ``src/launch.py:465-471``).

Sends ``--num-prompts`` streaming ``/v1/completions`` requests with at most
``--concurrency`` in flight, synthetic prompts of ``--input-len`` token ids,
``--output-len`` new tokens each (``ignore_eos``), and reports the metrics the
deployment is judged by: output tokens/s over the whole run, p50/p90/p99 time to
first token, p50 inter-token latency.  By default all requests are released at
once (a closed loop of ``--concurrency`` clients: time to first token then
mostly measures how long the initial burst's prefill takes); ``--request-rate``
releases them as a Poisson process instead, which is what shows whether a newly
arrived request gets in next to a full decode batch.

    python launch.py bench serve --base-url http://127.0.0.1:8000 --served-model llama-3-8b \
        --num-prompts 256 --concurrency 64 --input-len 128 --output-len 64
"""
from __future__ import annotations

import argparse
import asyncio
import json
import random
import time
from typing import List, Optional


async def _one(session, url: str, headers: dict, payload: dict, result: dict) -> None:
    t0 = time.perf_counter()
    first: Optional[float] = None
    stamps: List[float] = []
    ntok = 0
    async with session.post(url, json=payload, headers=headers) as resp:
        resp.raise_for_status()
        async for raw in resp.content:
            line = raw.decode().strip()
            if not line.startswith("data:"):
                continue
            data = line[5:].strip()
            if data == "[DONE]":
                break
            chunk = json.loads(data)
            now = time.perf_counter()
            if first is None:
                first = now - t0
            stamps.append(now)
            usage = chunk.get("usage")
            if usage:
                ntok = usage.get("completion_tokens", ntok)
    result["ttft"].append(first if first is not None else float("nan"))
    result["tokens"] += ntok if ntok else len(stamps)
    if len(stamps) > 1:
        result["itl"].extend(b - a for a, b in zip(stamps, stamps[1:]))


async def run(args) -> dict:
    import aiohttp
    rng = random.Random(args.seed)
    headers = {"Authorization": f"Bearer {args.api_key}"} if args.api_key else {}
    url = args.base_url.rstrip("/") + "/v1/completions"
    result = {"ttft": [], "itl": [], "tokens": 0}
    sem = asyncio.Semaphore(args.concurrency)
    timeout = aiohttp.ClientTimeout(total=args.timeout)
    # aiohttp caps a session at 100 connections by default: lift it, or requests queue in the client
    async with aiohttp.ClientSession(timeout=timeout, connector=aiohttp.TCPConnector(limit=0)) as session:
        async def guarded(i: int) -> None:
            prompt = [rng.randrange(3, args.vocab_size) for _ in range(args.input_len)]
            out_len = args.output_len
            jitter = float(getattr(args, "output_len_jitter", 0.0) or 0.0)
            if jitter > 0:      # identical lengths make the requests finish (and the next ones arrive) in lock-step
                out_len = max(1, round(args.output_len * (1.0 + rng.uniform(-jitter, jitter))))
            payload = {"model": args.model, "prompt": prompt, "max_tokens": out_len, "temperature": 0.0,
                       "ignore_eos": True, "stream": True, "stream_options": {"include_usage": True}}
            if arrivals is not None:
                await asyncio.sleep(max(0.0, t0 + arrivals[i] - time.perf_counter()))
            async with sem:
                await _one(session, url, headers, payload, result)
        arrivals = None
        rate = float(getattr(args, "request_rate", 0.0) or 0.0)
        if rate > 0:
            t, arrivals = 0.0, []
            for _ in range(args.num_prompts):
                arrivals.append(t)
                t += rng.expovariate(rate)
        t0 = time.perf_counter()
        await asyncio.gather(*(guarded(i) for i in range(args.num_prompts)))
        elapsed = time.perf_counter() - t0

    def pct(xs: List[float], p: float) -> Optional[float]:
        xs = sorted(x for x in xs if x == x)
        return round(1e3 * xs[min(len(xs) - 1, int(p * len(xs)))], 2) if xs else None
    return {"num_prompts": args.num_prompts, "concurrency": args.concurrency, "input_len": args.input_len,
            "output_len": args.output_len, "output_len_jitter": float(getattr(args, "output_len_jitter", 0.0) or 0.0),
            "request_rate": float(getattr(args, "request_rate", 0.0) or 0.0),
            "elapsed_s": round(elapsed, 3),
            "output_tokens_per_s": round(result["tokens"] / elapsed, 1),
            "requests_per_s": round(args.num_prompts / elapsed, 2),
            "ttft_p50_ms": pct(result["ttft"], 0.5), "ttft_p90_ms": pct(result["ttft"], 0.9),
            "ttft_p99_ms": pct(result["ttft"], 0.99),
            "itl_p50_ms": pct(result["itl"], 0.5)}


def add_args(ap: argparse.ArgumentParser) -> None:
    ap.add_argument("--base-url", default="http://127.0.0.1:8000")
    ap.add_argument("--served-model", default=None, help="model name to request (default: first served)")
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--concurrency", type=int, default=16)
    ap.add_argument("--vocab-size", type=int, default=32000)
    ap.add_argument("--timeout", type=float, default=1800.0)
    ap.add_argument("--request-rate", type=float, default=0.0,
                    help="release requests as a Poisson process of this many per second (0: all at once)")
    ap.add_argument("--output-len-jitter", type=float, default=0.0,
                    help="draw each request's output length uniformly from output-len x (1 +- this fraction)")


async def _first_model(args) -> str:
    import aiohttp
    headers = {"Authorization": f"Bearer {args.api_key}"} if args.api_key else {}
    async with aiohttp.ClientSession() as session:
        async with session.get(args.base_url.rstrip("/") + "/v1/models", headers=headers) as resp:
            resp.raise_for_status()
            return (await resp.json())["data"][0]["id"]


def run_sync(args) -> dict:
    if not getattr(args, "seed", None):
        args.seed = 0
    args.model = args.served_model or asyncio.run(_first_model(args))
    return asyncio.run(run(args))


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    add_args(ap)
    ap.add_argument("--num-prompts", type=int, default=64)
    ap.add_argument("--input-len", type=int, default=128)
    ap.add_argument("--output-len", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--model", dest="served_model", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    print(json.dumps(run_sync(args)))


if __name__ == "__main__":
    main()
