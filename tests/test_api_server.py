"""OpenAI-compatible server on a tiny CPU model: endpoints, SSE streaming, auth, tool parsing."""
import asyncio
import json

import httpx
import pytest

from vllm_distributed_amd.entrypoints.cli import build_parser
from vllm_distributed_amd.entrypoints.openai import api_server
from vllm_distributed_amd.entrypoints.openai.tool_parsers import ToolParserManager
from vllm_distributed_amd.parallel import state as pstate


def test_tool_parsers():
    tools = [{"type": "function", "function": {"name": "get_weather", "parameters": {
        "type": "object", "properties": {"city": {"type": "string"}, "days": {"type": "integer"},
                                         "opts": {"type": "object"}}}}}]
    p = ToolParserManager.get_tool_parser("qwen3_coder")()
    text = ("Let me check.\n<tool_call>\n<function=get_weather>\n<parameter=city>\nParis\n</parameter>\n"
            "<parameter=days>\n3\n</parameter>\n<parameter=opts>\n{\"units\": \"C\"}\n</parameter>\n"
            "</function>\n</tool_call>")
    content, calls = p.extract(text, tools)
    assert content == "Let me check." and len(calls) == 1
    assert calls[0]["function"]["name"] == "get_weather"
    assert json.loads(calls[0]["function"]["arguments"]) == {"city": "Paris", "days": 3, "opts": {"units": "C"}}
    assert p.extract("no tools here", tools) == ("no tools here", [])
    h = ToolParserManager.get_tool_parser("hermes")()
    c, calls = h.extract('<tool_call>{"name": "f", "arguments": {"x": 1}}</tool_call>', None)
    assert c is None and json.loads(calls[0]["function"]["arguments"]) == {"x": 1}
    l3 = ToolParserManager.get_tool_parser("llama3_json")()
    c, calls = l3.extract('{"name": "f", "parameters": {"y": [1, 2]}}', None)
    assert calls[0]["function"]["name"] == "f"
    with pytest.raises(KeyError):
        ToolParserManager.get_tool_parser("nope")


def test_cli_parses_reference_command_line():
    # the documented head command (.env.server) must parse
    argv = ("serve koushd/Qwen3-Coder-480B-A35B-Instruct-AWQ -tp 2 -pp 2 --max-model-len 262144 "
            "--served-model-name qwen3 --enable-auto-tool-choice --tool-call-parser qwen3_coder "
            "--host 0.0.0.0 --port 8000 --api-key token-abc123").split()
    a = build_parser().parse_args(argv)
    assert a.model_tag.startswith("koushd/") and a.tensor_parallel_size == 2 and a.pipeline_parallel_size == 2
    assert a.max_model_len == 262144 and a.served_model_name == ["qwen3"] and a.port == 8000
    assert a.enable_auto_tool_choice and a.tool_call_parser == "qwen3_coder" and a.api_key == "token-abc123"
    a = build_parser().parse_args(["serve", "m", "--max_model_len=64"])
    assert a.max_model_len == 64
    # options of vLLM's own CLI that tune machinery absent here are accepted (and reported), rope overrides map
    # onto the HF config, and an option that asks for a missing feature is refused
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    a = build_parser().parse_args(
        ["serve", "llama-tiny", "--swap-space", "4", "--disable-log-requests", "-O", "3", "--max-seq-len-to-capture",
         "8192", "--rope-theta", "1000000", "--rope-scaling", '{"rope_type": "dynamic", "factor": 2.0}',
         "--no-enable-prefix-caching", "--device", "cpu"])
    a.model = a.model_tag
    e = EngineArgs.from_cli_args(a)
    assert e.hf_overrides == {"rope_scaling": {"rope_type": "dynamic", "factor": 2.0}, "rope_theta": 1000000.0}
    assert e.enable_prefix_caching is False and a.compat_swap_space == "4" and a.compat_disable_log_requests
    sa = build_parser().parse_args(["serve", "llama-tiny", "--device", "cpu", "--speculative-config",
                                    '{"method": "ngram", "num_speculative_tokens": 3}'])
    sa.model = sa.model_tag
    cfg = EngineArgs.from_cli_args(sa).create_engine_config()
    assert cfg.speculative_config.num_speculative_tokens == 3 and not cfg.scheduler_config.async_scheduling
    bad = build_parser().parse_args(["serve", "llama-tiny", "--device", "cpu", "--speculative-config",
                                     '{"method": "medusa"}'])
    bad.model = bad.model_tag
    with pytest.raises(ValueError):
        EngineArgs.from_cli_args(bad).create_engine_config()


@pytest.mark.parametrize("frontend_mp", [True, False])
def test_server_endpoints_end_to_end(frontend_mp):
    """Every endpoint, with the engine core in its own process (the default) and
    on a thread of the server process (--disable-frontend-multiprocessing)."""
    pstate.destroy_model_parallel()
    args = build_parser().parse_args(
        ("serve llama-tiny --device cpu --dtype float32 --max-model-len 256 --block-size 16 "
         "--num-gpu-blocks-override 128 --served-model-name tiny --api-key sekret "
         "--enable-auto-tool-choice --tool-call-parser hermes --port 0 --enable-dev-endpoints"
         + ("" if frontend_mp else " --disable-frontend-multiprocessing")).split())
    args.model = args.model_tag

    async def main():
        async with api_server.build_async_engine_client(args) as engine:
            app = api_server.build_app(args)
            await api_server.init_app_state(engine, await engine.get_vllm_config(), app.state, args)
            transport = httpx.ASGITransport(app=app)
            hdr = {"Authorization": "Bearer sekret"}
            async with httpx.AsyncClient(transport=transport, base_url="http://t") as c:
                assert (await c.get("/health")).status_code == 200
                assert (await c.get("/version")).json()["version"]
                assert (await c.get("/v1/models")).status_code == 401          # key required
                # ... on everything that generates, tokenizes or administers, not only under /v1
                assert (await c.post("/invocations", json={"prompt": "x", "max_tokens": 1})).status_code == 401
                assert (await c.post("/tokenize", json={"prompt": "abc"})).status_code == 401
                assert (await c.post("/sleep")).status_code == 401
                assert (await c.get("/v1/models", headers={"Authorization": "Bearer sekre"})).status_code == 401
                # a bad request costs an error reply and nothing else, whichever process owns the core:
                # over-long and empty prompts, a negative seed, then a normal request
                too_long = await c.post("/v1/completions", headers=hdr, json={"prompt": [5] * 400, "max_tokens": 1})
                assert too_long.status_code == 400, too_long.text
                assert (await c.post("/v1/completions", headers=hdr, json={"prompt": [], "max_tokens": 1})).status_code == 400
                neg = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 3, "temperature": 1.0, "seed": -1})
                assert neg.status_code == 200, neg.text
                assert (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 3, "temperature": 1.0, "seed": -7})).status_code == 400
                big = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 3, "temperature": 1.0, "seed": 2 ** 63 - 1, "n": 2})
                assert big.status_code == 200, big.text
                # a chat template in the request body is refused unless the operator opted in
                assert (await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 1,
                    "chat_template": "{{ messages[0].content }}"})).status_code == 400
                m = (await c.get("/v1/models", headers=hdr)).json()
                assert m["data"][0]["id"] == "tiny"
                r = await c.post("/v1/completions", headers=hdr, json={
                    "model": "tiny", "prompt": "hello", "max_tokens": 5, "temperature": 0, "logprobs": 1})
                assert r.status_code == 200, r.text
                body = r.json()
                assert body["usage"]["completion_tokens"] == 5 and body["choices"][0]["finish_reason"] == "length"
                assert len(body["choices"][0]["logprobs"]["token_logprobs"]) == 5
                # n samples per prompt: one choice each, reproducible under a seed, prompt counted once
                r = await c.post("/v1/completions", headers=hdr, json={
                    "model": "tiny", "prompt": ["hello", "bye"], "max_tokens": 4, "temperature": 1.0, "seed": 3, "ignore_eos": True,
                    "n": 3})
                body = r.json()
                assert r.status_code == 200 and [ch["index"] for ch in body["choices"]] == list(range(6))
                assert body["usage"]["completion_tokens"] == 24
                r2 = await c.post("/v1/completions", headers=hdr, json={
                    "model": "tiny", "prompt": ["hello", "bye"], "max_tokens": 4, "temperature": 1.0, "seed": 3, "ignore_eos": True,
                    "n": 3})
                assert [ch["text"] for ch in r2.json()["choices"]] == [ch["text"] for ch in body["choices"]]
                r = await c.post("/v1/chat/completions", headers=hdr, json={
                    "model": "tiny", "messages": [{"role": "user", "content": "hi"}], "max_tokens": 3, "n": 2})
                assert r.status_code == 200 and len(r.json()["choices"]) == 2
                # best_of: five seeded samples, the two most probable returned; usage counts all five;
                # the same five come back (in sample order) from n = 5 with the same seed
                r = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 4, "temperature": 1.0, "seed": 11, "ignore_eos": True,
                    "n": 2, "best_of": 5})
                assert r.status_code == 200, r.text
                best = r.json()
                assert len(best["choices"]) == 2 and best["usage"]["completion_tokens"] == 20
                allfive = (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 4, "temperature": 1.0, "seed": 11, "ignore_eos": True,
                    "n": 5, "logprobs": 0})).json()
                ranked = sorted(allfive["choices"], key=lambda ch: -sum(ch["logprobs"]["token_logprobs"]))
                assert [ch["text"] for ch in best["choices"]] == [ch["text"] for ch in ranked[:2]]
                assert (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 2, "n": 2, "best_of": 4, "stream": True})).status_code == 400
                assert (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 2, "n": 3, "best_of": 2})).status_code == 400
                assert (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 2, "best_of": 100000})).status_code == 400
                assert (await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 1, "n": 5000})).status_code == 400
                # beam search: n is the beam width, beams come back best first; width 1 is greedy decoding
                greedy = (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": [1, 2, 3, 4], "max_tokens": 4, "temperature": 0, "ignore_eos": True, "logprobs": 0})).json()
                b1 = (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": [1, 2, 3, 4], "max_tokens": 4, "use_beam_search": True, "n": 1, "ignore_eos": True})).json()
                # (texts differ only in how a dangling partial UTF-8 sequence is rendered: compare the scores)
                assert abs(b1["choices"][0]["cumulative_logprob"]
                           - sum(greedy["choices"][0]["logprobs"]["token_logprobs"])) < 1e-3
                r = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": [1, 2, 3, 4], "max_tokens": 4, "use_beam_search": True, "n": 3, "ignore_eos": True})
                assert r.status_code == 200, r.text
                beams = r.json()["choices"]
                scores = [ch["cumulative_logprob"] for ch in beams]
                assert len(beams) == 3 and scores == sorted(scores, reverse=True)
                assert scores[0] >= b1["choices"][0]["cumulative_logprob"] - 1e-6
                r = await c.post("/v1/chat/completions", headers=hdr, json={
                    "model": "tiny", "messages": [{"role": "user", "content": "hi"}], "max_tokens": 3,
                    "use_beam_search": True, "n": 2})
                assert r.status_code == 200 and len(r.json()["choices"]) == 2, r.text
                r2 = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": [1, 2, 3, 4], "max_tokens": 4, "temperature": 0, "stream": True,
                    "stream_options": {"include_usage": True}})
                lines = [l for l in r2.text.split("\n\n") if l.startswith("data: ")]
                assert lines[-1] == "data: [DONE]"
                usage = json.loads(lines[-2][6:])["usage"]
                assert usage["completion_tokens"] == 4
                r3 = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 6, "temperature": 0})
                assert r3.status_code == 200, r3.text
                assert r3.json()["choices"][0]["message"]["role"] == "assistant"
                assert r3.json()["usage"]["completion_tokens"] == 6
                r4 = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 4, "stream": True, "seed": 3})
                chunks = [json.loads(l[6:]) for l in r4.text.split("\n\n")
                          if l.startswith("data: ") and l != "data: [DONE]"]
                assert chunks[0]["choices"][0]["delta"]["role"] == "assistant"
                assert chunks[-1]["choices"][0]["finish_reason"] == "length"
                r5 = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 4, "stream": True, "seed": 3,
                    "n": 3, "ignore_eos": True, "stream_options": {"include_usage": True}})
                ev = [json.loads(l[6:]) for l in r5.text.split("\n\n") if l.startswith("data: ") and l != "data: [DONE]"]
                per = {k: [e for e in ev if e["choices"] and e["choices"][0]["index"] == k] for k in range(3)}
                assert all(per[k][0]["choices"][0]["delta"].get("role") == "assistant" for k in range(3))
                assert all(per[k][-1]["choices"][0]["finish_reason"] == "length" for k in range(3))
                assert ev[-1]["usage"]["completion_tokens"] == 12
                assert (await c.post("/v1/completions", headers=hdr,
                                     json={"model": "other", "prompt": "x"})).status_code == 404
                # logit edits travel with the request: a +100 bias pins every token, the allow list bounds them
                r6 = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "hello", "max_tokens": 4, "temperature": 0, "ignore_eos": True, "logprobs": 0,
                    "logit_bias": {"65": 100}})
                assert r6.status_code == 200, r6.text
                assert r6.json()["choices"][0]["text"] == "AAAA"
                r7 = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 6, "temperature": 0,
                    "ignore_eos": True, "allowed_token_ids": [66, 67], "bad_words": ["B"]})
                assert r7.status_code == 200, r7.text
                assert r7.json()["choices"][0]["message"]["content"] == "CCCCCC"
                r8 = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "x" * 50, "max_tokens": 2, "temperature": 0, "truncate_prompt_tokens": 8})
                assert r8.status_code == 200 and r8.json()["usage"]["prompt_tokens"] == 8
                bad = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "x", "max_tokens": 2, "logit_bias": {"99999999": 5}})
                assert bad.status_code == 400, bad.text
                # echo + logprobs scores the prompt too (the evaluation-harness form); prompt_logprobs is vLLM's field
                r9 = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": [1, 2, 3, 4, 5], "max_tokens": 1, "temperature": 0, "echo": True, "logprobs": 1})
                lp = r9.json()["choices"][0]["logprobs"]
                assert r9.status_code == 200 and len(lp["token_logprobs"]) == 6 and lp["token_logprobs"][0] is None
                assert all(isinstance(v, float) for v in lp["token_logprobs"][1:])
                # JSON mode: the content parses as an object whatever the (random) model prefers
                rj = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 200, "seed": 1,
                    "response_format": {"type": "json_object"},
                    "logit_bias": {str(ord('"')): 4, str(ord("}")): 5, str(ord(":")): 3}})
                assert rj.status_code == 200, rj.text
                if rj.json()["choices"][0]["finish_reason"] == "stop":
                    assert isinstance(json.loads(rj.json()["choices"][0]["message"]["content"]), dict)
                else:
                    assert rj.json()["choices"][0]["message"]["content"].lstrip().startswith("{")
                assert (await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}],
                    "response_format": {"type": "json_schema", "json_schema": {}}})).status_code == 400
                # regular-expression and JSON-schema constraints (a DFA over the vocabulary's bytes)
                rr = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "x", "guided_regex": "(yes|no)!{1,2}", "max_tokens": 16, "seed": 3})
                assert rr.status_code == 200, rr.text
                assert rr.json()["choices"][0]["text"] in ("yes!", "yes!!", "no!", "no!!")
                assert rr.json()["choices"][0]["finish_reason"] == "stop"
                schema = {"type": "object", "required": ["a", "b"],
                          "properties": {"a": {"type": "integer"}, "b": {"enum": ["x", "y"]}, "c": {"type": "boolean"}}}
                for body in ({"response_format": {"type": "json_schema", "json_schema": {"name": "t", "schema": schema}}},
                             {"guided_json": schema}, {"structured_outputs": {"json": schema}}):
                    rg = await c.post("/v1/chat/completions", headers=hdr, json=dict(
                        body, messages=[{"role": "user", "content": "hi"}], max_tokens=64, seed=4))
                    assert rg.status_code == 200, rg.text
                    got = json.loads(rg.json()["choices"][0]["message"]["content"])
                    assert isinstance(got["a"], int) and got["b"] in ("x", "y") and set(got) <= {"a", "b", "c"}
                gg = await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "x", "max_tokens": 24, "seed": 5,
                    "guided_grammar": 'root ::= "<" [a-c]{2,4} ">" | "(" root ")"'})
                assert gg.status_code == 200, gg.text
                txt = gg.json()["choices"][0]["text"]
                assert gg.json()["choices"][0]["finish_reason"] == "stop" and txt.count("(") == txt.count(")")
                assert 4 <= len(txt.strip("()")) <= 6 and txt.strip("()")[0] == "<" and txt.strip("()")[-1] == ">"
                for bad_body in ({"guided_regex": "(unclosed"}, {"guided_grammar": "root ::= x"},
                                 {"guided_regex": "a", "guided_choice": ["a"]},
                                 {"guided_json": {"type": "object", "properties": {"a": {"$ref": "http://x/y"}}}}):
                    assert (await c.post("/v1/completions", headers=hdr, json=dict(
                        bad_body, prompt="x"))).status_code == 400, bad_body
                assert (await c.post("/v1/completions", headers=hdr, json={
                    "prompt": "x", "logprobs": 5000})).status_code == 400       # --max-logprobs (20)
                gc = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 16, "temperature": 1.0, "seed": 2,
                    "guided_choice": ["positive", "negative", "neutral"]})
                assert gc.status_code == 200, gc.text
                assert gc.json()["choices"][0]["message"]["content"] in ("positive", "negative", "neutral")
                # a named tool choice: the reply is that call, its arguments one JSON object (streamed as fragments)
                tool = {"type": "function", "function": {"name": "get_weather", "parameters": {"type": "object"}}}
                named = {"messages": [{"role": "user", "content": "hi"}], "max_tokens": 200, "seed": 1, "tools": [tool],
                         "tool_choice": {"type": "function", "function": {"name": "get_weather"}},
                         "logit_bias": {str(ord('"')): 4, str(ord("}")): 5, str(ord(":")): 3}}
                rt = (await c.post("/v1/chat/completions", headers=hdr, json=named)).json()
                call = rt["choices"][0]["message"]["tool_calls"][0]
                assert call["function"]["name"] == "get_weather" and rt["choices"][0]["message"]["content"] is None
                if rt["choices"][0]["finish_reason"] == "tool_calls":
                    assert isinstance(json.loads(call["function"]["arguments"]), dict)
                rs = await c.post("/v1/chat/completions", headers=hdr, json=dict(named, stream=True))
                evs = [json.loads(l[6:]) for l in rs.text.split("\n\n") if l.startswith("data: ") and l != "data: [DONE]"]
                frags = [tc for e in evs for tc in e["choices"][0]["delta"].get("tool_calls", [])]
                assert frags[0]["function"]["name"] == "get_weather"
                assert "".join(f["function"].get("arguments", "") for f in frags) == call["function"]["arguments"]
                assert (await c.post("/v1/chat/completions", headers=hdr, json=dict(
                    named, tool_choice={"type": "function", "function": {"name": "nope"}}))).status_code == 400
                # tool_choice "required": the reply is a list of calls, arguments following each tool's schema
                tool2 = {"type": "function", "function": {"name": "add", "parameters": {
                    "type": "object", "required": ["x"], "properties": {"x": {"type": "integer"}}}}}
                rq = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 200, "seed": 5,
                    "tools": [tool2], "tool_choice": "required", "logit_bias": {str(ord("]")): 6}})
                assert rq.status_code == 200, rq.text
                ch = rq.json()["choices"][0]
                assert ch["finish_reason"] == "tool_calls" and ch["message"]["content"] is None
                assert ch["message"]["tool_calls"] and all(
                    tc["function"]["name"] == "add" and isinstance(json.loads(tc["function"]["arguments"])["x"], int)
                    for tc in ch["message"]["tool_calls"])
                rqs = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 200, "seed": 5, "stream": True,
                    "tools": [tool2], "tool_choice": "required", "logit_bias": {str(ord("]")): 6}})
                evq = [json.loads(l[6:]) for l in rqs.text.split("\n\n") if l.startswith("data: ") and l != "data: [DONE]"]
                assert any(e["choices"][0]["delta"].get("tool_calls") for e in evq)
                assert evq[-1]["choices"][0]["finish_reason"] == "tool_calls"
                assert (await c.post("/v1/chat/completions", headers=hdr, json=dict(
                    named, tools=[], tool_choice="required"))).status_code == 400
                # embeddings / pooling / scoring: a prefill whose pooled hidden state comes back
                em = await c.post("/v1/embeddings", headers=hdr, json={"input": ["hello world", "hello world", "zq"]})
                assert em.status_code == 200, em.text
                ed = em.json()["data"]
                assert [d["index"] for d in ed] == [0, 1, 2] and ed[0]["object"] == "embedding"
                assert ed[0]["embedding"] == ed[1]["embedding"] != ed[2]["embedding"]
                assert abs(sum(x * x for x in ed[0]["embedding"]) - 1.0) < 1e-4
                assert em.json()["usage"]["prompt_tokens"] > 0
                import base64
                import struct
                eb = await c.post("/v1/embeddings", headers=hdr, json={
                    "input": "hello world", "encoding_format": "base64", "dimensions": 8})
                raw = base64.b64decode(eb.json()["data"][0]["embedding"])
                assert len(raw) == 32 and abs(sum(x * x for x in struct.unpack("<8f", raw)) - 1.0) < 1e-4
                pm = await c.post("/pooling", headers=hdr, json={"input": [[1, 2, 3, 4]], "pooling_type": "mean"})
                assert pm.status_code == 200 and pm.json()["data"][0]["object"] == "pooling"
                assert abs(sum(x * x for x in pm.json()["data"][0]["data"]) - 1.0) > 1e-3      # raw, not normalised
                sc = await c.post("/score", headers=hdr, json={"text_1": "hello", "text_2": ["hello", "zzz other"]})
                assert sc.status_code == 200, sc.text
                scores = [d["score"] for d in sc.json()["data"]]
                assert abs(scores[0] - 1.0) < 1e-4 and scores[1] < 0.999
                assert (await c.post("/v1/score", headers=hdr, json={
                    "text_1": ["a", "b", "c"], "text_2": ["a", "b"]})).status_code == 400
                assert (await c.post("/v1/embeddings", headers=hdr, json={"input": []})).status_code == 400
                assert (await c.post("/v1/embeddings", json={"input": "x"})).status_code == 401
                # the Responses API rides on the chat pipeline
                rp = await c.post("/v1/responses", headers=hdr, json={
                    "input": "hi", "instructions": "be brief", "max_output_tokens": 5, "temperature": 0})
                assert rp.status_code == 200, rp.text
                ro = rp.json()
                assert ro["object"] == "response" and ro["status"] in ("completed", "incomplete")
                assert ro["output"][0]["type"] == "message" and ro["output"][0]["content"][0]["type"] == "output_text"
                assert ro["output_text"] == ro["output"][0]["content"][0]["text"]
                assert ro["usage"]["output_tokens"] == 5 and ro["usage"]["input_tokens"] > 0
                assert (await c.get(f"/v1/responses/{ro['id']}", headers=hdr)).json()["output"] == ro["output"]
                r2 = await c.post("/v1/responses", headers=hdr, json={
                    "input": [{"type": "message", "role": "user", "content": [{"type": "input_text", "text": "more"}]}],
                    "previous_response_id": ro["id"], "max_output_tokens": 3, "temperature": 0})
                assert r2.status_code == 200 and r2.json()["usage"]["input_tokens"] > ro["usage"]["input_tokens"]
                rs = await c.post("/v1/responses", headers=hdr, json={
                    "input": "hi", "max_output_tokens": 5, "temperature": 0, "stream": True, "ignore_eos": True})
                kinds = [l.split("\n")[0][7:] for l in rs.text.split("\n\n") if l.startswith("event: ")]
                assert kinds[0] == "response.created" and kinds[-1] in ("response.completed", "response.incomplete")
                assert "response.output_text.delta" in kinds and "response.output_item.done" in kinds
                last = json.loads(rs.text.split("\n\n")[-2].split("\n")[1][6:])
                assert last["response"]["usage"]["output_tokens"] == 5
                deltas = [json.loads(l.split("\n")[1][6:])["delta"] for l in rs.text.split("\n\n")
                          if l.startswith("event: response.output_text.delta")]
                assert "".join(deltas) == last["response"]["output_text"]
                rt2 = await c.post("/v1/responses", headers=hdr, json={
                    "input": "hi", "max_output_tokens": 120, "seed": 5, "tool_choice": "required",
                    "tools": [{"type": "function", "name": "add", "parameters": {
                        "type": "object", "required": ["x"], "properties": {"x": {"type": "integer"}}}}],
                    "logit_bias": {str(ord("]")): 6}})
                assert rt2.status_code == 200, rt2.text
                assert any(it["type"] == "function_call" and it["name"] == "add" for it in rt2.json()["output"])
                assert (await c.delete(f"/v1/responses/{ro['id']}", headers=hdr)).json()["deleted"] is True
                assert (await c.get(f"/v1/responses/{ro['id']}", headers=hdr)).status_code == 404
                assert (await c.post("/v1/responses", headers=hdr, json={
                    "input": "x", "previous_response_id": "resp_nope"})).status_code == 400
                r10 = await c.post("/v1/chat/completions", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 2, "prompt_logprobs": 0})
                pl = r10.json()["prompt_logprobs"]
                assert r10.status_code == 200 and pl[0] is None and len(pl) == r10.json()["usage"]["prompt_tokens"]
                assert all(len(d) == 1 and "logprob" in next(iter(d.values())) for d in pl[1:])
                t = (await c.post("/tokenize", headers=hdr, json={"prompt": "abc"})).json()
                assert t["count"] == len(t["tokens"])
                d = (await c.post("/detokenize", headers=hdr, json={"tokens": t["tokens"]})).json()
                assert "abc" in d["prompt"]
                met = (await c.get("/metrics")).text
                assert "vllm:generation_tokens_total" in met and "vllm:time_to_first_token_seconds" in met
                assert (await c.post("/reset_prefix_cache", headers=hdr)).status_code == 200
                inv = await c.post("/invocations", headers=hdr, json={"prompt": "hello", "max_tokens": 3, "temperature": 0})
                assert inv.status_code == 200 and inv.json()["object"] == "text_completion"
                inv = await c.post("/invocations", headers=hdr, json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 3})
                assert inv.status_code == 200 and inv.json()["object"] == "chat.completion"
                assert (await c.post("/invocations", headers=hdr, json={"x": 1})).status_code == 400
                assert (await c.get("/is_sleeping", headers=hdr)).json() == {"is_sleeping": False}
                # browser clients: the CORS preflight is answered without a key, responses carry the allow header
                pre = await c.options("/v1/chat/completions", headers={
                    "Origin": "http://ui.example", "Access-Control-Request-Method": "POST",
                    "Access-Control-Request-Headers": "authorization,content-type"})
                assert pre.status_code == 200 and pre.headers["access-control-allow-origin"] == "*"
                got = await c.get("/v1/models", headers=dict(hdr, Origin="http://ui.example"))
                assert got.headers["access-control-allow-origin"] == "*"
    asyncio.run(main())
    pstate.destroy_model_parallel()


def test_chat_and_complete_clients_stream_from_a_server():
    """The terminal clients: model discovery, bearer token, SSE parsing for
    both endpoints, conversation history in chat mode."""
    import io
    from vllm_distributed_amd.entrypoints import client
    seen = []

    def handler(request: httpx.Request) -> httpx.Response:
        assert request.headers["authorization"] == "Bearer sekret"
        if request.url.path == "/v1/models":
            return httpx.Response(200, json={"data": [{"id": "tiny"}]})
        body = json.loads(request.content)
        seen.append((request.url.path, body))
        assert body["model"] == "tiny" and body["stream"] is True
        if request.url.path == "/v1/completions":
            events = [{"choices": [{"index": 0, "text": t}]} for t in (" wor", "ld")]
        else:
            events = [{"choices": [{"index": 0, "delta": {"role": "assistant"}}]},
                      {"choices": [{"index": 0, "delta": {"content": "hi "}}]},
                      {"choices": [{"index": 0, "delta": {"content": "there"}, "finish_reason": "stop"}]}]
        sse = "".join(f"data: {json.dumps(e)}\n\n" for e in events) + "data: [DONE]\n\n"
        return httpx.Response(200, content=sse.encode(), headers={"content-type": "text/event-stream"})

    transport = httpx.MockTransport(handler)
    a = build_parser().parse_args("complete --url http://t/v1 --api-key sekret -q hello --max-tokens 7".split())
    out = io.StringIO()
    a.dispatch_function(a, transport=transport, out=out)
    assert out.getvalue() == " world\n"
    assert seen[-1][0] == "/v1/completions" and seen[-1][1]["prompt"] == "hello" and seen[-1][1]["max_tokens"] == 7
    a = build_parser().parse_args(["chat", "--url", "http://t/v1", "--api-key", "sekret", "-q", "hey",
                                   "--system-prompt", "be brief"])
    out = io.StringIO()
    a.dispatch_function(a, transport=transport, out=out)
    assert out.getvalue() == "hi there\n"
    assert [m["role"] for m in seen[-1][1]["messages"]] == ["system", "user"]


@pytest.mark.slow
def test_real_socket_server_with_bench_serve_and_clients():
    """`launch.py serve` on a real port (uvicorn, engine core in its own process),
    driven by `launch.py bench serve` and the `complete` client."""
    import os
    import subprocess
    import sys
    import time
    from vllm_distributed_amd.utils import get_open_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = get_open_port()
    env = dict(os.environ, VLLM_LOGGING_LEVEL="WARNING", OMP_NUM_THREADS="2")
    server = subprocess.Popen(
        [sys.executable, os.path.join(root, "launch.py"), "serve", "llama-tiny", "--device", "cpu", "--dtype", "float32",
         "--max-model-len", "256", "--block-size", "16", "--num-gpu-blocks-override", "256", "--host", "127.0.0.1",
         "--port", str(port), "--served-model-name", "tiny", "--uvicorn-log-level", "warning"],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        url = f"http://127.0.0.1:{port}"
        deadline = time.time() + 120
        while True:
            try:
                if httpx.get(url + "/health", timeout=2).status_code == 200:
                    break
            except httpx.HTTPError:
                pass
            assert server.poll() is None, server.stdout.read()[-3000:]
            assert time.time() < deadline, "server did not come up"
            time.sleep(0.5)
        r = subprocess.run([sys.executable, os.path.join(root, "launch.py"), "bench", "serve", "--base-url", url,
                            "--num-prompts", "8", "--concurrency", "4", "--input-len", "16", "--output-len", "8",
                            "--vocab-size", "500"], env=env, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["num_prompts"] == 8 and j["output_tokens_per_s"] > 0 and j["ttft_p50_ms"] > 0
        r = subprocess.run([sys.executable, os.path.join(root, "launch.py"), "complete", "--url", url + "/v1",
                            "-q", "hello", "--max-tokens", "5"], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    finally:
        server.terminate()
        try:
            server.wait(timeout=30)
        except subprocess.TimeoutExpired:
            server.kill()


def test_generation_config_supplies_defaults_and_extra_eos_ids(tmp_path):
    """generation_config.json next to the weights: its sampling fields are the defaults for requests
    that leave them unset, every id in its eos_token_id list ends a sequence, and
    ``--generation-config vllm`` turns both off."""
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy")
    save_dummy_checkpoint(mc, str(tmp_path))
    (tmp_path / "config.json").write_text(json.dumps(mc.hf_config))
    (tmp_path / "generation_config.json").write_text(json.dumps(
        {"do_sample": True, "temperature": 0.0, "top_k": 5, "max_new_tokens": 7, "eos_token_id": [2, 77]}))
    mc2 = ModelConfig(model=str(tmp_path), dtype="float32")
    assert mc2.default_sampling_params() == {"temperature": 0.0, "top_k": 5, "max_tokens": 7}
    assert 77 in mc2.eos_token_ids()
    assert ModelConfig(model=str(tmp_path), dtype="float32", generation_config="vllm").default_sampling_params() == {}

    def serve(extra: str):
        pstate.destroy_model_parallel()
        args = build_parser().parse_args(
            (f"serve {tmp_path} --device cpu --dtype float32 --max-model-len 128 --block-size 16 "
             f"--num-gpu-blocks-override 64 --served-model-name tiny --port 0 "
             f"--disable-frontend-multiprocessing{extra}").split())
        args.model = args.model_tag

        async def main():
            async with api_server.build_async_engine_client(args) as engine:
                app = api_server.build_app(args)
                await api_server.init_app_state(engine, await engine.get_vllm_config(), app.state, args)
                async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
                    a = (await c.post("/v1/chat/completions", json={
                        "messages": [{"role": "user", "content": "hi"}], "ignore_eos": True})).json()
                    b = (await c.post("/v1/chat/completions", json={
                        "messages": [{"role": "user", "content": "hi"}], "ignore_eos": True})).json()
                    stop = (await c.post("/v1/completions", json={
                        "prompt": "hello", "max_tokens": 6, "logit_bias": {"77": 100}})).json()
                    return a, b, stop
        out = asyncio.run(main())
        pstate.destroy_model_parallel()
        return out

    a, b, stop = serve("")
    # greedy by default and capped at 7 new tokens, both from the file
    assert a["usage"]["completion_tokens"] == 7 and a["choices"][0]["message"] == b["choices"][0]["message"]
    assert stop["choices"][0]["finish_reason"] == "stop" and stop["usage"]["completion_tokens"] == 1
    a, b, stop = serve(" --generation-config vllm")
    assert a["usage"]["completion_tokens"] > 7              # runs to the context limit: no cap from the file
    assert stop["choices"][0]["finish_reason"] == "length" and stop["usage"]["completion_tokens"] == 6


def test_reasoning_splitters_are_chunking_invariant():
    import random
    from vllm_distributed_amd.entrypoints.openai.reasoning import (DeepSeekR1ReasoningSplitter as D,
                                                                   Qwen3ReasoningSplitter as Q)
    cases = [("<think>abc</think>hello", ("abc", "hello"), ("abc", "hello")),
             ("  \n<think>a<b</thin</think>\n\nhi <think>", ("a<b</thin", "\n\nhi <think>"), ("a<b</thin", "\n\nhi <think>")),
             ("plain answer", ("", "plain answer"), ("plain answer", "")),
             ("abc</think>rest", ("", "abc</think>rest"), ("abc", "rest")),
             ("<think>never ends", ("never ends", ""), ("never ends", "")),
             ("<thi", ("", "<thi"), ("<thi", "")), ("", ("", ""), ("", ""))]
    rnd = random.Random(0)
    for text, want_q, want_d in cases:
        for cls, want in ((Q, want_q), (D, want_d)):
            assert cls.split(text) == want, (cls.__name__, text)
            for _ in range(40):             # any way of cutting the text into deltas gives the same split
                s, r, c, i = cls(), "", "", 0
                while i < len(text):
                    k = rnd.randint(1, 4)
                    a, b = s.feed(text[i:i + k])
                    r, c, i = r + a, c + b, i + k
                a, b = s.flush()
                assert (r + a, c + b) == want, (cls.__name__, text)


def test_reasoning_parser_routes_chat_output_to_reasoning_content():
    """--reasoning-parser deepseek_r1: text before </think> is reasoning_content.  A random-init model never
    closes the block, so the whole completion lands there, streamed and not."""
    pstate.destroy_model_parallel()
    args = build_parser().parse_args(
        ("serve llama-tiny --device cpu --dtype float32 --max-model-len 128 --block-size 16 "
         "--num-gpu-blocks-override 64 --port 0 --disable-frontend-multiprocessing "
         "--reasoning-parser deepseek_r1").split())
    args.model = args.model_tag

    async def main():
        async with api_server.build_async_engine_client(args) as engine:
            app = api_server.build_app(args)
            await api_server.init_app_state(engine, await engine.get_vllm_config(), app.state, args)
            async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
                body = {"messages": [{"role": "user", "content": "hi"}], "max_tokens": 12, "temperature": 0,
                        "ignore_eos": True, "allowed_token_ids": list(range(97, 123))}
                r = (await c.post("/v1/chat/completions", json=body)).json()
                msg = r["choices"][0]["message"]
                assert msg["content"] is None and len(msg["reasoning_content"]) == 12
                s = await c.post("/v1/chat/completions", json=dict(body, stream=True))
                ev = [json.loads(l[6:]) for l in s.text.split("\n\n") if l.startswith("data: ") and l != "data: [DONE]"]
                streamed = "".join(e["choices"][0]["delta"].get("reasoning_content", "") for e in ev)
                assert streamed == msg["reasoning_content"]
                assert not any(e["choices"][0]["delta"].get("content") for e in ev)
    asyncio.run(main())
    pstate.destroy_model_parallel()


def test_chat_completions_with_image_parts():
    """``image_url`` content parts of a chat request reach a vision-language model as images: a base64 data URL
    is decoded, its placeholder is widened to the tower's 16 embeddings (usage counts them), two different
    pictures give two different replies and the same picture the same reply; remote URLs and models without a
    vision tower answer 400."""
    import base64
    import io
    import numpy as np
    from PIL import Image
    pstate.destroy_model_parallel()

    def data_url(seed):
        img = Image.fromarray((np.random.default_rng(seed).random((40, 50, 3)) * 255).astype("uint8"))
        buf = io.BytesIO()
        img.save(buf, format="PNG")
        return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()

    def body(url, text="what is this?"):
        return {"messages": [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": url}},
                                                          {"type": "text", "text": text}]}],
                "max_tokens": 8, "temperature": 0, "ignore_eos": True}

    async def main(model, expect_images, image_tokens=16):
        args = build_parser().parse_args(
            (f"serve {model} --device cpu --dtype float32 --max-model-len 256 --block-size 16 "
             "--num-gpu-blocks-override 64 --port 0 --disable-frontend-multiprocessing").split())
        args.model = args.model_tag
        async with api_server.build_async_engine_client(args) as engine:
            app = api_server.build_app(args)
            await api_server.init_app_state(engine, await engine.get_vllm_config(), app.state, args)
            async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
                if not expect_images:
                    r = await c.post("/v1/chat/completions", json=body(data_url(1)))
                    assert r.status_code == 400 and "does not take images" in r.text
                    return
                assert app.state.image_placeholder == "<511>"
                a = (await c.post("/v1/chat/completions", json=body(data_url(1)))).json()
                b = (await c.post("/v1/chat/completions", json=body(data_url(2)))).json()
                a2 = (await c.post("/v1/chat/completions", json=body(data_url(1)))).json()
                plain = (await c.post("/v1/chat/completions", json={
                    "messages": [{"role": "user", "content": "what is this?"}], "max_tokens": 8, "temperature": 0,
                    "ignore_eos": True})).json()
                text = [x["choices"][0]["message"]["content"] for x in (a, b, a2)]
                assert text[0] == text[2] and text[0] != text[1]
                # one placeholder line ("<511>" + newline) became 16 image positions + the newline
                # (an any-resolution model: as many as the 40 x 50 picture's aspect leaves of its tile grid)
                assert a["usage"]["prompt_tokens"] == plain["usage"]["prompt_tokens"] + image_tokens + 1
                r = await c.post("/v1/chat/completions", json=body("https://example.com/cat.png"))
                assert r.status_code == 400 and "data: URL" in r.text
                r = await c.post("/v1/chat/completions", json=body("data:image/png;base64,AAAA"))
                assert r.status_code == 400 and "undecodable" in r.text
                t = (await c.post("/tokenize", json={"messages": body(data_url(1))["messages"]})).json()
                assert t["tokens"].count(511) == 1
    asyncio.run(main("llava-tiny", True))
    pstate.destroy_model_parallel()
    from vllm_distributed_amd.multimodal import anyres_tokens
    n = anyres_tokens({"vision_config": {"image_size": 28, "patch_size": 7},
                       "image_grid_pinpoints": [[28, 56], [56, 28], [56, 56], [84, 28], [28, 84]]}, (40, 50))
    assert n > 16
    asyncio.run(main("llava-next-tiny", True, n))
    pstate.destroy_model_parallel()
    asyncio.run(main("llama-tiny", False))
    pstate.destroy_model_parallel()


def test_chat_templates_render_in_a_sandbox():
    """Templates come from model repositories and, with --trust-request-chat-template, from
    request bodies: attribute walks into Python internals must fail, ordinary templates work."""
    import jinja2
    from vllm_distributed_amd.tokenizer import render_chat_template
    msgs = [{"role": "user", "content": "hi"}]
    assert render_chat_template("{% for m in messages %}{{ m.role }}:{{ m.content }}{% endfor %}", msgs) == "user:hi"
    for evil in ("{{ ''.__class__.__mro__[1].__subclasses__() }}",
                 "{{ messages.__class__.__base__.__subclasses__() }}",
                 "{{ cycler.__init__.__globals__.os.popen('id').read() }}",
                 "{% set _ = messages.append({'role': 'x'}) %}ok"):
        with pytest.raises((jinja2.exceptions.SecurityError, jinja2.exceptions.UndefinedError)):
            render_chat_template(evil, msgs)
    assert msgs == [{"role": "user", "content": "hi"}]


def test_admin_endpoints_need_the_dev_flag_and_request_templates_need_trust():
    pstate.destroy_model_parallel()
    base = ("serve llama-tiny --device cpu --dtype float32 --max-model-len 128 --block-size 16 "
            "--num-gpu-blocks-override 64 --port 0 --disable-frontend-multiprocessing")
    args = build_parser().parse_args((base + " --trust-request-chat-template").split())
    args.model = args.model_tag

    async def main():
        async with api_server.build_async_engine_client(args) as engine:
            app = api_server.build_app(args)
            await api_server.init_app_state(engine, await engine.get_vllm_config(), app.state, args)
            async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
                for path in ("/sleep", "/wake_up", "/start_profile", "/stop_profile", "/reset_prefix_cache"):
                    assert (await c.post(path)).status_code == 404, path
                ok = await c.post("/v1/chat/completions", json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 1,
                    "chat_template": "{{ messages[0].content }}"})
                assert ok.status_code == 200, ok.text
                bad = await c.post("/v1/chat/completions", json={
                    "messages": [{"role": "user", "content": "hi"}], "max_tokens": 1,
                    "chat_template": "{{ ''.__class__.__mro__[1].__subclasses__() }}"})
                assert bad.status_code == 400 and "template" in bad.text
    asyncio.run(main())
    pstate.destroy_model_parallel()


def test_tool_parsers_of_the_other_model_families():
    import json as _json
    get = lambda n: ToolParserManager.get_tool_parser(n)()                       # noqa: E731
    args = lambda c: _json.loads(c["function"]["arguments"])                      # noqa: E731
    # Mistral: a JSON list after the marker, or NAME{json} per marker
    content, calls = get("mistral").extract('Sure.[TOOL_CALLS][{"name": "add", "arguments": {"a": 1, "b": 2}}]')
    assert content == "Sure." and calls[0]["function"]["name"] == "add" and args(calls[0]) == {"a": 1, "b": 2}
    content, calls = get("mistral").extract('[TOOL_CALLS]get_weather{"city": "Paris"}[TOOL_CALLS]now{}')
    assert content is None and [c["function"]["name"] for c in calls] == ["get_weather", "now"]
    assert get("mistral").extract("no call here") == ("no call here", [])
    # DeepSeek-V3
    t = ("Let me check.<｜tool▁calls▁begin｜><｜tool▁call▁begin｜>function"
         "<｜tool▁sep｜>get_weather\n```json\n{\"city\": \"Oslo\"}\n```<｜tool▁call▁end｜>"
         "<｜tool▁calls▁end｜>")
    content, calls = get("deepseek_v3").extract(t)
    assert content == "Let me check." and calls[0]["function"]["name"] == "get_weather" and args(calls[0]) == {"city": "Oslo"}
    # GLM-4.5: key / value tags, typed by the schema
    tools = [{"type": "function", "function": {"name": "book", "parameters": {"type": "object", "properties": {
        "nights": {"type": "integer"}, "city": {"type": "string"}, "late": {"type": "boolean"}}}}}]
    t = ("<tool_call>book\n<arg_key>nights</arg_key>\n<arg_value>3</arg_value>\n<arg_key>city</arg_key>\n"
         "<arg_value>Rome</arg_value>\n<arg_key>late</arg_key>\n<arg_value>true</arg_value>\n</tool_call>")
    content, calls = get("glm45").extract(t, tools)
    assert content is None and args(calls[0]) == {"nights": 3, "city": "Rome", "late": True}
    # Granite, Phi-4-mini
    _, calls = get("granite").extract('<|tool_call|>[{"name": "f", "arguments": {"x": 1}}]')
    assert calls[0]["function"]["name"] == "f" and args(calls[0]) == {"x": 1}
    _, calls = get("granite").extract('<tool_call>[{"name": "g", "arguments": {}}]')
    assert calls[0]["function"]["name"] == "g"
    content, calls = get("phi4_mini_json").extract('ok functools[{"name": "h", "arguments": {"q": "a b"}}]')
    assert content == "ok" and args(calls[0]) == {"q": "a b"}
    # pythonic: literals only; anything else is plain text
    _, calls = get("pythonic").extract('[get_weather(city="Paris", days=3), tools.now()]')
    assert [c["function"]["name"] for c in calls] == ["get_weather", "tools.now"] and args(calls[0]) == {"city": "Paris", "days": 3}
    bad = '[get_weather(city=__import__("os").getcwd())]'
    assert get("pythonic").extract(bad) == (bad, [])
    assert get("pythonic").extract("[1, 2, 3]") == ("[1, 2, 3]", [])
