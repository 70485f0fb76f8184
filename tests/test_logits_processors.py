"""Per-request logit edits: logit_bias, allowed_token_ids, bad_words, min_tokens and
prompt truncation, through the engine on CPU (llama-tiny, byte tokenizer) and over
HTTP.  These mirror the OpenAI / vLLM request fields of the same names, which the
reference serves through vLLM's sampler (reference src/launch.py:413-457 brings up
that app unchanged)."""
import pytest

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.parallel import state as pstate

KW = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
          num_gpu_blocks_override=64)


@pytest.fixture(scope="module")
def llm():
    pstate.destroy_model_parallel()
    m = LLM(**KW)
    yield m
    m.shutdown()
    pstate.destroy_model_parallel()


def _gen(llm, sp, prompt="hello world"):
    return llm.generate([prompt], sp)[0]


def test_sampling_params_validation():
    sp = SamplingParams(logit_bias={"5": 500, 7: -3.5})
    assert sp.logit_bias == {5: 100.0, 7: -3.5} and sp.has_logit_adjust
    assert not SamplingParams().has_logit_adjust
    assert not SamplingParams(min_tokens=3).has_logit_adjust        # nothing to ban until the engine fills the list
    with pytest.raises(ValueError):
        SamplingParams(logit_bias={"x": 1})
    with pytest.raises(ValueError):
        SamplingParams(allowed_token_ids=[])
    with pytest.raises(ValueError):
        SamplingParams(truncate_prompt_tokens=0)
    with pytest.raises(ValueError):
        SamplingParams(logit_bias={10 ** 9: 1.0}).check_vocab(256)


def test_logit_bias_forces_and_forbids_tokens(llm):
    base = _gen(llm, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)).outputs[0].token_ids
    forced = _gen(llm, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logit_bias={42: 100.0}))
    assert forced.outputs[0].token_ids == [42] * 6
    first = base[0]
    avoided = _gen(llm, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logit_bias={first: -100.0}))
    assert avoided.outputs[0].token_ids[0] != first
    # a biased request next to an unbiased one in the same batch leaves the other untouched
    outs = llm.generate(["hello world", "hello world"],
                        [SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True),
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logit_bias={42: 100.0})])
    assert outs[0].outputs[0].token_ids == base and outs[1].outputs[0].token_ids == [42] * 6


def test_allowed_token_ids_restrict_greedy_and_sampled_outputs(llm):
    allowed = [3, 17, 99]
    g = _gen(llm, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True, allowed_token_ids=allowed))
    assert set(g.outputs[0].token_ids) <= set(allowed)
    s = _gen(llm, SamplingParams(temperature=1.5, seed=3, max_tokens=24, ignore_eos=True, allowed_token_ids=allowed))
    assert set(s.outputs[0].token_ids) <= set(allowed) and len(s.outputs[0].token_ids) == 24
    with pytest.raises(ValueError):
        _gen(llm, SamplingParams(max_tokens=2, allowed_token_ids=[10 ** 7]))


def test_bad_words_single_and_multi_token(llm):
    sp0 = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
    base = _gen(llm, sp0).outputs[0].token_ids
    # single token: the greedy first choice can never appear
    one = _gen(llm, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True, bad_words_token_ids=[[base[0]]]))
    assert base[0] not in one.outputs[0].token_ids
    # two tokens: the sequence (base[0], base[1]) is broken at its second token, the first stays
    two = _gen(llm, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True,
                                   bad_words_token_ids=[[base[0], base[1]]]))
    t = two.outputs[0].token_ids
    assert t[0] == base[0] and t[1] != base[1]
    assert all(not (a == base[0] and b == base[1]) for a, b in zip(t, t[1:]))
    # the string form is tokenised by the engine (byte tokenizer: one token per byte)
    word = bytes([base[0] % 256]).decode("latin-1")
    if word.isascii() and word.strip():
        s = _gen(llm, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True, bad_words=[word]))
        assert base[0] not in s.outputs[0].token_ids


def test_min_tokens_holds_back_eos_and_stop_tokens(llm):
    eos = llm.engine.eos_token_id
    assert eos is not None
    # EOS is made overwhelmingly likely: it may only appear once min_tokens outputs exist
    o = _gen(llm, SamplingParams(temperature=0.0, max_tokens=10, min_tokens=4, logit_bias={eos: 100.0}))
    ids = o.outputs[0].token_ids
    assert len(ids) == 5 and eos not in ids[:4] and ids[4] == eos and o.outputs[0].finish_reason == "stop"
    # same for a stop token id
    o = _gen(llm, SamplingParams(temperature=0.0, max_tokens=10, min_tokens=3, stop_token_ids=[77],
                                 ignore_eos=True, logit_bias={77: 100.0}))
    ids = o.outputs[0].token_ids
    assert len(ids) == 4 and 77 not in ids[:3] and ids[3] == 77


def test_truncate_prompt_tokens_keeps_the_tail(llm):
    long = "x" * 40 + "hello world"
    sp = SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True)
    want = llm.generate([{"prompt_token_ids": llm.engine.tokenizer.encode(long)[-11:]}], sp)[0]
    got = _gen(llm, SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True, truncate_prompt_tokens=11), long)
    assert len(got.prompt_token_ids) == 11
    assert got.outputs[0].token_ids == want.outputs[0].token_ids


def test_prompt_logprobs_match_a_full_forward_across_chunks_and_cache_hits():
    """prompt_logprobs: every prompt token's log-probability given its prefix.  Checked against
    one whole-prompt pass (a request that generates after the prompt and reports ``logprobs``
    for the same tokens), with the prefill split into 16-token chunks and with the prompt
    already sitting in the prefix cache (which must be bypassed, not trusted)."""
    import math
    pstate.destroy_model_parallel()
    m = LLM(**dict(KW, max_num_batched_tokens=16, enable_chunked_prefill=True, enable_prefix_caching=True))
    prompt = [(7 * i + 3) % 250 + 1 for i in range(45)]
    # reference: feed the first token, force the rest through logit_bias one token at a time, read their logprobs
    want = [None]
    for i in range(1, len(prompt)):
        o = m.generate([{"prompt_token_ids": prompt[:i]}],
                       SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True, logprobs=3))[0]
        lp = o.outputs[0].logprobs[0]
        want.append(lp)
    # the prompt is now fully cached: a prompt_logprobs request must still score every position
    o = m.generate([{"prompt_token_ids": prompt}],
                   SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True, prompt_logprobs=2))[0]
    got = o.prompt_logprobs
    assert got is not None and len(got) == len(prompt) and got[0] is None
    for i in range(1, len(prompt)):
        assert prompt[i] in got[i] and len(got[i]) in (2, 3)
        top_ref = max(want[i], key=want[i].get)
        assert top_ref in got[i] and math.isclose(got[i][top_ref], want[i][top_ref], abs_tol=1e-4)
    # two requests, one with and one without, in the same batch
    outs = m.generate([{"prompt_token_ids": prompt[:20]}, {"prompt_token_ids": prompt[:30]}],
                      [SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True),
                       SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True, prompt_logprobs=0)])
    assert outs[0].prompt_logprobs is None
    assert outs[1].prompt_logprobs[0] is None and len(outs[1].prompt_logprobs) == 30
    for t, d, ref in zip(prompt[1:30], outs[1].prompt_logprobs[1:], got[1:30]):
        assert math.isclose(d[t], ref[t], abs_tol=1e-4)
    m.shutdown()
    pstate.destroy_model_parallel()


def test_json_automaton_accepts_exactly_json_objects():
    import json
    from vllm_distributed_amd.ops.grammar import JsonGrammar, TokenTable
    good = ['{"a": 1, "b": [true, false, null, -0.5e+3, "x\\n\\u00e9"], "c": {"d": {}}}', ' {\n  "k" : [ ] }  ',
            '{"a":0}', '{"a":10.25E-2 }', '{"\\"q\\"": "\\\\"}', '{"n":[[[]],{"z":[1,2]}]}']
    bad = ['[1]', '"x"', '{"a": 01}', '{a:1}', '{"a":1,}', '{"a": tru }', '{"a":1}}', '{"a": "x\ny"}', '{"a":-}',
           '{"a":1.}', '{"a":.5}', '{"a":1 "b":2}', '{"a":[1,]}', '{"a":"\\x"}', '{"a":"\\u12g4"}', '{"a":1e}']
    for t in good:
        json.loads(t)
        g = JsonGrammar()
        assert g.advance(t.encode()) and g.done, t
        for i in range(len(t)):             # every prefix is a live state, and not "done" before the last brace
            p = JsonGrammar()
            assert p.advance(t[:i].encode()) and (not p.done or t[:i].strip() == t.strip()), (t, i)
    for t in bad:
        g = JsonGrammar()
        assert not (g.advance(t.encode()) and g.done), t
    # a rejected chunk leaves the state where it was
    g = JsonGrammar()
    assert g.advance(b'{"a":') and not g.advance(b'x') and g.advance(b'1}') and g.done
    # masks: multi-byte tokens are judged as a whole; empty (special) tokens never pass
    table = TokenTable([b'{', b'{"', b'"}', b'}', b'', b'a":', b' ', b'1', b'{}'], 12)
    assert table.n == 12
    g = JsonGrammar()
    assert g.mask(table).tolist() == [1, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0]
    g.advance(b'{"')
    # inside a key: any text, or close it - but '"}' closes the key and then lacks its colon
    assert g.mask(table).tolist() == [1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 0]
    g.advance(b'a":1')
    assert g.mask(table).tolist() == [0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0]


def test_json_object_mode_yields_parseable_json_from_random_weights():
    """A random-init model under the JSON constraint: whatever it samples parses, ends with EOS
    right after the closing brace, and an unconstrained neighbour in the batch is unaffected."""
    import json
    pstate.destroy_model_parallel()
    llm = LLM(**KW)
    base = _gen(llm, SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)).outputs[0].token_ids
    outs = llm.generate(
        ["hello world"] * 4,
        [SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)] +
        # a nudge towards closing quotes / braces keeps random weights from rambling inside one string
        [SamplingParams(temperature=1.0, seed=s, max_tokens=120, json_object=True,
                        logit_bias={ord('"'): 4.0, ord('}'): 4.0, ord(':'): 3.0}) for s in (1, 2, 3)])
    assert outs[0].outputs[0].token_ids == base
    finished = 0
    for o in outs[1:]:
        c = o.outputs[0]
        if c.finish_reason == "stop":
            finished += 1
            assert c.token_ids[-1] == llm.engine.eos_token_id
            assert isinstance(json.loads(c.text), dict)
        else:                                   # ran out of tokens mid-object: still a valid prefix
            from vllm_distributed_amd.ops.grammar import JsonGrammar
            assert JsonGrammar().advance(bytes(t for t in c.token_ids if t < 256))
    assert finished >= 1
    llm.shutdown()
    pstate.destroy_model_parallel()


def test_token_bytes_of_a_byte_level_bpe_tokenizer_round_trip(tmp_path):
    """The vocabulary byte table the grammar walks: for a byte-level BPE tokenizer (trained here,
    offline) the bytes of the encoded ids concatenate back to the UTF-8 text, specials are empty."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    from vllm_distributed_amd.tokenizer import HFTokenizer
    tk = Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    corpus = ['{"name": "value", "list": [1, 2, 3], "nested": {"k": true}}', "héllo wörld ünïcode 你好",
              "the quick brown fox jumps over the lazy dog " * 3]
    tk.train_from_iterator(corpus * 20, trainers.BpeTrainer(
        vocab_size=400, special_tokens=["<|eos|>"], initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|eos|>").save_pretrained(str(tmp_path))
    tok = HFTokenizer(str(tmp_path))
    table = tok.token_bytes()
    assert len(table) == tok.vocab_size and table[tok.eos_token_id] == b""
    for text in corpus + ['  {"a":\n\t"b"}', "é€"]:
        ids = tok.encode(text, add_special_tokens=False)
        assert b"".join(table[i] for i in ids) == text.encode("utf-8"), text
    assert any(len(t) > 1 for t in table)            # merges exist, so multi-byte tokens are covered


def test_token_bytes_of_a_sentencepiece_style_vocabulary():
    """SentencePiece spellings: U+2581 is a space, ``<0xNN>`` is one raw byte, specials are empty."""
    from vllm_distributed_amd.tokenizer import HFTokenizer

    class _Tok:
        vocab = ["<s>", "</s>", "<0x0A>", "<0xE4>", "▁the", "▁", "ing", "▁{\"", "é"]
        all_special_ids = [0, 1]
        added_tokens_decoder = {0: None, 1: None}

        def convert_ids_to_tokens(self, ids):
            return [self.vocab[i] for i in ids]

    tok = HFTokenizer.__new__(HFTokenizer)
    tok.tok, tok.vocab_size, tok.all_special_ids = _Tok(), len(_Tok.vocab), [0, 1]
    assert tok.token_bytes() == [b"", b"", b"\n", b"\xe4", b" the", b" ", b"ing", b' {"', "é".encode()]


def test_guided_choice_yields_exactly_one_of_the_strings():
    from vllm_distributed_amd.ops.grammar import ChoiceGrammar, TokenTable
    table = TokenTable([b"y", b"ye", b"yes", b"n", b"no", b"o", b"s", b"", b"x", b"yesterday"], 10)
    g = ChoiceGrammar(["yes", "no", "yesterday"])
    assert g.mask(table).tolist() == [1, 1, 1, 1, 1, 0, 0, 0, 0, 1]
    assert g.advance(b"ye") and not g.advance(b"x") and not g.done and not g.may_stop
    assert g.mask(table).tolist() == [0, 0, 0, 0, 0, 0, 1, 0, 0, 0]
    assert g.advance(b"s") and g.may_stop and not g.done          # "yesterday" is still possible
    assert g.advance(b"terday") and g.done
    pstate.destroy_model_parallel()
    llm = LLM(**KW)
    choices = ["alpha", "beta", "alphabet"]
    outs = llm.generate(["hello world"] * 6,
                        [SamplingParams(temperature=1.0, seed=s, max_tokens=20, guided_choice=choices) for s in range(6)])
    for o in outs:
        assert o.outputs[0].text in choices and o.outputs[0].finish_reason == "stop"
    with pytest.raises(ValueError):
        SamplingParams(guided_choice=[])
    llm.shutdown()
    pstate.destroy_model_parallel()


def test_regex_dfa_agrees_with_python_re_and_masks_tokens():
    """The byte-level DFA against Python's ``re`` on random strings; dead transitions are trimmed,
    so a token is allowed exactly when some completion still matches."""
    import random
    import re
    from vllm_distributed_amd.ops.grammar import RegexGrammar, TokenTable, compile_regex
    rng = random.Random(0)
    patterns = [r"(\d{3})-[a-c]+x?", r"(ab|a)(bc|c)?d*", r"[^a-y]{2,4}!", r"(?:foo|ba[rz]){1,3}", r"\w+@\w+\.(com|org)",
                r"-?(0|[1-9][0-9]*)(\.[0-9]+)?", r'"([^"\\]|\\.)*"', r"a{0}b", r"x{2,}", r"\s*\S+\s*"]
    alphabet = "abcdxyz019-!@. \"\\fo_r"
    for pat in patterns:
        rx = re.compile(pat.encode())
        for _ in range(300):
            s = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9))).encode()
            g = RegexGrammar(pat)
            ok = g.advance(s)
            assert (ok and g.may_stop) == (rx.fullmatch(s) is not None), (pat, s)
    g = RegexGrammar(r"(\d{3})-[a-c]+x?")
    assert g.advance(b"12") and not g.may_stop and g.advance(b"3-ab") and g.may_stop and not g.done
    assert not g.advance(b"d") and g.state >= 0 and g.advance(b"x") and g.done
    toks = [b"1", b"12", b"1234", b"-", b"a", b"", b"12-", b"123-a", b"9x"]
    t = TokenTable(toks, len(toks))
    assert RegexGrammar(r"(\d{3})-[a-c]+x?").mask(t).tolist() == [1, 1, 0, 0, 0, 0, 0, 1, 0]
    g = RegexGrammar(r"(\d{3})-[a-c]+x?")
    g.advance(b"12")
    assert g.mask(t).tolist() == [1, 0, 0, 0, 0, 0, 0, 0, 0]
    for bad in ["(", "a{5,2}", "[z-a]", "\\p", "a(?=b)", "a{200000}", "(a{1000}){1000}"]:
        with pytest.raises(ValueError):
            compile_regex(bad)
    assert compile_regex("é+").num_states >= 2       # UTF-8 literals are byte sequences


def test_json_schema_lowering_accepts_and_rejects():
    import json
    from vllm_distributed_amd.ops.grammar import RegexGrammar, json_schema_to_regex
    from vllm_distributed_amd.sampling_params import SamplingParams
    schema = {"type": "object", "properties": {
        "name": {"type": "string", "minLength": 1}, "age": {"type": "integer"},
        "tags": {"type": "array", "items": {"enum": ["a", "b"]}, "minItems": 1, "maxItems": 3},
        "addr": {"$ref": "#/$defs/addr"}, "when": {"type": "string", "format": "date"},
        "v": {"anyOf": [{"type": "number"}, {"type": "null"}]}},
        "required": ["name", "tags"],
        "$defs": {"addr": {"type": "object", "required": ["zip"],
                           "properties": {"zip": {"type": "string", "pattern": "^[0-9]{5}$"}}}}}
    rx = json_schema_to_regex(schema)
    good = ['{"name":"x","tags":["a"]}', '{"name": "y\\n", "age": -12, "tags": ["a", "b"], "addr": {"zip": "12345"}}',
            '{"name":"é","tags":["b","b","a"],"when":"2024-01-31","v":null}', '{"name":"q","tags":["a"],"v":-1.5e3}']
    bad = ['{"name":"x"}', '{"tags":["a"],"name":"x"}', '{"name":"x","tags":[]}', '{"name":"x","tags":["c"]}',
           '{"name":"","tags":["a"]}', '{"name":"x","tags":["a","a","a","a"]}',
           '{"name":"x","tags":["a"],"addr":{"zip":"1234"}}', '{"name":"x","tags":["a"],"extra":1}']
    for s in good:
        json.loads(s)
        g = RegexGrammar(rx)
        assert g.advance(s.encode()) and g.may_stop, s
    for s in bad:
        g = RegexGrammar(rx)
        assert not (g.advance(s.encode()) and g.may_stop), s
    g = RegexGrammar(json_schema_to_regex({}))                # no type: any JSON value
    assert g.advance(b'{"a": [1, null], "b": {"c": "d"}}') and g.may_stop        # up to two levels of nesting
    sp = SamplingParams(guided_json=schema)
    assert sp.guided_regex == rx and sp.guided_json is None and sp.has_logit_adjust
    for broken in ({"type": "object", "properties": {"a": {"$ref": "http://x"}}}, {"type": "frob"}, {"enum": []}):
        with pytest.raises(ValueError):
            SamplingParams(guided_json=broken)
    with pytest.raises(ValueError):
        SamplingParams(guided_regex="a", json_object=True)


ARITH = r"""
# sums of numbers and bracketed sums
root ::= expr
expr ::= term (("+" | "-") term)*
term ::= [0-9]+ | "(" expr ")"
"""
JSON_GBNF = r"""
root   ::= object
value  ::= object | array | string | number | "true" | "false" | "null"
object ::= "{" ws ( string ws ":" ws value ( ws "," ws string ws ":" ws value )* )? ws "}"
array  ::= "[" ws ( value ( ws "," ws value )* )? ws "]"
string ::= "\"" ( [^"\\\x00-\x1f] | "\\" ["\\/bfnrt] )* "\""
number ::= "-"? ( "0" | [1-9] [0-9]* ) ( "." [0-9]+ )?
ws     ::= [ \t\n]{0,4}
"""


def test_context_free_grammar_matcher_and_masks():
    """The stack-set matcher of csrc/core/cfg_grammar.cpp: nesting a regular expression cannot count,
    JSON against ``json.loads`` on random edits of valid documents, token masks through the vocabulary
    trie, and the grammars it must refuse."""
    import json
    import random
    from vllm_distributed_amd.ops.grammar import CfgGrammar, TokenTable, compile_grammar
    g = CfgGrammar(ARITH)
    assert not g.may_stop and not g.done
    assert g.advance(b"12") and g.may_stop and not g.done
    assert g.advance(b"+((3-4)") and not g.may_stop          # one bracket still open
    assert not g.advance(b"+)") and g.advance(b")") and g.may_stop
    assert not g.advance(b")") and not g.advance(b"x")       # rejected input leaves the position alone
    toks = [b"1", b"23", b"+", b"+(", b"(", b")", b"a", b"", b"1+", b"))", b"-9"]
    t = TokenTable(toks, len(toks))
    assert dict(zip(toks, g.mask(t).tolist())) == {b"1": 0, b"23": 0, b"+": 1, b"+(": 1, b"(": 0, b")": 0, b"a": 0,
                                                   b"": 0, b"1+": 0, b"))": 0, b"-9": 1}
    assert dict(zip(toks, CfgGrammar(ARITH).mask(t).tolist())) == {
        b"1": 1, b"23": 1, b"+": 0, b"+(": 0, b"(": 1, b")": 0, b"a": 0, b"": 0, b"1+": 1, b"))": 0, b"-9": 0}
    # JSON objects: the grammar and json.loads agree on documents and on single-character corruptions of them
    rng = random.Random(1)
    docs = ['{"a": [1, 2.5, {"b": "x\\n\u00e9"}], "c": null}'.replace("\u00e9", "é"), '{}', '{"k":{"k":{"k":[[[-0.5]]]}}}',
            '{"t": true, "f": false, "s": "a b"}']
    for d in docs:
        gj = CfgGrammar(JSON_GBNF)
        assert gj.advance(d.encode()) and gj.may_stop and gj.done, d
        for _ in range(60):
            i = rng.randrange(len(d))
            bad = d[:i] + rng.choice('{}[],:"x1 ') + d[i + 1:]
            try:
                ok = isinstance(json.loads(bad), dict)
            except ValueError:
                ok = False
            gj = CfgGrammar(JSON_GBNF)
            got = gj.advance(bad.encode()) and gj.may_stop
            if ok != got:      # the grammar is stricter than json.loads only about whitespace runs and number forms
                assert ok and not got and ("     " in bad or "e" in bad.lower() or "+" in bad), (bad, ok, got)
    # repetition counts and the any-byte dot
    gr = CfgGrammar('root ::= "a"{2,3} .{2} [^x]?')
    assert not gr.advance(b"a\xff\xfe") and gr.advance(b"aa") and not gr.may_stop and gr.advance(b"a")
    assert gr.advance(b"\xff\x00") and gr.may_stop and not gr.advance(b"x") and gr.advance(b"y") and gr.done
    for bad in ['root ::= root "a" | "b"', 'root ::= "a', 'rot ::= "a"', 'root ::= x', 'root ::= [z-a]',
                'root ::= ("a"', 'root ::= "a"{3,2}', 'root ::= a\na ::= b "x" | "y"\nb ::= a?  "z"',
                'root ::= "a"\nroot ::= "b"']:
        with pytest.raises(ValueError):
            compile_grammar(bad)
    from vllm_distributed_amd.sampling_params import SamplingParams
    assert SamplingParams(guided_grammar=ARITH).has_logit_adjust
    with pytest.raises(ValueError):
        SamplingParams(guided_grammar=ARITH, guided_regex="a")
    with pytest.raises(ValueError):
        SamplingParams(guided_grammar="root ::= (")


def test_guided_grammar_holds_generation_to_the_language():
    """Random-weight model, temperature 1: whatever it prefers, the text is a sentence of the grammar
    (balanced brackets) or a prefix of one when max_tokens cut it short."""
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.ops.grammar import CfgGrammar
    llm = LLM(**KW)
    sps = [SamplingParams(temperature=1.0, seed=s, max_tokens=24, guided_grammar=ARITH,
                          logit_bias={str(ord("(")): 3.0, str(ord(")")): 2.0}) for s in range(6)]
    outs = llm.generate(["x"] * len(sps), sps)
    finished = 0
    for o in outs:
        text = o.outputs[0].text.encode()
        g = CfgGrammar(ARITH)
        assert text and g.advance(text), text
        if o.outputs[0].finish_reason == "stop":
            finished += 1
            assert g.may_stop and text.count(b"(") == text.count(b")"), text
    assert finished >= 1
    llm.shutdown()


def test_constraint_compilers_refuse_hostile_inputs_instead_of_crashing():
    """A request-supplied pattern or grammar reaches native recursive-descent parsers inside the serving
    process: deep nesting, empty-operand repeats and oversize texts must come back as ValueError, quickly."""
    import time

    from vllm_distributed_amd.ops.grammar import compile_grammar, compile_regex
    from vllm_distributed_amd.sampling_params import MAX_CONSTRAINT_BYTES
    t0 = time.time()
    for bad in ("(" * 300000, "(" * 300 + "a" + ")" * 300, "(){100000}" * 50, "a" * (MAX_CONSTRAINT_BYTES + 1),
                "a****************", "(?:(?:a{1000}){1000}){1000}"):
        with pytest.raises(ValueError):
            compile_regex(bad)
        with pytest.raises(ValueError):
            SamplingParams(guided_regex=bad)
    for bad in ("root ::= " + "(" * 300000, "root ::= " + "(" * 300 + '"a"' + ")" * 300,
                "root ::= " + '"a" ' * (MAX_CONSTRAINT_BYTES // 4 + 1), 'root ::= ("a"{256}){256}' * 40):
        with pytest.raises(ValueError):
            compile_grammar(bad)
        with pytest.raises(ValueError):
            SamplingParams(guided_grammar=bad)
    assert time.time() - t0 < 30
    compile_regex("(" * 200 + "a" + ")" * 200)                         # deep, but within the limit
    compile_grammar("root ::= " + "(" * 200 + '"a"' + ")" * 200)


def test_pythonic_tool_parser_keeps_hostile_output_as_text():
    from vllm_distributed_amd.entrypoints.openai.tool_parsers import PythonicToolParser
    p = PythonicToolParser(None)
    for text in ("[" + "f(a=" * 5000 + "1" + ")" * 5000 + "]", '[f(**{"a": 1})]', "[f(a=" + "[" * 100000 + "]" * 100000 + ")]"):
        content, calls = p.extract(text)
        assert calls == [] and content == text
    content, calls = p.extract('[get_weather(city="Paris", days=3)]')
    assert content is None and len(calls) == 1
