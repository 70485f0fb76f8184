"""Control-plane RPC: frame codec, peer semantics, handshake (CPU only)."""
import asyncio
import gc

import pytest
from hypothesis import given, settings, strategies as st

from vllm_distributed_amd.rpc import (
    AuthenticationError, FrameDecoder, FrameError, MemoryTransport, RPCConnectionError,
    RPCResultError, RPCTimeoutError, RpcProxy, StreamTransport, client_handshake,
    encode_frame, prepare_peer_readloop, server_handshake, JsonCodec)


def run(coro):
    return asyncio.run(coro)


# ------------------------------------------------------------------ frames
@given(st.lists(st.tuples(st.sampled_from([0, 1, 2]), st.binary(max_size=300)), max_size=8),
       st.integers(min_value=1, max_value=17))
@settings(max_examples=50, deadline=None)
def test_frame_roundtrip_any_chunking(frames, chunk):
    wire = b"".join(encode_frame(t, p) for t, p in frames)
    dec = FrameDecoder()
    out = []
    for i in range(0, len(wire), chunk):
        out += dec.feed(wire[i:i + chunk])
    assert out == frames
    assert dec.pending_bytes == 0


def test_frame_header_layout():
    f = encode_frame(1, b"abc")
    assert f[:4] == (4).to_bytes(4, "big") and f[4] == 1 and f[5:] == b"abc"


def test_frame_truncated_and_oversized():
    dec = FrameDecoder(max_frame_bytes=16)
    assert dec.feed(encode_frame(0, b"hello")[:-1]) == []
    with pytest.raises(FrameError):
        FrameDecoder(max_frame_bytes=16).feed((100).to_bytes(4, "big") + b"\x00")
    with pytest.raises(FrameError):
        FrameDecoder().feed((0).to_bytes(4, "big") + b"\x00")
    with pytest.raises(FrameError):
        FrameDecoder().feed((2).to_bytes(4, "big") + b"\x09x")


# ------------------------------------------------------------------- peers
async def make_pair(codec=None):
    loop = asyncio.get_running_loop()
    ta, tb = MemoryTransport.pair(*( [codec] if codec else [] ))
    pa, rla = await prepare_peer_readloop(loop, ta, "a", "b")
    pb, rlb = await prepare_peer_readloop(loop, tb, "b", "a")
    tasks = [loop.create_task(rla()), loop.create_task(rlb())]
    return pa, pb, ta, tb, tasks


async def shutdown(ta, tb, tasks):
    ta.close()
    await asyncio.gather(*tasks, return_exceptions=True)


def test_param_by_value_and_proxy_call():
    async def main():
        pa, pb, ta, tb, tasks = await make_pair()
        pb.params["n"] = 7
        pb.params["cfg"] = {"a": [1, 2.5, "x", None, True], "__rpc": "user-key"}
        pb.params["add"] = lambda x, y: x + y

        async def aadd(x, y):
            await asyncio.sleep(0)
            return x * y
        pb.params["mul"] = aadd
        assert await pa.get_param("n") == 7
        assert await pa.getParam("cfg") == {"a": [1, 2.5, "x", None, True], "__rpc": "user-key"}
        add = await pa.get_param("add")
        assert isinstance(add, RpcProxy)
        assert await add(2, 3) == 5
        mul = await pa.get_param("mul")
        assert await mul(4, 5) == 20
        with pytest.raises(RPCResultError) as ei:
            await pa.get_param("missing")
        assert ei.value.name == "KeyError"
        await shutdown(ta, tb, tasks)
    run(main())


def test_multiple_sideband_buffers_keep_order():
    async def main():
        pa, pb, ta, tb, tasks = await make_pair()
        pb.params["cat"] = lambda a, b, c: (a + b"|" + b + b"|" + c, len(a))
        cat = await pa.get_param("cat")
        out = await cat(b"first", b"second-longer", bytearray(b"3"))
        assert out == [b"first|second-longer|3", 5]
        await shutdown(ta, tb, tasks)
    run(main())


def test_json_codec_with_buffers():
    async def main():
        pa, pb, ta, tb, tasks = await make_pair(JsonCodec)
        pb.params["echo"] = lambda b: b[::-1]
        echo = await pa.get_param("echo")
        assert await echo(b"\x00\x01\xff") == b"\xff\x01\x00"
        await shutdown(ta, tb, tasks)
    run(main())


def test_remote_exception_and_object_methods():
    class Counter:
        def __init__(self):
            self.n = 0
            self.notes = []

        def inc(self, k=1):
            self.n += k
            return self.n

        def boom(self):
            raise ValueError("bad things")

        def note(self, s):
            self.notes.append(s)

    setattr(Counter, "__proxy_oneway_methods", ["note"])
    setattr(Counter, "__proxy_props", {"kind": "counter"})

    async def main():
        pa, pb, ta, tb, tasks = await make_pair()
        c = Counter()
        pb.params["c"] = c
        rc = await pa.get_param("c")
        assert rc.kind == "counter"
        assert await rc.inc() == 1 and await rc.inc(5) == 6
        with pytest.raises(RPCResultError) as ei:
            await rc.boom()
        assert ei.value.name == "ValueError" and "bad things" in ei.value.message
        assert "boom" in (ei.value.remote_stack or "")
        assert await rc.note("hi") is None  # one-way: no reply awaited
        assert await rc.inc() == 7          # ordering: note was processed before
        assert c.notes == ["hi"]
        with pytest.raises(RPCResultError):
            await rc._private()
        await shutdown(ta, tb, tasks)
    run(main())


def test_callback_roundtrip_of_own_proxy():
    async def main():
        pa, pb, ta, tb, tasks = await make_pair()
        seen = []

        def register(cb):
            seen.append(cb)
            return cb           # hand the caller's own function back

        pb.params["register"] = register

        def my_cb(x):
            return x + 1
        reg = await pa.get_param("register")
        back = await reg(my_cb)
        assert back is my_cb                 # resolved to the local object, not a proxy
        assert isinstance(seen[0], RpcProxy)
        assert await seen[0](41) == 42       # b can call back into a
        await shutdown(ta, tb, tasks)
    run(main())


def test_async_iterator_bridging():
    async def main():
        pa, pb, ta, tb, tasks = await make_pair()

        async def gen(n):
            for i in range(n):
                yield i * i
        pb.params["gen"] = gen
        g = await pa.get_param("gen")
        it = await g(4)
        assert [x async for x in it] == [0, 1, 4, 9]
        await shutdown(ta, tb, tasks)
    run(main())


def test_finalize_releases_remote_object():
    async def main():
        pa, pb, ta, tb, tasks = await make_pair()

        class Obj:
            def ping(self):
                return "pong"
        pb.params["make"] = lambda: Obj()
        make = await pa.get_param("make")
        o = await make()
        assert await o.ping() == "pong"
        n_before = pb.stats()["local_proxies"]
        del o
        gc.collect()
        for _ in range(10):
            await asyncio.sleep(0.01)
        assert pb.stats()["local_proxies"] == n_before - 1
        await shutdown(ta, tb, tasks)
    run(main())


def test_kill_rejects_pending_and_timeout_enforced():
    async def main():
        pa, pb, ta, tb, tasks = await make_pair()

        async def slow():
            await asyncio.sleep(30)
        pb.params["slow"] = slow
        slow_p = await pa.get_param("slow")
        with pytest.raises(RPCTimeoutError):
            await slow_p(timeout=0.05)
        pending = asyncio.ensure_future(slow_p())
        await asyncio.sleep(0.05)
        tb.close()  # connection drops
        with pytest.raises(RPCConnectionError):
            await pending
        with pytest.raises(RPCConnectionError):
            await pa.get_param("anything")
        await asyncio.gather(*tasks, return_exceptions=True)
    run(main())


# --------------------------------------------------------------- handshake
def test_handshake_over_tcp_accepts_and_rejects():
    async def main():
        results = []

        async def on_client(reader, writer):
            t = StreamTransport(reader, writer)
            try:
                await server_handshake(t, "s3cret", timeout=2)
                results.append("ok")
            except AuthenticationError:
                results.append("rejected")
            finally:
                writer.close()
        server = await asyncio.start_server(on_client, "127.0.0.1", 0)
        port = server.sockets[0].getsockname()[1]
        r, w = await asyncio.open_connection("127.0.0.1", port)
        await client_handshake(StreamTransport(r, w), "s3cret", timeout=2)
        w.close()
        r, w = await asyncio.open_connection("127.0.0.1", port)
        with pytest.raises(AuthenticationError):
            await client_handshake(StreamTransport(r, w), "wrong", timeout=2)
        w.close()
        await asyncio.sleep(0.05)
        server.close()
        await server.wait_closed()
        assert results == ["ok", "rejected"]
    run(main())


def test_rpc_suite_is_clean_under_asyncio_debug_mode():
    """The control-plane tests again with ``PYTHONASYNCIODEBUG=1`` (slow-callback
    and never-awaited-coroutine detection) and warnings as errors for coroutine
    misuse: cross-thread hand-offs must all go through the loop."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONASYNCIODEBUG="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-W", "error::RuntimeWarning", "-m", "pytest", "-q", "-x",
                        os.path.join(root, "tests", "test_rpc.py"), "-k", "not asyncio_debug_mode",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "never awaited" not in r.stdout + r.stderr
