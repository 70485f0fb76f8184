"""Symmetric object-proxy RPC peer.

Two peers joined by any transport can (a) read each other's named *params*,
(b) call functions/objects the other side handed out, and (c) pass values
either *by value* (plain data, bytes, registered serialisable types) or *by
reference* (everything else becomes a proxy the receiver can call back into).
This is the control-plane contract of the reference's ``RpcPeer``
(reference ``src/rpc.py:157-619``: message types ``param``/``apply``/``result``/
``finalize``, by-value vs proxy rules ``:284-391``, error envelopes ``:249-263``,
distributed GC ``:393-426,569-583``, async-iterator bridging ``:77-120``,
``kill`` ``:220-235``).

Design differences (deliberate):

* envelopes are tagged ``{"__rpc": kind, ...}`` and contain only plain data, so
  the wire codec can be msgpack/JSON (non-executable); pickled payloads travel
  only as opaque sideband byte buffers that the application unpickles itself;
* sideband buffers are consumed FIFO by index, so any number of ``bytes``
  arguments per message is correct (the reference pops LIFO, ``src/rpc_reader.py:26-38``);
* proxies are reference counted per (object, peer): an object sent twice shares
  one id and is dropped when every remote handle has been finalised;
* every call can carry a timeout that is enforced on the caller.
"""
from __future__ import annotations

import asyncio
import inspect
import itertools
import os
import traceback
import weakref
from typing import Any, Awaitable, Callable, Dict, List, Optional, Tuple

TAG = "__rpc"
_PRIMITIVES = (type(None), bool, int, float, str)


class RPCResultError(Exception):
    """An exception raised on the other side of the connection."""

    def __init__(self, name: str, message: str, stack: Optional[str] = None):
        super().__init__(f"{name}: {message}" if name else message)
        self.name = name
        self.message = message
        self.remote_stack = stack


class RPCTimeoutError(TimeoutError):
    pass


class RPCConnectionError(ConnectionError):
    """The peer was killed (transport closed) while a call was pending."""


class RpcSerializer:
    """By-value codec for one type: ``serialize`` must return plain data and/or
    ``bytes`` (which ride as sideband buffers)."""

    def serialize(self, value: Any, ctx: dict) -> Any:  # pragma: no cover - interface
        raise NotImplementedError

    def deserialize(self, value: Any, ctx: dict) -> Any:  # pragma: no cover - interface
        raise NotImplementedError


async def maybe_await(value: Any) -> Any:
    if inspect.isawaitable(value):
        return await value
    return value


class RpcProxyMethod:
    __slots__ = ("_proxy", "_name")

    def __init__(self, proxy: "RpcProxy", name: str):
        self._proxy = proxy
        self._name = name

    def __call__(self, *args: Any, timeout: Optional[float] = None) -> Awaitable[Any]:
        p = self._proxy
        oneway = self._name in p._rpc_oneway
        return p._rpc_peer.apply(p._rpc_id, self._name, list(args), oneway=oneway, timeout=timeout)


class RpcProxy:
    """Local handle for an object that lives on the other peer."""

    def __init__(self, peer: "RpcPeer", proxy_id: str, ctor: str, props: Optional[dict],
                 oneway: Optional[List[str]], is_aiter: bool):
        object.__setattr__(self, "_rpc_peer", peer)
        object.__setattr__(self, "_rpc_id", proxy_id)
        object.__setattr__(self, "_rpc_ctor", ctor)
        object.__setattr__(self, "_rpc_props", props or {})
        object.__setattr__(self, "_rpc_oneway", set(oneway or ()))
        object.__setattr__(self, "_rpc_is_aiter", is_aiter)

    def __getattr__(self, name: str) -> Any:
        if name.startswith("_rpc_") or name.startswith("__"):
            raise AttributeError(name)
        props = object.__getattribute__(self, "_rpc_props")
        if name in props:
            return props[name]
        return RpcProxyMethod(self, name)

    def __call__(self, *args: Any, timeout: Optional[float] = None) -> Awaitable[Any]:
        return self._rpc_peer.apply(self._rpc_id, None, list(args), timeout=timeout)

    # async-iterator bridging: ``async for x in proxy`` pulls items remotely
    def __aiter__(self):
        return self

    async def __anext__(self):
        try:
            return await self._rpc_peer.apply(self._rpc_id, "__anext__", [])
        except RPCResultError as e:
            if e.name == "StopAsyncIteration":
                raise StopAsyncIteration from None
            raise

    async def aclose(self):
        try:
            await self._rpc_peer.apply(self._rpc_id, "aclose", [])
        except (RPCResultError, RPCConnectionError):
            pass

    def __repr__(self) -> str:
        return f"<RpcProxy {self._rpc_ctor} id={self._rpc_id} via {self._rpc_peer.peer_name}>"


SendFn = Callable[[dict, Optional[List[bytes]]], Any]


class RpcPeer:
    """One end of an RPC connection.

    ``send(message, buffers)`` is injected, so the peer itself never touches a
    socket and two peers can be wired back-to-back in tests.
    """

    # attribute names an object may define to control its proxying
    PROP_PROXY_PROPS = "__proxy_props"
    PROP_PROXY_ONEWAY = "__proxy_oneway_methods"

    def __init__(self, self_name: str, peer_name: str, send: SendFn):
        self.self_name = self_name
        self.peer_name = peer_name
        self._send = send
        self.params: Dict[str, Any] = {}
        self.serializers: Dict[str, RpcSerializer] = {}
        self._serializer_types: List[Tuple[type, str]] = []
        self._id_prefix = os.urandom(3).hex()
        self._counter = itertools.count(1)
        # objects we handed out: id -> [obj, refcount]
        self._local: Dict[str, list] = {}
        self._local_ids: Dict[int, str] = {}
        # live proxies for remote objects: id -> weakref
        self._remote: Dict[str, "weakref.ReferenceType[RpcProxy]"] = {}
        self._pending: Dict[str, asyncio.Future] = {}
        self.killed: Optional[BaseException] = None
        self._kill_listeners: List[Callable[[BaseException], None]] = []
        self.loop: Optional[asyncio.AbstractEventLoop] = None
        self.tags: Dict[str, Any] = {}

    # ------------------------------------------------------------------ setup
    def add_serializer(self, typ: type, name: str, serializer: RpcSerializer) -> None:
        self.serializers[name] = serializer
        self._serializer_types.append((typ, name))

    def on_killed(self, cb: Callable[[BaseException], None]) -> None:
        if self.killed is not None:
            cb(self.killed)
        else:
            self._kill_listeners.append(cb)

    def _new_id(self) -> str:
        return f"{self._id_prefix}{next(self._counter):x}"

    # --------------------------------------------------------------- lifecycle
    def kill(self, reason: Any = None) -> None:
        """Reject every pending call and refuse new ones.  Call from the loop
        thread that owns the peer."""
        if self.killed is not None:
            return
        if isinstance(reason, BaseException):
            err: BaseException = RPCConnectionError(f"peer {self.peer_name} killed: {reason!r}")
            err.__cause__ = reason
        else:
            err = RPCConnectionError(f"peer {self.peer_name} killed: {reason or 'connection closed'}")
        self.killed = err
        pending, self._pending = self._pending, {}
        for fut in pending.values():
            if not fut.done():
                fut.set_exception(err)
        self._local.clear()
        self._local_ids.clear()
        self.params.clear()
        for cb in self._kill_listeners:
            try:
                cb(err)
            except Exception:  # pragma: no cover
                traceback.print_exc()
        self._kill_listeners.clear()

    # -------------------------------------------------------------- serialise
    def _error_envelope(self, e: BaseException) -> dict:
        return {TAG: "err", "name": type(e).__name__, "message": str(e),
                "stack": "".join(traceback.format_exception(type(e), e, e.__traceback__))}

    def serialize(self, value: Any, ctx: dict) -> Any:
        if isinstance(value, _PRIMITIVES):
            return value
        if isinstance(value, (list, tuple)):
            return [self.serialize(v, ctx) for v in value]
        if isinstance(value, dict):
            out = {}
            plain = True
            for k, v in value.items():
                if not isinstance(k, str):
                    plain = False
                    break
                out[k] = self.serialize(v, ctx)
            if plain:
                return {TAG: "dict", "v": out} if TAG in out else out
            return {TAG: "map", "v": [[self.serialize(k, ctx), self.serialize(v, ctx)]
                                      for k, v in value.items()]}
        if isinstance(value, (bytes, bytearray, memoryview)):
            bufs = ctx.setdefault("buffers", [])
            bufs.append(bytes(value) if not isinstance(value, bytes) else value)
            return {TAG: "buf", "i": len(bufs) - 1}
        if isinstance(value, BaseException):
            return self._error_envelope(value)
        if isinstance(value, RpcProxy):
            if value._rpc_peer is self:
                # the other side owns it: tell it to use its own object
                return {TAG: "local", "id": value._rpc_id}
        for typ, name in self._serializer_types:
            if isinstance(value, typ):
                return {TAG: "ser", "name": name,
                        "v": self.serialize(self.serializers[name].serialize(value, ctx), ctx)}
        return self._proxy_envelope(value)

    def _proxy_envelope(self, value: Any) -> dict:
        key = id(value)
        pid = self._local_ids.get(key)
        if pid is not None and pid in self._local and self._local[pid][0] is value:
            self._local[pid][1] += 1
        else:
            pid = self._new_id()
            self._local[pid] = [value, 1]
            self._local_ids[key] = pid
        props = getattr(value, self.PROP_PROXY_PROPS, None)
        oneway = getattr(value, self.PROP_PROXY_ONEWAY, None)
        env = {TAG: "proxy", "id": pid, "ctor": type(value).__name__}
        if props:
            env["props"] = self.serialize(dict(props), {})
        if oneway:
            env["oneway"] = list(oneway)
        if hasattr(value, "__anext__"):
            env["aiter"] = True
        return env

    def deserialize(self, value: Any, ctx: dict) -> Any:
        if isinstance(value, _PRIMITIVES):
            return value
        if isinstance(value, list):
            return [self.deserialize(v, ctx) for v in value]
        if isinstance(value, (bytes, bytearray)):
            return bytes(value)
        if not isinstance(value, dict):
            return value
        kind = value.get(TAG)
        if kind is None:
            return {k: self.deserialize(v, ctx) for k, v in value.items()}
        if kind == "dict":
            return {k: self.deserialize(v, ctx) for k, v in value["v"].items()}
        if kind == "map":
            return {_hashable(self.deserialize(k, ctx)): self.deserialize(v, ctx)
                    for k, v in value["v"]}
        if kind == "buf":
            bufs = ctx.get("buffers") or []
            i = value["i"]
            if not 0 <= i < len(bufs):
                raise RPCResultError("ProtocolError", f"missing sideband buffer {i}")
            return bufs[i]
        if kind == "err":
            return RPCResultError(value.get("name", ""), value.get("message", ""), value.get("stack"))
        if kind == "local":
            entry = self._local.get(value["id"])
            if entry is None:
                raise RPCResultError("ProtocolError", f"unknown local proxy {value['id']}")
            return entry[0]
        if kind == "ser":
            ser = self.serializers.get(value["name"])
            if ser is None:
                raise RPCResultError("ProtocolError", f"no serializer {value['name']!r}")
            return ser.deserialize(self.deserialize(value["v"], ctx), ctx)
        if kind == "proxy":
            return self._make_proxy(value)
        raise RPCResultError("ProtocolError", f"unknown envelope {kind!r}")

    def _make_proxy(self, env: dict) -> RpcProxy:
        pid = env["id"]
        props = self.deserialize(env.get("props"), {}) if env.get("props") else None
        proxy = RpcProxy(self, pid, env.get("ctor", "object"), props, env.get("oneway"),
                         bool(env.get("aiter")))
        # every envelope bumped the owner's refcount once; every proxy object
        # releases exactly one count when collected
        weakref.finalize(proxy, RpcPeer._finalize_remote, weakref.ref(self), pid)
        self._remote[pid] = weakref.ref(proxy)
        return proxy

    @staticmethod
    def _finalize_remote(peer_ref: "weakref.ReferenceType[RpcPeer]", pid: str) -> None:
        peer = peer_ref()
        if peer is None or peer.killed is not None:
            return
        loop = peer.loop
        msg = {"type": "finalize", "id": pid}

        def _do():
            if peer.killed is None:
                try:
                    res = peer._dispatch_send(msg, None)
                    if inspect.isawaitable(res):
                        asyncio.ensure_future(res)
                except Exception:
                    pass
        try:
            if loop is not None and loop.is_running():
                loop.call_soon_threadsafe(_do)
        except RuntimeError:
            pass

    # ------------------------------------------------------------------- send
    def _dispatch_send(self, msg: dict, buffers: Optional[List[bytes]]) -> Any:
        return self._send(msg, buffers)

    async def _send_async(self, msg: dict, buffers: Optional[List[bytes]]) -> None:
        await maybe_await(self._dispatch_send(msg, buffers))

    # ------------------------------------------------------------------ calls
    async def get_param(self, name: str, timeout: Optional[float] = None) -> Any:
        return await self._request({"type": "param", "param": name}, None, timeout)

    getParam = get_param  # reference spelling (src/rpc.py:610)

    async def apply(self, proxy_id: str, method: Optional[str], args: list,
                    oneway: bool = False, timeout: Optional[float] = None) -> Any:
        ctx: dict = {}
        msg = {"type": "apply", "proxy": proxy_id, "method": method,
               "args": self.serialize(args, ctx)}
        if oneway:
            if self.killed is not None:
                raise self.killed
            msg["oneway"] = True
            await self._send_async(msg, ctx.get("buffers"))
            return None
        return await self._request(msg, ctx.get("buffers"), timeout)

    async def _request(self, msg: dict, buffers: Optional[List[bytes]],
                       timeout: Optional[float]) -> Any:
        if self.killed is not None:
            raise self.killed
        loop = asyncio.get_running_loop()
        if self.loop is None:
            self.loop = loop
        call_id = self._new_id()
        fut: asyncio.Future = loop.create_future()
        self._pending[call_id] = fut
        msg["id"] = call_id
        try:
            await self._send_async(msg, buffers)
            if timeout is not None and timeout > 0:
                try:
                    return await asyncio.wait_for(fut, timeout)
                except asyncio.TimeoutError:
                    raise RPCTimeoutError(
                        f"rpc {msg.get('method') or msg.get('param') or 'call'} to "
                        f"{self.peer_name} timed out after {timeout}s") from None
            return await fut
        finally:
            self._pending.pop(call_id, None)

    # --------------------------------------------------------------- incoming
    async def handle_message(self, msg: dict, buffers: Optional[List[bytes]] = None) -> None:
        if self.killed is not None:
            return
        if self.loop is None:
            self.loop = asyncio.get_running_loop()
        ctx = {"buffers": buffers or []}
        mtype = msg.get("type")
        if mtype == "result":
            fut = self._pending.get(msg.get("id"))
            if fut is None or fut.done():
                return
            try:
                if "error" in msg:
                    err = self.deserialize(msg["error"], ctx)
                    fut.set_exception(err if isinstance(err, BaseException)
                                      else RPCResultError("Error", str(err)))
                else:
                    fut.set_result(self.deserialize(msg.get("value"), ctx))
            except Exception as e:  # malformed result
                if not fut.done():
                    fut.set_exception(e)
            return
        if mtype == "finalize":
            entry = self._local.get(msg.get("id"))
            if entry is not None:
                entry[1] -= 1
                if entry[1] <= 0:
                    self._local.pop(msg["id"], None)
                    self._local_ids.pop(id(entry[0]), None)
            return
        if mtype == "param":
            await self._reply(msg, lambda: self._lookup_param(msg.get("param")))
            return
        if mtype == "apply":
            await self._reply(msg, lambda: self._invoke(msg, ctx))
            return
        raise RPCResultError("ProtocolError", f"unknown message type {mtype!r}")

    def _lookup_param(self, name: Any) -> Any:
        if name not in self.params:
            raise KeyError(f"param {name!r} not found on {self.self_name}")
        return self.params[name]

    def _invoke(self, msg: dict, ctx: dict) -> Any:
        entry = self._local.get(msg.get("proxy"))
        if entry is None:
            raise RPCResultError("ProtocolError",
                                 f"proxy {msg.get('proxy')!r} not found on {self.self_name}")
        target = entry[0]
        args = self.deserialize(msg.get("args") or [], ctx)
        method = msg.get("method")
        if method is None:
            return target(*args)
        if method.startswith("_") and method not in ("__anext__", "__aiter__"):
            raise AttributeError(f"refusing to call private method {method!r}")
        return getattr(target, method)(*args)

    async def _reply(self, msg: dict, fn: Callable[[], Any]) -> None:
        oneway = bool(msg.get("oneway")) or msg.get("id") is None
        reply: dict = {"type": "result", "id": msg.get("id")}
        ctx: dict = {}
        try:
            value = await maybe_await(fn())
            reply["value"] = self.serialize(value, ctx)
        except BaseException as e:  # noqa: BLE001 - must be reported to the caller
            if isinstance(e, (KeyboardInterrupt, SystemExit, asyncio.CancelledError)):
                raise
            ctx = {}
            reply["error"] = self._error_envelope(e)
        if oneway or self.killed is not None:
            return
        try:
            await self._send_async(reply, ctx.get("buffers"))
        except Exception as e:
            self.kill(e)

    # ------------------------------------------------------------ diagnostics
    def stats(self) -> dict:
        return {"local_proxies": len(self._local), "pending": len(self._pending),
                "remote_proxies": sum(1 for r in self._remote.values() if r() is not None)}


def _hashable(k: Any) -> Any:
    return tuple(k) if isinstance(k, list) else k
