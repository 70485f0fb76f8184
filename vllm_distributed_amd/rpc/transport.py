"""RPC transports, frame codec, authentication handshake and read loop.

Frame format (identical on every byte-stream transport; reference
``src/rpc_reader.py:73-82,116-135,155-174``)::

    uint32 big-endian  length = len(payload) + 1
    uint8              type: 0 = serialised message, 1 = raw sideband buffer,
                             2 = handshake
    bytes              payload

Transports (reference ``src/rpc_reader.py:41-206``):

* ``StreamTransport``      asyncio streams (TCP, unix, socketpair) - remote workers
* ``ConnectionTransport``  ``multiprocessing.Connection`` - workers spawned by the head
* ``FileTransport``        a pair of OS file descriptors (pipes / stdio)
* ``MemoryTransport``      in-process queue pair for tests

Messages are encoded with msgpack (or JSON); neither can execute code on
decode.  Application payloads that *are* pickles ride as opaque sideband
buffers and are only unpickled by the application after the connection has
passed the HMAC challenge/response in ``server_handshake``/``client_handshake``.
Writes await ``drain()`` so a slow reader exerts back-pressure (the reference
never drains, ``src/rpc_reader.py:166-174``).
"""
from __future__ import annotations

import asyncio
import concurrent.futures
import hashlib
import hmac
import json
import os
import struct
from typing import Any, Callable, List, Optional, Tuple

import msgpack

from .peer import RpcPeer, RPCConnectionError

FRAME_MESSAGE = 0
FRAME_BUFFER = 1
FRAME_HANDSHAKE = 2
_HEADER = struct.Struct(">IB")
MAX_FRAME_BYTES = 1 << 31  # 2 GiB: refuse absurd lengths from a confused peer


class FrameError(ValueError):
    pass


def encode_frame(ftype: int, payload: bytes) -> bytes:
    if len(payload) + 1 > MAX_FRAME_BYTES:
        raise FrameError(f"frame too large: {len(payload)} bytes")
    return _HEADER.pack(len(payload) + 1, ftype) + payload


class FrameDecoder:
    """Incremental decoder: ``feed`` arbitrary byte chunks, get whole frames."""

    def __init__(self, max_frame_bytes: int = MAX_FRAME_BYTES):
        self._buf = bytearray()
        self._max = max_frame_bytes

    def feed(self, data: bytes) -> List[Tuple[int, bytes]]:
        self._buf += data
        out: List[Tuple[int, bytes]] = []
        while True:
            if len(self._buf) < _HEADER.size:
                return out
            length, ftype = _HEADER.unpack_from(self._buf, 0)
            if length < 1 or length > self._max:
                raise FrameError(f"bad frame length {length}")
            if ftype not in (FRAME_MESSAGE, FRAME_BUFFER, FRAME_HANDSHAKE):
                raise FrameError(f"bad frame type {ftype}")
            end = 4 + length
            if len(self._buf) < end:
                return out
            out.append((ftype, bytes(self._buf[_HEADER.size:end])))
            del self._buf[:end]

    @property
    def pending_bytes(self) -> int:
        return len(self._buf)


# ---------------------------------------------------------------- codecs
class MsgpackCodec:
    name = "msgpack"

    @staticmethod
    def dumps(obj: Any) -> bytes:
        return msgpack.packb(obj, use_bin_type=True)

    @staticmethod
    def loads(data: bytes) -> Any:
        return msgpack.unpackb(data, raw=False, strict_map_key=False)


class JsonCodec:
    name = "json"

    @staticmethod
    def dumps(obj: Any) -> bytes:
        return json.dumps(obj, separators=(",", ":")).encode()

    @staticmethod
    def loads(data: bytes) -> Any:
        return json.loads(data)


# ------------------------------------------------------------- transports
class RpcTransport:
    """Moves frames.  ``read`` returns ``(frame_type, payload)``."""

    codec = MsgpackCodec

    async def prepare(self) -> None:
        return None

    async def read(self) -> Tuple[int, bytes]:
        raise NotImplementedError

    async def write(self, ftype: int, payload: bytes) -> None:
        raise NotImplementedError

    async def flush(self) -> None:
        return None

    def close(self) -> None:
        return None


class StreamTransport(RpcTransport):
    def __init__(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter,
                 codec: Any = MsgpackCodec):
        self.reader = reader
        self.writer = writer
        self.codec = codec
        self._pending: list = []        # frames written since the last flush: they leave in one send

    async def read(self) -> Tuple[int, bytes]:
        try:
            header = await self.reader.readexactly(_HEADER.size)
            length, ftype = _HEADER.unpack(header)
            if length < 1 or length > MAX_FRAME_BYTES:
                raise FrameError(f"bad frame length {length}")
            payload = await self.reader.readexactly(length - 1) if length > 1 else b""
        except asyncio.IncompleteReadError as e:
            raise RPCConnectionError("stream closed") from e
        return ftype, payload

    async def write(self, ftype: int, payload: bytes) -> None:
        self._pending.append(_HEADER.pack(len(payload) + 1, ftype))
        if payload:
            self._pending.append(payload)

    async def flush(self) -> None:
        if self._pending:
            frames, self._pending = self._pending, []
            # a call's sideband buffer and its message: one system call, not four
            self.writer.write(b"".join(frames) if sum(map(len, frames)) <= 65536 else frames[0])
            if sum(map(len, frames)) > 65536:
                self.writer.writelines(frames[1:])
        await self.writer.drain()

    def close(self) -> None:
        try:
            self.writer.close()
        except Exception:
            pass


class ConnectionTransport(RpcTransport):
    """``multiprocessing.Connection``: blocking recv on a one-thread executor,
    sends on another so a full pipe never stalls the event loop."""

    def __init__(self, connection: Any, codec: Any = MsgpackCodec):
        self.connection = connection
        self.codec = codec
        self.executor = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix="rpc-recv")
        self._send_executor = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix="rpc-send")

    def _recv(self) -> Tuple[int, bytes]:
        try:
            data = self.connection.recv_bytes()
        except (EOFError, OSError) as e:
            raise RPCConnectionError("connection closed") from e
        if not data:
            raise FrameError("empty frame")
        return data[0], data[1:]

    async def read(self) -> Tuple[int, bytes]:
        return await asyncio.get_running_loop().run_in_executor(self.executor, self._recv)

    def _send(self, data: bytes) -> None:
        try:
            self.connection.send_bytes(data)
        except (OSError, ValueError) as e:
            raise RPCConnectionError("connection closed") from e

    async def write(self, ftype: int, payload: bytes) -> None:
        await asyncio.get_running_loop().run_in_executor(
            self._send_executor, self._send, bytes([ftype]) + payload)

    def close(self) -> None:
        try:
            self.connection.close()
        except Exception:
            pass
        self.executor.shutdown(wait=False)
        self._send_executor.shutdown(wait=False)


class FileTransport(RpcTransport):
    """Frames over a pair of file descriptors (JSON messages by default so a
    non-Python peer can speak it)."""

    def __init__(self, read_fd: int, write_fd: int, codec: Any = JsonCodec):
        self.read_fd = read_fd
        self.write_fd = write_fd
        self.codec = codec
        self.executor = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix="rpc-fd")

    def _read_exact(self, n: int) -> bytes:
        chunks = []
        while n > 0:
            b = os.read(self.read_fd, n)
            if not b:
                raise RPCConnectionError("fd closed")
            chunks.append(b)
            n -= len(b)
        return b"".join(chunks)

    def _recv(self) -> Tuple[int, bytes]:
        length, ftype = _HEADER.unpack(self._read_exact(_HEADER.size))
        if length < 1 or length > MAX_FRAME_BYTES:
            raise FrameError(f"bad frame length {length}")
        return ftype, self._read_exact(length - 1)

    async def read(self) -> Tuple[int, bytes]:
        return await asyncio.get_running_loop().run_in_executor(self.executor, self._recv)

    async def write(self, ftype: int, payload: bytes) -> None:
        data = encode_frame(ftype, payload)
        view = memoryview(data)
        while view:
            n = os.write(self.write_fd, view)
            view = view[n:]

    def close(self) -> None:
        self.executor.shutdown(wait=False)


class MemoryTransport(RpcTransport):
    """Half of an in-process pipe; ``MemoryTransport.pair()`` builds both."""

    _CLOSED = object()

    def __init__(self, rx: "asyncio.Queue", tx: "asyncio.Queue", codec: Any = MsgpackCodec):
        self.rx, self.tx, self.codec = rx, tx, codec
        self.bytes_written = 0

    @classmethod
    def pair(cls, codec: Any = MsgpackCodec) -> Tuple["MemoryTransport", "MemoryTransport"]:
        a, b = asyncio.Queue(), asyncio.Queue()
        return cls(a, b, codec), cls(b, a, codec)

    async def read(self) -> Tuple[int, bytes]:
        item = await self.rx.get()
        if item is self._CLOSED:
            raise RPCConnectionError("memory transport closed")
        return item

    async def write(self, ftype: int, payload: bytes) -> None:
        self.bytes_written += len(payload) + _HEADER.size
        self.tx.put_nowait((ftype, bytes(payload)))

    def close(self) -> None:
        self.tx.put_nowait(self._CLOSED)
        self.rx.put_nowait(self._CLOSED)


# -------------------------------------------------------------- handshake
class AuthenticationError(ConnectionError):
    pass


def _mac(secret: bytes, role: bytes, nonce: bytes, cnonce: bytes) -> bytes:
    return hmac.new(secret, role + b"|" + nonce + b"|" + cnonce, hashlib.sha256).digest()


def _secret_bytes(secret: Optional[str]) -> bytes:
    return (secret or "").encode()


async def server_handshake(transport: RpcTransport, secret: Optional[str],
                           timeout: float = 10.0) -> None:
    """Challenge the connecting side; raises ``AuthenticationError`` on mismatch.
    Mutual: the client also verifies the server knows the secret."""
    key = _secret_bytes(secret)
    nonce = os.urandom(16)
    await transport.write(FRAME_HANDSHAKE, MsgpackCodec.dumps({"v": 1, "nonce": nonce}))
    await transport.flush()
    ftype, payload = await asyncio.wait_for(transport.read(), timeout)
    if ftype != FRAME_HANDSHAKE:
        raise AuthenticationError("expected handshake frame")
    try:
        reply = MsgpackCodec.loads(payload)
        cnonce, mac = bytes(reply["cnonce"]), bytes(reply["mac"])
    except Exception as e:
        raise AuthenticationError("malformed handshake") from e
    if not hmac.compare_digest(mac, _mac(key, b"client", nonce, cnonce)):
        await transport.write(FRAME_HANDSHAKE, MsgpackCodec.dumps({"ok": False}))
        await transport.flush()
        raise AuthenticationError("client failed authentication (VLLM_RPC_SECRET mismatch)")
    await transport.write(FRAME_HANDSHAKE, MsgpackCodec.dumps(
        {"ok": True, "mac": _mac(key, b"server", nonce, cnonce)}))
    await transport.flush()


async def client_handshake(transport: RpcTransport, secret: Optional[str],
                           timeout: float = 10.0) -> None:
    key = _secret_bytes(secret)
    ftype, payload = await asyncio.wait_for(transport.read(), timeout)
    if ftype != FRAME_HANDSHAKE:
        raise AuthenticationError("expected handshake frame")
    try:
        nonce = bytes(MsgpackCodec.loads(payload)["nonce"])
    except Exception as e:
        raise AuthenticationError("malformed handshake") from e
    cnonce = os.urandom(16)
    await transport.write(FRAME_HANDSHAKE, MsgpackCodec.dumps(
        {"cnonce": cnonce, "mac": _mac(key, b"client", nonce, cnonce)}))
    await transport.flush()
    ftype, payload = await asyncio.wait_for(transport.read(), timeout)
    try:
        reply = MsgpackCodec.loads(payload)
    except Exception as e:
        raise AuthenticationError("malformed handshake") from e
    if ftype != FRAME_HANDSHAKE or not reply.get("ok"):
        raise AuthenticationError("server rejected authentication (VLLM_RPC_SECRET mismatch)")
    if not hmac.compare_digest(bytes(reply.get("mac", b"")), _mac(key, b"server", nonce, cnonce)):
        raise AuthenticationError("server failed authentication")


# -------------------------------------------------------------- read loop
async def prepare_peer_readloop(loop: asyncio.AbstractEventLoop, transport: RpcTransport,
                                self_name: str = "local", peer_name: str = "remote",
                                ) -> Tuple[RpcPeer, Callable[[], Any]]:
    """Build a peer on ``transport`` and the coroutine function that pumps it.

    The returned ``read_loop()`` runs until the transport fails or closes, then
    kills the peer (rejecting all pending calls) and re-raises.  Sideband
    buffers that precede a message are delivered with it, in order
    (reference ``src/rpc_reader.py:209-254``).
    """
    await transport.prepare()
    write_lock = asyncio.Lock()
    codec = transport.codec

    async def send(message: dict, buffers: Optional[List[bytes]]) -> None:
        data = codec.dumps(message)
        async with write_lock:  # buffers and their message stay adjacent
            if buffers:
                for b in buffers:
                    await transport.write(FRAME_BUFFER, b)
            await transport.write(FRAME_MESSAGE, data)
            await transport.flush()

    peer = RpcPeer(self_name, peer_name, send)
    peer.loop = loop
    tasks: set = set()

    async def read_loop() -> None:
        buffers: List[bytes] = []
        try:
            while True:
                ftype, payload = await transport.read()
                if ftype == FRAME_BUFFER:
                    buffers.append(payload)
                    continue
                if ftype != FRAME_MESSAGE:
                    raise FrameError(f"unexpected frame type {ftype}")
                message = codec.loads(payload)
                if not isinstance(message, dict):
                    raise FrameError("message is not a map")
                msg_buffers, buffers = buffers, []
                task = loop.create_task(peer.handle_message(message, msg_buffers))
                tasks.add(task)
                task.add_done_callback(_task_done)
        except BaseException as e:
            peer.kill(e)
            raise
        finally:
            transport.close()

    def _task_done(task: "asyncio.Task") -> None:
        tasks.discard(task)
        if not task.cancelled() and task.exception() is not None:
            peer.kill(task.exception())
            transport.close()

    return peer, read_loop
