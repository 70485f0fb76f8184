"""uvicorn launcher: serves a FastAPI app on a pre-bound socket, handles
SIGINT/SIGTERM, optional TLS (with certificate refresh on change)."""
from __future__ import annotations

import asyncio
import os
import signal
import socket
from typing import Any, Optional

import uvicorn
from fastapi import FastAPI

from ..logger import init_logger

logger = init_logger(__name__)


class _SSLRefresher:
    """Reload the server's TLS context when the key/cert files change."""

    def __init__(self, config: uvicorn.Config, paths, period: float = 5.0):
        self.config, self.paths, self.period = config, [p for p in paths if p], period
        self._mtimes = {p: self._mtime(p) for p in self.paths}
        self._task: Optional[asyncio.Task] = None

    @staticmethod
    def _mtime(p: str) -> float:
        try:
            return os.stat(p).st_mtime
        except OSError:
            return 0.0

    def start(self) -> None:
        self._task = asyncio.get_running_loop().create_task(self._loop())

    def stop(self) -> None:
        if self._task:
            self._task.cancel()

    async def _loop(self) -> None:
        while True:
            await asyncio.sleep(self.period)
            changed = [p for p in self.paths if self._mtime(p) != self._mtimes[p]]
            if changed and self.config.ssl is not None:
                try:
                    self.config.ssl.load_cert_chain(self.config.ssl_certfile, self.config.ssl_keyfile)
                    if self.config.ssl_ca_certs:
                        self.config.ssl.load_verify_locations(self.config.ssl_ca_certs)
                    logger.info("reloaded TLS certificates")
                except Exception as e:  # noqa: BLE001
                    logger.error("TLS reload failed: %r", e)
                for p in changed:
                    self._mtimes[p] = self._mtime(p)


async def serve_http(app: FastAPI, sock: Optional[socket.socket], enable_ssl_refresh: bool = False,
                     **uvicorn_kwargs: Any):
    """Serve until SIGINT/SIGTERM (or cancellation); returns the coroutine that
    completes the server's shutdown."""
    for k in ("ssl_keyfile", "ssl_certfile", "ssl_ca_certs"):
        if uvicorn_kwargs.get(k) is None:
            uvicorn_kwargs.pop(k, None)
    if "ssl_certfile" not in uvicorn_kwargs:
        uvicorn_kwargs.pop("ssl_cert_reqs", None)
    config = uvicorn.Config(app, **uvicorn_kwargs)
    config.load()
    server = uvicorn.Server(config)
    loop = asyncio.get_running_loop()
    server_task = loop.create_task(server.serve(sockets=[sock] if sock else None))
    refresher = None
    if enable_ssl_refresh and config.ssl is not None:
        refresher = _SSLRefresher(config, [config.ssl_keyfile, config.ssl_certfile, config.ssl_ca_certs])
        refresher.start()

    def on_signal() -> None:
        server.should_exit = True

    try:
        loop.add_signal_handler(signal.SIGINT, on_signal)
        loop.add_signal_handler(signal.SIGTERM, on_signal)
    except (NotImplementedError, RuntimeError, ValueError):
        pass      # not the main thread (tests)
    try:
        await server_task
    except asyncio.CancelledError:
        server.should_exit = True
        await asyncio.wait([server_task], timeout=5)
    finally:
        if refresher:
            refresher.stop()

    async def _done() -> None:
        return None
    return _done()
