"""Stage-to-stage transfer of ``(hidden, residual)`` for pipeline parallelism.

The reference runs up to ``PP`` batches in flight (reference
``src/launch.py:298-302``) and leaves the transfer to vLLM's NCCL send/recv on
the compute stream.  Here a stage boundary gets

* its own two-rank RCCL communicator per edge (``stage s`` <-> ``stage s+1``,
  bootstrapped over the PP group's CPU channel), so transfers never queue
  behind the TP collectives of either stage and a middle stage can receive and
  send at the same time (one communicator is only ever driven from one stream);
* a receive stream and a send stream with events: the compute stream waits for
  the incoming tensors only when the first layer needs them, and the outgoing
  transfer of batch *i* overlaps the compute of batch *i+1*;
* one message per boundary crossing: ``hidden`` and ``residual`` travel as one
  grouped RCCL operation (one launch), or as one concatenated tensor on gloo.

Without RCCL (CPU, or several ranks sharing a device) the same interface runs
over ``torch.distributed`` on the PP group.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..logger import init_logger
from .state import GroupCoordinator

logger = init_logger(__name__)


class PipelineLink:
    def __init__(self, pp: GroupCoordinator, device: torch.device, use_rccl: bool = True):
        self.pp = pp
        self.device = device
        self.gpu = device.type == "cuda"
        self.prev = self.next = None          # edge communicators (RCCL), if any
        self.recv_stream = self.send_stream = None
        self._send_done = None                # event: the last outgoing transfer has read its buffers
        self._keep = None                     # tensors of the transfer in flight
        if self.gpu:
            self.recv_stream = torch.cuda.Stream(priority=-1)
            self.send_stream = torch.cuda.Stream(priority=-1)
        if self.gpu and use_rccl and pp.world_size > 1 and not pp.staged:
            try:
                self._make_edge_communicators()
            except Exception as e:  # noqa: BLE001 - torch.distributed carries the transfers instead
                logger.warning("pipeline edge communicators unavailable (%r): using torch.distributed send/recv", e)
                self.prev = self.next = None

    # ------------------------------------------------------------------ setup
    def _make_edge_communicators(self) -> None:
        import ctypes

        from .rccl import RcclCommunicator, _check, load_rccl, ncclUniqueId
        lib = load_rccl()
        pp, r = self.pp, self.pp.rank_in_group
        uid = None
        if not pp.is_last_rank:               # the lower rank of an edge names it
            u = ncclUniqueId()
            _check(lib.ncclGetUniqueId(ctypes.byref(u)))
            uid = bytes(u.internal)
        ids = pp.bootstrap_all_gather(uid, "edge-ids")
        # every rank joins its upstream edge first: stage 0 has none and opens the chain
        if not pp.is_first_rank:
            self.prev = RcclCommunicator(rank=1, world_size=2, unique_id=ids[r - 1])
        if not pp.is_last_rank:
            self.next = RcclCommunicator(rank=0, world_size=2, unique_id=ids[r])
        ok = all(pp.all_gather_object(True))
        if ok:
            logger.info("pp stage %d: dedicated edge communicators active", r)

    @property
    def accelerated(self) -> bool:
        return self.prev is not None or self.next is not None

    # --------------------------------------------------------------- transfer
    def recv(self, hidden: torch.Tensor, residual: Optional[torch.Tensor]) -> None:
        """Fill ``hidden`` (and ``residual``) from the previous stage.  On the GPU the transfer
        runs on the receive stream; the current stream waits for it before returning control,
        i.e. at the point where the first layer would read the tensors."""
        pp = self.pp
        if self.prev is not None:
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)                  # the buffers are free once earlier compute has used them
            self.recv_stream.wait_event(ready)
            with torch.cuda.stream(self.recv_stream):
                self.prev.group_start()
                try:
                    self.prev.recv(hidden, 0)
                    if residual is not None:
                        self.prev.recv(residual, 0)
                finally:
                    self.prev.group_end()
                done = torch.cuda.Event()
                done.record(self.recv_stream)
            cur.wait_event(done)
            return
        if residual is None:
            got = pp.recv(hidden.shape, hidden.dtype)
            hidden.copy_(got)
            return
        got = pp.recv((2,) + tuple(hidden.shape), hidden.dtype)      # one message
        hidden.copy_(got[0])
        residual.copy_(got[1])

    def send(self, hidden: torch.Tensor, residual: Optional[torch.Tensor]) -> None:
        """Ship ``hidden`` (and ``residual``) to the next stage without holding up the compute
        stream: the send stream waits for the producer, the caller goes on."""
        pp = self.pp
        if self.next is not None:
            cur = torch.cuda.current_stream()
            produced = torch.cuda.Event()
            produced.record(cur)
            self.send_stream.wait_event(produced)
            with torch.cuda.stream(self.send_stream):
                self.next.group_start()
                try:
                    self.next.send(hidden, 1)
                    if residual is not None:
                        self.next.send(residual, 1)
                finally:
                    self.next.group_end()
                self._send_done = torch.cuda.Event()
                self._send_done.record(self.send_stream)
            self._keep = (hidden, residual)    # alive until the transfer has read them
            for t in self._keep:
                if t is not None:
                    t.record_stream(self.send_stream)
            return
        if residual is None:
            pp.send(hidden)
        else:
            pp.send(torch.stack([hidden, residual]))                 # one message

    def wait_send(self) -> None:
        """Make the current stream wait until the last outgoing transfer has read its buffers
        (before a static output buffer is overwritten by the next batch)."""
        if self._send_done is not None:
            torch.cuda.current_stream().wait_event(self._send_done)

    def close(self) -> None:
        for c in (self.prev, self.next):
            if c is not None:
                c.close()
        self.prev = self.next = None
