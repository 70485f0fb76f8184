"""Control-plane RPC: symmetric object-proxy peers over framed transports."""
from .peer import (RPCConnectionError, RPCResultError, RPCTimeoutError, RpcPeer,
                   RpcProxy, RpcProxyMethod, RpcSerializer, maybe_await)
from .transport import (AuthenticationError, ConnectionTransport, FileTransport,
                        FrameDecoder, FrameError, JsonCodec, MemoryTransport,
                        MsgpackCodec, RpcTransport, StreamTransport,
                        client_handshake, encode_frame, prepare_peer_readloop,
                        server_handshake, FRAME_BUFFER, FRAME_HANDSHAKE, FRAME_MESSAGE)

__all__ = [n for n in dir() if not n.startswith("_")]
