"""Minimal synchronous RFC 6455 WebSocket transport — thread per connection, blocking ``recv`` / ``send``.

The reference serves its protocol through ``websockets.sync.server.serve`` (whisper_live/server.py:26,874-887); that
wheel is not in this image, so the few pieces of its surface the server actually touches are
restated here over the standard library: ``serve(handler, host, port, process_request=...)`` as a context manager with
``serve_forever()`` / ``shutdown()``; a connection object with ``recv()`` (``str`` for text frames, ``bytes`` for
binary), ``send()``, ``close()``, ``respond(status, text)`` and ``request.path`` / ``request.headers``; and
``ConnectionClosed``. ``connect(uri)`` is the matching client end, used by the tests and by ``bench.py --via-server``.

Scope: the base protocol only — no extensions (permessage-deflate is never offered, so stock clients fall back to
plain frames), no subprotocols. Handles 7/16/64-bit payload lengths, masking, fragmented messages, ping/pong and the
close handshake. Audio packets are 16 KiB binary frames (whisper_live/client.py:433,547), so unmasking is vectorised
with numpy rather than done per byte.
"""
from __future__ import annotations

import base64
import hashlib
import http
import os
import select
import socket
import struct
import threading
from types import SimpleNamespace
from typing import Callable, Dict, Optional, Tuple, Union
from urllib.parse import urlparse

import numpy as np

GUID = b"258EAFA5-E914-47DA-95CA-C5AB0DC85B11"
OP_CONT, OP_TEXT, OP_BINARY, OP_CLOSE, OP_PING, OP_PONG = 0x0, 0x1, 0x2, 0x8, 0x9, 0xA
MAX_MESSAGE_BYTES = 1 << 26          # 64 MiB: far above any audio packet, bounds a hostile length field
MAX_HEADER_BYTES = 1 << 16


class ConnectionClosed(Exception):
    """Raised by recv()/send() once the peer has closed (or the socket died)."""

    def __init__(self, code: int = 1006, reason: str = ""):
        super().__init__(f"connection closed: {code} {reason}".strip())
        self.code, self.reason = code, reason


class Headers(dict):
    """Case-insensitive header lookup (``request.headers.get("Authorization", "")``, server.py:35)."""

    def __setitem__(self, k, v):
        super().__setitem__(k.lower(), v)

    def __getitem__(self, k):
        return super().__getitem__(k.lower())

    def get(self, k, default=None):
        return super().get(k.lower(), default)

    def __contains__(self, k):
        return super().__contains__(k.lower())


def accept_key(key: str) -> str:
    return base64.b64encode(hashlib.sha1(key.encode("ascii") + GUID).digest()).decode("ascii")


def apply_mask(payload: bytes, mask: bytes) -> bytes:
    n = len(payload)
    if n == 0:
        return b""
    a = np.frombuffer(payload, dtype=np.uint8)
    m = np.frombuffer((mask * (n // 4 + 1))[:n], dtype=np.uint8)
    return np.bitwise_xor(a, m).tobytes()


def encode_frame(opcode: int, payload: bytes, mask: bool, fin: bool = True) -> bytes:
    n = len(payload)
    b0 = (0x80 if fin else 0) | opcode
    mbit = 0x80 if mask else 0
    if n < 126:
        head = struct.pack("!BB", b0, mbit | n)
    elif n < (1 << 16):
        head = struct.pack("!BBH", b0, mbit | 126, n)
    else:
        head = struct.pack("!BBQ", b0, mbit | 127, n)
    if not mask:
        return head + payload
    key = os.urandom(4)
    return head + key + apply_mask(payload, key)


class _Reader:
    """Buffered exact reads off a socket."""

    def __init__(self, sock: socket.socket, initial: bytes = b""):
        self.sock, self.buf = sock, bytearray(initial)

    def read(self, n: int) -> bytes:
        while len(self.buf) < n:
            try:
                chunk = self.sock.recv(max(65536, n - len(self.buf)))
            except OSError as e:
                raise ConnectionClosed(1006, str(e)) from None
            if not chunk:
                raise ConnectionClosed(1006, "eof")
            self.buf += chunk
        out = bytes(self.buf[:n])
        del self.buf[:n]
        return out

    def read_until(self, sep: bytes, limit: int) -> bytes:
        while True:
            i = self.buf.find(sep)
            if i >= 0:
                out = bytes(self.buf[: i + len(sep)])
                del self.buf[: i + len(sep)]
                return out
            if len(self.buf) > limit:
                raise ConnectionClosed(1009, "header too large")
            try:
                chunk = self.sock.recv(65536)
            except OSError as e:
                raise ConnectionClosed(1006, str(e)) from None
            if not chunk:
                raise ConnectionClosed(1006, "eof")
            self.buf += chunk


def read_frame(rd: _Reader, expect_mask: Optional[bool]) -> Tuple[bool, int, bytes]:
    b0, b1 = rd.read(2)
    fin, opcode = bool(b0 & 0x80), b0 & 0x0F
    if b0 & 0x70:
        raise ConnectionClosed(1002, "reserved bits set (no extension negotiated)")
    masked, n = bool(b1 & 0x80), b1 & 0x7F
    if n == 126:
        (n,) = struct.unpack("!H", rd.read(2))
    elif n == 127:
        (n,) = struct.unpack("!Q", rd.read(8))
    if n > MAX_MESSAGE_BYTES:
        raise ConnectionClosed(1009, "message too big")
    if opcode >= 0x8 and (n > 125 or not fin):
        raise ConnectionClosed(1002, "bad control frame")
    if expect_mask is not None and masked != expect_mask:
        raise ConnectionClosed(1002, "client frames must be masked, server frames must not")
    key = rd.read(4) if masked else b""
    payload = rd.read(n)
    return fin, opcode, (apply_mask(payload, key) if masked else payload)


class Connection:
    """One end of an established WebSocket. ``is_client`` decides the masking direction."""

    def __init__(self, sock: socket.socket, reader: _Reader, is_client: bool, request=None):
        self.sock, self._rd, self.is_client, self.request = sock, reader, is_client, request
        self._send_lock = threading.Lock()
        self._closed = False            # a close frame was sent or received / the socket is gone
        self.close_code: Optional[int] = None
        self.close_reason = ""
        try:
            self.remote_address = sock.getpeername()
        except OSError:
            self.remote_address = None

    # ---- sending ---------------------------------------------------------------------------------------------
    def _send_frame(self, opcode: int, payload: bytes):
        data = encode_frame(opcode, payload, mask=self.is_client)
        with self._send_lock:
            try:
                self.sock.sendall(data)
            except OSError as e:
                self._closed = True
                raise ConnectionClosed(1006, str(e)) from None

    def send(self, message: Union[str, bytes, bytearray, memoryview]):
        if self._closed:
            raise ConnectionClosed(self.close_code or 1006, self.close_reason)
        if isinstance(message, str):
            self._send_frame(OP_TEXT, message.encode("utf-8"))
        else:
            self._send_frame(OP_BINARY, bytes(message))

    def ping(self, data: bytes = b""):
        self._send_frame(OP_PING, data)

    # ---- receiving -------------------------------------------------------------------------------------------
    def recv(self, timeout: Optional[float] = None) -> Union[str, bytes]:
        if self._closed:
            raise ConnectionClosed(self.close_code or 1006, self.close_reason)
        if timeout is not None and not self._rd.buf:
            # wait for the START of a message only; once bytes are flowing the frame is read to its end, so a
            # timeout can never strand a half-consumed frame
            try:
                ready, _, _ = select.select([self.sock], [], [], timeout)
            except (OSError, ValueError) as e:
                self._closed = True
                raise ConnectionClosed(1006, str(e)) from None
            if not ready:
                raise TimeoutError("recv timed out")
        parts, kind = [], None
        try:
            self.sock.settimeout(None)
            while True:
                fin, opcode, payload = read_frame(self._rd, expect_mask=not self.is_client)
                if opcode == OP_PING:
                    self._send_frame(OP_PONG, payload)
                    continue
                if opcode == OP_PONG:
                    continue
                if opcode == OP_CLOSE:
                    self.close_code = struct.unpack("!H", payload[:2])[0] if len(payload) >= 2 else 1005
                    self.close_reason = payload[2:].decode("utf-8", "replace")
                    if not self._closed:
                        try:
                            self._send_frame(OP_CLOSE, payload[:2])
                        except ConnectionClosed:
                            pass
                    self._closed = True
                    self._shutdown()
                    raise ConnectionClosed(self.close_code, self.close_reason)
                if opcode in (OP_TEXT, OP_BINARY):
                    if kind is not None:
                        raise ConnectionClosed(1002, "new message inside a fragmented one")
                    kind = opcode
                elif opcode == OP_CONT:
                    if kind is None:
                        raise ConnectionClosed(1002, "continuation without a start")
                else:
                    raise ConnectionClosed(1002, f"unknown opcode {opcode}")
                parts.append(payload)
                if sum(map(len, parts)) > MAX_MESSAGE_BYTES:
                    raise ConnectionClosed(1009, "message too big")
                if fin:
                    data = b"".join(parts)
                    return data.decode("utf-8") if kind == OP_TEXT else data
        except ConnectionClosed:
            if not self._closed:
                self._closed = True
                self._shutdown()
            raise

    def __iter__(self):
        try:
            while True:
                yield self.recv()
        except ConnectionClosed:
            return

    # ---- closing ---------------------------------------------------------------------------------------------
    def _shutdown(self):
        try:
            self.sock.shutdown(socket.SHUT_RDWR)
        except OSError:
            pass
        try:
            self.sock.close()
        except OSError:
            pass

    def close(self, code: int = 1000, reason: str = ""):
        """Send a close frame, wait briefly for the echo, drop the socket. Idempotent."""
        if self._closed:
            return
        self._closed = True
        self.close_code = code
        try:
            self._send_frame(OP_CLOSE, struct.pack("!H", code) + reason.encode("utf-8")[:123])
            self.sock.settimeout(1.0)
            for _ in range(64):                       # drain until the peer's close echo (bounded)
                _fin, opcode, _p = read_frame(self._rd, expect_mask=None)
                if opcode == OP_CLOSE:
                    break
        except (ConnectionClosed, OSError, socket.timeout):
            pass
        self._shutdown()

    def respond(self, status, text: str):
        """Build an HTTP response for ``process_request`` to return instead of upgrading (server.py:42)."""
        status = http.HTTPStatus(status)
        return SimpleNamespace(status=status, body=text.encode("utf-8"))


# ---- handshake -----------------------------------------------------------------------------------------------------
def _parse_http_head(raw: bytes) -> Tuple[str, Headers]:
    lines = raw.decode("latin-1").split("\r\n")
    hdr = Headers()
    for ln in lines[1:]:
        if ":" in ln:
            k, v = ln.split(":", 1)
            hdr[k.strip()] = v.strip()
    return lines[0], hdr


def _http_reply(sock: socket.socket, status: http.HTTPStatus, body: bytes, extra: Dict[str, str] = None):
    head = [f"HTTP/1.1 {status.value} {status.phrase}", f"Content-Length: {len(body)}", "Content-Type: text/plain",
            "Connection: close"]
    for k, v in (extra or {}).items():
        head.append(f"{k}: {v}")
    try:
        sock.sendall(("\r\n".join(head) + "\r\n\r\n").encode("latin-1") + body)
    except OSError:
        pass


def server_handshake(sock: socket.socket, process_request: Optional[Callable] = None) -> Optional[Connection]:
    rd = _Reader(sock)
    start, hdr = _parse_http_head(rd.read_until(b"\r\n\r\n", MAX_HEADER_BYTES))
    parts = start.split(" ")
    if len(parts) < 3 or parts[0] != "GET":
        _http_reply(sock, http.HTTPStatus.METHOD_NOT_ALLOWED, b"websocket endpoint\n")
        return None
    request = SimpleNamespace(path=parts[1], headers=hdr)
    conn = Connection(sock, rd, is_client=False, request=request)
    key = hdr.get("Sec-WebSocket-Key")
    if "websocket" not in hdr.get("Upgrade", "").lower() or key is None:
        _http_reply(sock, http.HTTPStatus.UPGRADE_REQUIRED, b"websocket upgrade required\n", {"Upgrade": "websocket"})
        return None
    if hdr.get("Sec-WebSocket-Version", "13") != "13":
        _http_reply(sock, http.HTTPStatus.BAD_REQUEST, b"unsupported websocket version\n", {"Sec-WebSocket-Version": "13"})
        return None
    if process_request is not None:
        resp = process_request(conn, request)
        if resp is not None:
            _http_reply(sock, resp.status, resp.body)
            return None
    sock.sendall(("HTTP/1.1 101 Switching Protocols\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
                  f"Sec-WebSocket-Accept: {accept_key(key)}\r\n\r\n").encode("ascii"))
    return conn


class InvalidStatus(Exception):
    def __init__(self, status: int, body: bytes = b""):
        super().__init__(f"server rejected WebSocket connection: HTTP {status}")
        self.status, self.body = status, body


def connect(uri: str, additional_headers: Optional[Dict[str, str]] = None, open_timeout: float = 10.0) -> Connection:
    u = urlparse(uri)
    if u.scheme != "ws":
        raise ValueError("only ws:// is supported")
    host, port = u.hostname, u.port or 80
    path = (u.path or "/") + (f"?{u.query}" if u.query else "")
    sock = socket.create_connection((host, port), timeout=open_timeout)
    sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
    key = base64.b64encode(os.urandom(16)).decode("ascii")
    head = [f"GET {path} HTTP/1.1", f"Host: {host}:{port}", "Upgrade: websocket", "Connection: Upgrade",
            f"Sec-WebSocket-Key: {key}", "Sec-WebSocket-Version: 13"]
    for k, v in (additional_headers or {}).items():
        head.append(f"{k}: {v}")
    sock.sendall(("\r\n".join(head) + "\r\n\r\n").encode("latin-1"))
    rd = _Reader(sock)
    start, hdr = _parse_http_head(rd.read_until(b"\r\n\r\n", MAX_HEADER_BYTES))
    status = int(start.split(" ")[1])
    if status != 101:
        n = int(hdr.get("Content-Length", "0") or 0)
        body = rd.read(n) if n else b""
        sock.close()
        raise InvalidStatus(status, body)
    if hdr.get("Sec-WebSocket-Accept") != accept_key(key):
        sock.close()
        raise ConnectionClosed(1002, "bad Sec-WebSocket-Accept")
    sock.settimeout(None)
    return Connection(sock, rd, is_client=True)


# ---- server --------------------------------------------------------------------------------------------------------
class Server:
    """Accept loop; every connection gets a daemon thread that runs ``handler(connection)`` and then closes it —
    the threading model of websockets.sync.server (one blocking handler per client; the reference's ``recv_audio`` is
    written against exactly that)."""

    def __init__(self, handler: Callable[[Connection], None], host: str, port: int,
                 process_request: Optional[Callable] = None, backlog: int = 128):
        self.handler, self.process_request = handler, process_request
        self.sock = socket.create_server((host, port), backlog=backlog, reuse_port=False)
        self.sock.settimeout(0.2)
        self.host, self.port = self.sock.getsockname()[:2]
        self._stop = threading.Event()
        self._threads = set()
        self._lock = threading.Lock()

    def _client(self, sock: socket.socket):
        conn = None
        try:
            sock.settimeout(10.0)
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn = server_handshake(sock, self.process_request)
            if conn is None:
                return
            sock.settimeout(None)
            self.handler(conn)
        except (ConnectionClosed, OSError, ValueError):
            pass
        finally:
            if conn is not None:
                conn.close()
            else:
                try:
                    sock.close()
                except OSError:
                    pass
            with self._lock:
                self._threads.discard(threading.current_thread())

    def serve_forever(self):
        while not self._stop.is_set():
            try:
                sock, _addr = self.sock.accept()
            except socket.timeout:
                continue
            except OSError:
                break
            t = threading.Thread(target=self._client, args=(sock,), daemon=True)
            with self._lock:
                self._threads.add(t)
            t.start()

    def shutdown(self):
        self._stop.set()
        try:
            self.sock.close()
        except OSError:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown()


def serve(handler, host: str, port: int, process_request: Optional[Callable] = None, **_ignored) -> Server:
    return Server(handler, host, port, process_request=process_request)
