"""PUSH / PULL message sockets between the Semi-PD processes.

The reference uses ZeroMQ ipc:// sockets (get_zmq_socket, utils.py; SemiPDPortArgs in
server_args.py:1117-1195).  pyzmq is not part of this image, so the same topology is built on
multiprocessing.connection over AF_UNIX sockets: a PULL end listens and accepts any number of PUSH
ends; messages are pickled Python objects; order is preserved per sender.
"""
from __future__ import annotations

import os
import threading
import time
from multiprocessing.connection import Client, Connection, Listener, wait
from typing import Any, List, Optional


class PullSocket:
    def __init__(self, address: str):
        self.address = address
        if os.path.exists(address):
            os.unlink(address)
        self._listener = Listener(address, family="AF_UNIX", backlog=64)
        self._conns: List[Connection] = []
        self._lock = threading.Lock()
        self._closed = False
        self._thread = threading.Thread(target=self._accept_loop, daemon=True)
        self._thread.start()

    def _accept_loop(self):
        while not self._closed:
            try:
                c = self._listener.accept()
            except (OSError, EOFError):
                return
            with self._lock:
                self._conns.append(c)

    def recv_pyobj(self, timeout: Optional[float] = None) -> Any:
        """Blocking receive (timeout None = forever); raises TimeoutError."""
        deadline = None if timeout is None else time.monotonic() + timeout
        while True:
            obj = self.recv_pyobj_nowait()
            if obj is not _NOTHING:
                return obj
            with self._lock:
                conns = list(self._conns)
            remaining = None if deadline is None else max(0.0, deadline - time.monotonic())
            if deadline is not None and remaining == 0.0:
                raise TimeoutError(f"no message on {self.address} within {timeout}s")
            if conns:
                wait(conns, 0.05 if remaining is None else min(0.05, remaining))
            else:
                time.sleep(0.001)

    def recv_pyobj_nowait(self) -> Any:
        """One message if any is ready, else the NOTHING sentinel (zmq.NOBLOCK drain loop,
        scheduler.py:599-612)."""
        with self._lock:
            conns = list(self._conns)
        for c in conns:
            try:
                if c.poll(0):
                    return c.recv()
            except (EOFError, OSError):
                with self._lock:
                    if c in self._conns:
                        self._conns.remove(c)
        return _NOTHING

    def close(self):
        self._closed = True
        try:
            self._listener.close()
        except OSError:
            pass
        with self._lock:
            for c in self._conns:
                c.close()
            self._conns.clear()
        if os.path.exists(self.address):
            try:
                os.unlink(self.address)
            except OSError:
                pass


class _Nothing:
    def __repr__(self):
        return "NOTHING"


_NOTHING = _Nothing()
NOTHING = _NOTHING


class PushSocket:
    def __init__(self, address: str, connect_timeout: float = 120.0):
        self.address = address
        self._conn: Optional[Connection] = None
        self._timeout = connect_timeout
        self._lock = threading.Lock()

    def _connect(self):
        deadline = time.monotonic() + self._timeout
        while True:
            try:
                self._conn = Client(self.address, family="AF_UNIX")
                return
            except (FileNotFoundError, ConnectionRefusedError):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"cannot connect to {self.address}")
                time.sleep(0.01)

    def send_pyobj(self, obj: Any):
        with self._lock:
            if self._conn is None:
                self._connect()
            self._conn.send(obj)

    def close(self):
        if self._conn is not None:
            self._conn.close()
            self._conn = None
