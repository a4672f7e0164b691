"""Where the time to the first token goes: with SEMIPD_TTFT_TRACE=<dir> every process on the path appends
`<wall clock> <event> <rid,rid,...>` lines to <dir>/<pid>.log (one host, one clock); tools/ttft_trace.py turns them
into per-hop percentiles.  Off (one dict lookup per call) without the variable."""
from __future__ import annotations

import os
import time

_DIR = os.environ.get("SEMIPD_TTFT_TRACE")
_fh = None


def mark(event: str, rids) -> None:
    if not _DIR:
        return
    global _fh
    if _fh is None:
        os.makedirs(_DIR, exist_ok=True)
        _fh = open(os.path.join(_DIR, f"{os.getpid()}.log"), "a", buffering=1)
    _fh.write(f"{time.time():.6f} {event} {','.join(rids)}\n")
