"""SEMIPD_TTFT_TRACE marks (semi_pd/ttft_trace.py) and tools/ttft_trace.py: the hop table is computed per request from the
LAST proposal before its admission, requests without a complete trace are left out, and the module is a no-op without
the variable."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_marks_are_written_per_process_and_the_tool_reads_them(tmp_path, monkeypatch):
    monkeypatch.setenv("SEMIPD_TTFT_TRACE", str(tmp_path))
    from semi_pd_amd.semi_pd import ttft_trace
    tt = importlib.reload(ttft_trace)
    try:
        tt.mark("client_send", ["a"])
        tt.mark("p_recv", ["a", "b"])
        logs = list(tmp_path.glob("*.log"))
        assert len(logs) == 1 and logs[0].name == f"{os.getpid()}.log"
        lines = logs[0].read_text().splitlines()
        assert [ln.split()[1:] for ln in lines] == [["client_send", "a"], ["p_recv", "a,b"]]
    finally:
        monkeypatch.delenv("SEMIPD_TTFT_TRACE")
        importlib.reload(ttft_trace)
    # a synthetic run: request x is proposed twice (refused once), y never gets its first token
    order = ["client_send", "p_recv", "p_propose", "d_got_proposal", "p_admitted", "p_launched", "p_done", "d_got_result",
             "d_streamed", "client_first_token"]
    t0 = 1000.0
    rows = []
    for rid, base in (("x", 0.0), ("z", 0.050)):
        times = [0.0, 0.001, 0.002, 0.0021, 0.0025, 0.0075, 0.0250, 0.0251, 0.0252, 0.0253]
        for ev, t in zip(order, times):
            rows.append((t0 + base + t, ev, rid))
    rows.append((t0 + 0.0015, "p_propose", "x"))          # an earlier, refused proposal of x
    rows.append((t0 + 0.0016, "d_got_proposal", "x"))
    rows.append((t0 + 0.1, "client_send", "y"))
    d = tmp_path / "run"
    d.mkdir()
    (d / "1.log").write_text("".join(f"{t:.6f} {ev} {rid}\n" for t, ev, rid in rows if ev.startswith(("client", "d_"))))
    (d / "2.log").write_text("".join(f"{t:.6f} {ev} {rid}\n" for t, ev, rid in rows if ev.startswith("p_")))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ttft_trace.py"), str(d)], capture_output=True,
                         text=True, check=True).stdout
    assert "2 requests with a complete trace" in out
    hop = {ln.split("->")[0].strip(): ln for ln in out.splitlines() if "->" in ln and "mean" in ln}
    assert "p50    1.00" in hop["p_recv"]                   # p_recv -> the ADMITTING proposal (2 ms), not the refused one
    assert "p50   17.50" in hop["p_launched"]
    assert "mean 25.30" in out.splitlines()[-2]


def test_mark_is_a_no_op_without_the_variable(tmp_path, monkeypatch):
    monkeypatch.delenv("SEMIPD_TTFT_TRACE", raising=False)
    from semi_pd_amd.semi_pd import ttft_trace
    tt = importlib.reload(ttft_trace)
    tt.mark("client_send", ["a"])
    assert not list(tmp_path.iterdir())
