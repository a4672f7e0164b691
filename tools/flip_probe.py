"""How far apart are two engines' log-probabilities on the same requests?  Runs the 32-layer Llama-3-8B (dummy weights) through
engines that differ in ONE thing (an environment knob, or Semi-PD against unified), 16 requests x 16 greedy tokens each, and prints
for every variant against the baseline: requests that stayed token-for-token equal, the gap at each first divergence (the
baseline's own log-probability difference between its token and the variant's), and the distribution of |delta logprob| of the
chosen token over all steps the two engines had the same history -- the noise a near-tie rule has to live with.

    python tools/flip_probe.py            (each engine in its own process: the knobs are read at import time)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENS = (64, 200, 1024, 7) * 4
STEPS = 16
BASE = {"SEMIPD_SL_NW": "8", "SEMIPD_SL_WIDE": "0", "SEMIPD_FUSED_DECODE_ATTN": "0"}     # the round-5 kernels
VARIANTS = [
    ("baseline again (run-to-run)", dict(BASE), False),
    ("narrow streaming workgroups", {**BASE, "SEMIPD_SL_NW": "0"}, False),
    ("wide streaming GEMM (65-128 rows)", {**BASE, "SEMIPD_SL_WIDE": "1"}, False),
    ("fused decode launch", {**BASE, "SEMIPD_FUSED_DECODE_ATTN": "1"}, False),
    ("all of round 6", {}, False),
    ("Semi-PD, round-5 kernels", dict(BASE), True),
    ("Semi-PD, all of round 6", {}, True),
]


def child(semi_pd: bool, out_path: str):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-pd_amd")]
    import numpy as np
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    from semi_pd_amd.models.llama import LLAMA3_8B
    from semi_pd_amd.server_args import ServerArgs
    cfg = LLAMA3_8B
    rs = np.random.RandomState(77)
    prompts = [[int((o + j) % cfg.vocab_size) for j in range(n)] for o, n in zip(rs.randint(0, cfg.vocab_size, size=len(LENS)), LENS)]
    eng = Engine(ServerArgs(model_config=cfg, context_length=1100, max_running_requests=32, max_total_tokens=30000,
                            cuda_graph_max_bs=32, watchdog_timeout=300.0, tune_prefill_gemm=False, enable_semi_pd=semi_pd))
    try:
        outs, lps = eng.generate(prompts, SamplingParams(max_new_tokens=STEPS, ignore_eos=True), timeout=600,
                                 return_logprob=True, top_logprobs_num=8)
    finally:
        eng.shutdown()
    json.dump({"tokens": outs, "token_lp": [lp["token"] for lp in lps],
               "top": [[[(float(a), int(b)) for a, b in (e[:2] for e in step)] for step in lp["top"]] for lp in lps]},
              open(out_path, "w"))


def compare(name, base, var):
    equal, gaps, deltas = 0, [], []
    for i, (a, b) in enumerate(zip(base["tokens"], var["tokens"])):
        s = next((j for j in range(len(a)) if a[j] != b[j]), len(a))
        equal += s == len(a)
        for j in range(min(s + 1, len(a))):             # same history up to and including step s
            if j < s:
                deltas.append(abs(base["token_lp"][i][j] - var["token_lp"][i][j]))
        if s < len(a):
            top = dict((t, lp) for lp, t in base["top"][i][s])
            gaps.append((i, s, round(top[a[s]] - top[b[s]], 4) if b[s] in top else None))
    deltas.sort()
    q = lambda p: deltas[min(len(deltas) - 1, int(p * len(deltas)))] if deltas else float("nan")
    print(f"{name:38s}: {equal:2d} of {len(base['tokens'])} requests equal over {STEPS} steps; |d logprob| of the chosen token over "
          f"{len(deltas)} common steps: p50 {q(0.5):.4f} p90 {q(0.9):.4f} p99 {q(0.99):.4f} max {deltas[-1] if deltas else 0:.4f}; "
          f"first divergences (request, step, baseline's gap): {gaps}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2] == "1", sys.argv[3])
        sys.exit(0)
    tmp = os.environ.get("TMPDIR", "/tmp")
    results = []
    for k, (name, env, semi) in enumerate([("baseline", dict(BASE), False)] + VARIANTS):
        path = os.path.join(tmp, f"flip_probe_{k}.json")
        e = dict(os.environ)
        for kk in ("SEMIPD_SL_NW", "SEMIPD_SL_WIDE", "SEMIPD_FUSED_DECODE_ATTN"):
            e.pop(kk, None)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "1" if semi else "0", path], env=e, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        results.append((name, json.load(open(path))))
        if k:
            compare(name, results[0][1], results[-1][1])
    by_name = dict(results)
    # the two engines on the SAME kernels (whether they are bit-identical depends on whether their batches were the same)
    compare("Semi-PD vs unified, both round 6", by_name["all of round 6"], by_name["Semi-PD, all of round 6"])
    compare("Semi-PD vs unified, both round 5", by_name["baseline"], by_name["Semi-PD, round-5 kernels"])
