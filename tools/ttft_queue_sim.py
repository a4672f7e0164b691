"""The prefill share as a single server: Poisson arrivals, deterministic service
S(n) = c0 + c1 * n milliseconds for a batch of n requests, every waiting request (up to `cap`) joins the next batch,
`gap` milliseconds of idle GPU between batches.  Prints TTFT p50 / mean / p99 for a few policies and shows what one
millisecond of batch time is worth.  python tools/ttft_queue_sim.py [rate] [c0] [c1]"""
import sys

import numpy as np


def sim(cap, c0, c1, rate, n=200000, seed=1, gap=1.0):
    rs = np.random.RandomState(seed)
    arr = np.cumsum(rs.exponential(1000.0 / rate, size=n))
    ttft = np.empty(n)
    i, t = 0, 0.0
    while i < n:
        if t < arr[i]:
            t = arr[i]
        j = i
        while j < n and arr[j] <= t and j - i < cap:
            j += 1
        t_end = t + gap + c0 + c1 * (j - i)
        ttft[i:j] = t_end - arr[i:j]
        t, i = t_end, j
    return np.percentile(ttft, 50), ttft.mean(), np.percentile(ttft, 99)


def main():
    rate = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
    c0 = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    c1 = float(sys.argv[3]) if len(sys.argv) > 3 else 18.3
    print(f"# {rate} req/s, S(n) = {c0} + {c1} n ms, 1 ms between batches: TTFT p50 / mean / p99 (ms)")
    for cap in (1, 2, 3, 8):
        print(f"at most {cap} request(s) per batch:", " / ".join(f"{x:.1f}" for x in sim(cap, c0, c1, rate)))
    print("batches 1 ms shorter:          ", " / ".join(f"{x:.1f}" for x in sim(8, c0 - 1.0, c1, rate)))
    print("no gap between batches:        ", " / ".join(f"{x:.1f}" for x in sim(8, c0, c1, rate, gap=0.0)))


if __name__ == "__main__":
    main()
