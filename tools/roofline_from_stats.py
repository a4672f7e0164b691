"""The decode instance's dominant kernel from a rocprofv3 *_kernel_stats.csv of the bench command, the way VERDICT r05 recomputed it:
every stream_gemm_glds_kernel launch of the process (one launch per dense layer of a decoder layer: qkv, o_proj, gate_up + SiLU, down),
total time / launches = average launch, against the algorithmic bytes of an average launch (a Llama-3-8B layer's 436.2 MB of weights + its
activations over its four launches: bench.py's `algorithmic_bytes_per_launch`, ~109.7 MB at 24-32 rows).

    python tools/roofline_from_stats.py profiles/r06_bench_n1_decode_process_kernel_stats.csv [bytes_per_launch]
"""
import csv
import sys

path = sys.argv[1]
bytes_per_launch = float(sys.argv[2]) if len(sys.argv) > 2 else 109.7e6
rows = list(csv.DictReader(open(path)))
sel = [r for r in rows if "stream_gemm_glds_kernel" in r["Name"] and ", true>" not in r["Name"].split("(")[0].replace("false, true", "")]
n = sum(int(r["Calls"]) for r in sel)
t = sum(float(r["TotalDurationNs"]) for r in sel)
print(f"{path}: {n} launches of stream_gemm_glds_kernel, {t / 1e9:.2f} s -> {t / n / 1e3:.1f} us per launch; "
      f"{bytes_per_launch / 1e6:.1f} MB per launch -> {bytes_per_launch / (t / n):.2f} GB/s = {bytes_per_launch / (t / n) / 8000:.3f} of 8 TB/s")
for key, label in (("decode_rope_attn_kernel", "fused decode launch"), ("decode_mfma_kernel", "decode attention stage 1"), ("rmsnorm_vec_kernel", "RMSNorm")):
    s2 = [r for r in rows if key in r["Name"]]
    if s2:
        n2 = sum(int(r["Calls"]) for r in s2)
        t2 = sum(float(r["TotalDurationNs"]) for r in s2)
        print(f"  {label}: {n2} launches, {t2 / n2 / 1e3:.1f} us per launch")
