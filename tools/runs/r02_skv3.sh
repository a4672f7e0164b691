O=gpurun_out/r02_skv; mkdir -p $O
SEMIPD_EXTEND_KV_HALVES=1 timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "extend" -x 2>&1 | tail -3
SEMIPD_EXTEND_KV_HALVES=2 timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "extend" -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "extend" -x 2>&1 | tail -3
for h in 1 2 0; do echo "== halves=$h (0 = heuristic)"; SEMIPD_EXTEND_KV_HALVES=$h timeout 600 python tools/kbench.py extend 2>&1 | grep "D=128" ; done | tee $O/kbench_extend_halves.txt
