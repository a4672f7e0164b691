# shared-KV extend attention: parity, then the kbench extend table with the new kernel and with the round-1 kernel
O=gpurun_out/r02_skv; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "extend" -x 2>&1 | tail -15 > $O/pytest_extend.txt
cat $O/pytest_extend.txt
timeout 600 python tools/kbench.py extend > $O/kbench_extend_shared_kv.txt 2>&1
SEMIPD_EXTEND_SHARED_KV=0 timeout 600 python tools/kbench.py extend > $O/kbench_extend_one_head.txt 2>&1
grep -v amdgpu $O/kbench_extend_shared_kv.txt; grep -v amdgpu $O/kbench_extend_one_head.txt
