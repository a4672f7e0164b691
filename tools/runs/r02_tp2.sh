O=gpurun_out/r02_tp2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "abort_of_a_failed" 2>&1 | tail -3
SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 1 --num-requests 48 --request-rate 8 --no-cpu-baseline --mem-fraction-static 0.3 > $O/bench_tp2_one_gpu.json 2> $O/bench_tp2.err
tail -c 1800 $O/bench_tp2_one_gpu.json; echo; grep -v "amdgpu.ids\|^frame #\|UserWarning\|warnings.warn\|socket.cpp\|Gloo" $O/bench_tp2.err | tail -12 | cut -c1-300
