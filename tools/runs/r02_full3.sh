O=gpurun_out/r02_full3; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
