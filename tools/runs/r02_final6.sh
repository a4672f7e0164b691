O=gpurun_out/r02_final6; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
