O=gpurun_out/r02_kbench_final; mkdir -p $O
{ echo "# round 2, final tree: tools/kbench.py all + stream_linear + fp8 (isolated kernels, one MI355X)"; timeout 900 python tools/kbench.py all; timeout 300 python tools/kbench.py stream_linear; timeout 300 python tools/kbench.py fp8; timeout 200 python tools/kbench_ext_quick.py; } > $O/kbench_all.txt 2>&1
tail -5 $O/kbench_all.txt; wc -l $O/kbench_all.txt
