O=gpurun_out/r02_v3slice2; mkdir -p $O
timeout 900 python bench.py --model deepseek-v3-slice --quantization fp8 --num-requests 96 --request-rate 8 --no-cpu-baseline --no-static-split-wave > $O/bench_deepseek_v3_slice_fp8.json 2> $O/err.txt
tail -c 1500 $O/bench_deepseek_v3_slice_fp8.json; tail -5 $O/err.txt
SEMIPD_MLA_SHARED=0 timeout 900 python bench.py --model deepseek-v3-slice --quantization fp8 --num-requests 96 --request-rate 8 --no-cpu-baseline --no-static-split-wave > $O/bench_deepseek_v3_slice_fp8_wide.json 2> $O/err2.txt
tail -c 1500 $O/bench_deepseek_v3_slice_fp8_wide.json
