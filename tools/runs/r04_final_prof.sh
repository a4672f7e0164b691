#!/bin/bash
# round 4: rocprofv3 kernel stats of the default bench command (without the extra waves), config 3 line with its saturation wave,
# the N = 2 command shape of the driver on one GPU (functional)
OUT=gpurun_out/r04_final_prof; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --no-side-configs --rate-sweep "" > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
for f in $(find /tmp/bench_prof -name "*kernel_stats.csv"); do n=$(python tools/stats_top.py $f | grep -c "extend_attn"); if [ "$n" -gt 0 ]; then cp $f $OUT/prefill_process_kernel_stats.csv; else if [ $(wc -l < $f) -gt 20 ]; then cp $f $OUT/decode_process_kernel_stats.csv; fi; fi; done
python - <<PY
import json
d = json.loads(open("$OUT/bench_under_rocprof.json").read().strip().splitlines()[-1])
print("under rocprof:", d["value"], "TTFT", d["p50_ttft_ms"], "TBT", d["p50_tbt_ms"], d["p99_tbt_ms"], "roofline", d["roofline"])
PY
for f in $OUT/*_kernel_stats.csv; do python tools/stats_top.py $f | head -12 | cut -c1-150; done
timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline --no-side-configs --no-static-split-wave --rate-sweep "" --steps 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python -c "
import json; d=json.loads(open('$OUT/bench_c3.json').read().strip().splitlines()[-1]); print('dsv2lite', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'], 'sat', d['saturation']['output_tok_s'], d['config']['workload'][:200])"
SEMIPD_BENCH_ALL_ON_GPU0=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29713 bench.py --gpus 2 --steps 1 --warmup 1 --model llama-tiny --num-requests 8 --request-rate 4 --fixed-load \
  --input-len 256 --output-len 32 --no-cpu-baseline --mem-fraction-static 0.1 --max-total-tokens 20000 --max-running-requests 8 \
  --rate-sweep "" --no-saturation-wave --no-side-configs --no-static-split-wave > $OUT/bench_gpus2_tp_dry_run_one_gpu.json 2> $OUT/bench_gpus2.err
echo "n2 rc=$?"; tail -c 700 $OUT/bench_gpus2_tp_dry_run_one_gpu.json; echo; tail -3 $OUT/bench_gpus2.err | cut -c1-300
