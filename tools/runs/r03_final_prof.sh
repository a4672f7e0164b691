#!/bin/bash
# round 3, final tree: rocprofv3 kernel stats of the default bench command (without the extra waves), config 3 line with
# its saturation wave
OUT=gpurun_out/r03_final_prof; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
for f in $(find /tmp/bench_prof -name "*kernel_stats.csv"); do n=$(python tools/stats_top.py $f | grep -c "extend_attn"); if [ "$n" -gt 0 ]; then cp $f $OUT/prefill_process_kernel_stats.csv; else if [ $(wc -l < $f) -gt 20 ]; then cp $f $OUT/decode_process_kernel_stats.csv; fi; fi; done
tail -c 600 $OUT/bench_under_rocprof.json; echo
for f in $OUT/*_kernel_stats.csv; do python tools/stats_top.py $f | head -10; done
timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline --rate-sweep "" --steps 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python -c "
import json; d=json.loads(open('$OUT/bench_c3.json').read().strip().splitlines()[-1]); print('dsv2lite', d['value'], d['p50_ttft_ms'], d['p99_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'], 'sat', d['saturation']['output_tok_s'])"
