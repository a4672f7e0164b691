#!/bin/bash
# round 2: rocprofv3 kernel stats of the default bench command, PMC traffic of the streaming GEMM, DeepSeek / fp8 tests,
# config 1 and config 3 lines
OUT=gpurun_out/r02_prof1; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_deepseek.py -x -q -m gpu > $OUT/pytest_fp8_deepseek.txt 2>&1; grep -E "^FAILED|^E   |passed|failed" $OUT/pytest_fp8_deepseek.txt | head -12
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
find $OUT/bench_prof -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
tail -c 400 $OUT/bench_under_rocprof.json; echo
for f in $(find $OUT/bench_prof -name "*kernel_stats.csv"); do python tools/stats_top.py $f | head -14; done
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stream_trace -- python $R/tools/pmc_target.py stream > $R/$OUT/stream_trace.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/$OUT/stream_fetch -- python $R/tools/pmc_target.py stream > $R/$OUT/stream_fetch.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/$OUT/stream_write -- python $R/tools/pmc_target.py stream > $R/$OUT/stream_write.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$OUT/stream_sq -- python $R/tools/pmc_target.py stream > $R/$OUT/stream_sq.log 2>&1 )
find $OUT -name "*kernel_trace.csv" -size +2M -delete
python - <<'PY'
import csv, glob, collections
for tag in ("fetch", "write", "sq"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02_prof1/stream_{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "stream_gemm_glds" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(tag, {k: (round(sum(v) / len(v)), len(v)) for k, v in acc.items()})
for f in glob.glob("gpurun_out/r02_prof1/stream_trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "stream_gemm" in row["Name"]: print("stream_gemm avg ns", row["AverageNs"], "calls", row["Calls"])
PY
grep algorithmic $OUT/stream_trace.log
timeout 600 python bench.py --model opt-125m --num-requests 32 --input-len 128 --output-len 64 --request-rate 0 --no-static-split-wave > $OUT/bench_config1_opt125m.json 2> $OUT/bench_config1.err; tail -c 700 $OUT/bench_config1_opt125m.json; echo
timeout 900 python bench.py --model deepseek-v2-lite --no-cpu-baseline > $OUT/bench_config3_deepseek_v2_lite.json 2> $OUT/bench_config3.err; python -c "
import json; d=json.loads(open('$OUT/bench_config3_deepseek_v2_lite.json').read().strip().splitlines()[-1]); print('dsv2lite', d['value'], d['p50_ttft_ms'], d['p50_tbt_ms'], d['p99_tbt_ms'], d.get('static_split_50_50'))"
