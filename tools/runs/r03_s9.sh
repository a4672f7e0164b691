#!/bin/bash
# round 3, call 9: rocprofv3 kernel stats of the default bench command (both instances) + PMC passes of the tiled GEMM
OUT=gpurun_out/r03_s9; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/bench_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --steps 2 --warmup 1 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err )
find $OUT/bench_prof -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
tail -c 1500 $OUT/bench_under_rocprof.json | cut -c1-1500; echo
for f in $(find $OUT/bench_prof -name "*kernel_stats.csv"); do python tools/stats_top.py $f | head -20; done
( cd /tmp && export TMPDIR=/tmp
  for t in gemm_tall256 gemm_tall4k; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${t}_trace -- python $R/tools/pmc_target.py $t > $R/$OUT/${t}_trace.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/$OUT/${t}_fetch -- python $R/tools/pmc_target.py $t > $R/$OUT/${t}_fetch.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/$OUT/${t}_write -- python $R/tools/pmc_target.py $t > $R/$OUT/${t}_write.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$OUT/${t}_sq -- python $R/tools/pmc_target.py $t > $R/$OUT/${t}_sq.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $R/$OUT/${t}_mfma -- python $R/tools/pmc_target.py $t > $R/$OUT/${t}_mfma.log 2>&1
  done )
find $OUT -name "*kernel_trace.csv" -size +2M -delete
python - <<'PY'
import csv, glob, collections
for t in ("gemm_tall256", "gemm_tall4k"):
    for tag in ("fetch", "write", "sq", "mfma"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"gpurun_out/r03_s9/{t}_{tag}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "gemm8p_kernel" in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(t, tag, {k: (round(sum(v) / len(v)), len(v)) for k, v in acc.items()})
    for f in glob.glob(f"gpurun_out/r03_s9/{t}_trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm8p" in row["Name"]: print(t, "avg ns", row["AverageNs"], "calls", row["Calls"], row["Name"][:60])
    import subprocess
    print(open(f"gpurun_out/r03_s9/{t}_trace.log").read().strip().splitlines()[-1])
PY
