#!/bin/bash
# round 3: HBM traffic (PMC) of the streaming GEMM on the 96-CU decode share of the default policy
OUT=gpurun_out/r03_pmc_stream; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && export HSA_CU_MASK=0:160-255
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python $R/tools/pmc_target.py stream > $R/$OUT/trace.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/$OUT/fetch -- python $R/tools/pmc_target.py stream > $R/$OUT/fetch.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/$OUT/write -- python $R/tools/pmc_target.py stream > $R/$OUT/write.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$OUT/sq -- python $R/tools/pmc_target.py stream > $R/$OUT/sq.log 2>&1 )
find $OUT -name "*kernel_trace.csv" -size +2M -delete
python - <<'PY'
import csv, glob, collections
for tag in ("fetch", "write", "sq"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r03_pmc_stream/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "stream_gemm_glds" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(tag, {k: (round(sum(v) / len(v)), len(v)) for k, v in acc.items()})
for f in glob.glob("gpurun_out/r03_pmc_stream/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "stream_gemm" in row["Name"]: print("stream_gemm avg ns", row["AverageNs"], "calls", row["Calls"])
print(open("gpurun_out/r03_pmc_stream/trace.log").read().strip().splitlines()[-1])
PY
