#!/bin/bash
# round 3: PMC of the grouped ping-pong GEMM in its 128 x 512 geometry (one DeepSeek-V2-Lite prefill request) on a 128-CU share
OUT=gpurun_out/r03_pmc_moe_mid; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && export HSA_CU_MASK=0:0-127
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python $R/tools/pmc_target.py moe1k > $R/$OUT/trace.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/$OUT/fetch -- python $R/tools/pmc_target.py moe1k > $R/$OUT/fetch.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/$OUT/write -- python $R/tools/pmc_target.py moe1k > $R/$OUT/write.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$OUT/sq -- python $R/tools/pmc_target.py moe1k > $R/$OUT/sq.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$OUT/mfma -- python $R/tools/pmc_target.py moe1k > $R/$OUT/mfma.log 2>&1 )
find $OUT -name "*kernel_trace.csv" -size +2M -delete
python - <<'PY' | tee gpurun_out/r03_pmc_moe_mid/summary.txt
import csv, glob, collections
for tag in ("fetch", "write", "sq", "mfma"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r03_pmc_moe_mid/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n = row.get("Kernel_Name", "")
            if "gemm8p_kernel" in n:
                acc[n.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kn, d in acc.items():
        print(tag, kn, {k: (round(sum(v) / len(v)), len(v)) for k, v in d.items()})
for f in glob.glob("gpurun_out/r03_pmc_moe_mid/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm8p" in row["Name"] or "moe_" in row["Name"]: print(row["Name"].split("(")[0][-70:], "avg ns", row["AverageNs"], "calls", row["Calls"])
print(open("gpurun_out/r03_pmc_moe_mid/trace.log").read().strip().splitlines()[-1])
PY
