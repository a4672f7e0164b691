#!/bin/bash
# round 2: where does the extend-attention kernel spend its cycles? (PMC passes, counters only + kernel trace)
OUT=gpurun_out/r02_pmc_skv; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/$OUT/counters_list.txt 2>&1
for which in extend8k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_$which -o t -- python $R/tools/pmc_target.py $which > $R/$OUT/trace_$which.log 2>&1
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" ; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/$OUT/pmc_${which}_$tag -o p -- python $R/tools/pmc_target.py $which > $R/$OUT/pmc_${which}_$tag.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for which in ("extend8k",):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02_pmc_skv/pmc_{which}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "extend_attn" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(which, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
    for f in glob.glob(f"gpurun_out/r02_pmc_skv/trace_{which}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "extend_attn" in row["Name"]:
                print("  avg ns", row["AverageNs"], "calls", row["Calls"])
PY
find gpurun_out/r02_pmc_skv -name "*kernel_trace.csv" -delete; find gpurun_out/r02_pmc_skv -name "*.db" -delete
