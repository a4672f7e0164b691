#!/bin/bash
OUT=gpurun_out/r02_pmc_mla; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -o t -- python $R/tools/pmc_target.py mla128 > $R/$OUT/trace.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/$OUT/pmc_$tag -o p -- python $R/tools/pmc_target.py mla128 > $R/$OUT/pmc_$tag.log 2>&1
done
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python - <<'PY'
import csv, glob, collections
for kn in ("mla_decode_shared_kernel",):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r02_pmc_mla/pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kn in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(kn, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
for f in glob.glob("gpurun_out/r02_pmc_mla/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mla" in row["Name"] or "stage2" in row["Name"] or "merge" in row["Name"]:
            print("  ", row["Name"][:70], "avg ns", row["AverageNs"], "calls", row["Calls"])
PY
