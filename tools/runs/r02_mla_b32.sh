OUT=gpurun_out/r02_mla_b32; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o t -- python $R/tools/pmc_target.py mla128_b32 > $R/$OUT/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r02_mla_b32/trace/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "mla_decode" in r["Kernel_Name"] or "stage2" in r["Kernel_Name"]]
    for r in rows:
        print(r["Kernel_Name"][:48], r["Grid_Size_X"] if "Grid_Size_X" in r else "", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
find $OUT -name "*kernel_trace.csv" -delete
