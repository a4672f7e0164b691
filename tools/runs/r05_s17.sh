#!/bin/bash
# round 5, call 17: PMC passes (FETCH_SIZE | WRITE_SIZE in separate runs, no trace domains beside --kernel-trace) of the two dominant decode kernels
# on the whole chip (the decode instance of the default policy is unmasked): the streaming GEMM (gate_up + SiLU, 32 rows) and the decode attention
# (B = 32, ctx 1100, the engine's split count)
OUT=gpurun_out/r05_s17; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tgt in stream decode32; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$tgt_$ctr
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tgt}_$ctr -- python $R/tools/pmc_target.py $tgt > $R/$OUT/${tgt}_$ctr.log 2>&1
    echo "== $tgt $ctr: $(grep -h algorithmic $R/$OUT/${tgt}_$ctr.log | tail -1)"
    python $R/tools/pmc_summary.py /tmp/pmc_${tgt}_$ctr stream_gemm_glds decode_mfma decode_stage2 splitk
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_${tgt}_stats -- python $R/tools/pmc_target.py $tgt > /dev/null 2>&1
  python $R/tools/stats_top.py $(find /tmp/pmc_${tgt}_stats -name "*kernel_stats.csv" | head -1) | grep -E "stream_gemm|decode_mfma|decode_stage2|splitk" | cut -c1-150
done 2>&1 | tee $R/$OUT/pmc_summary.txt
