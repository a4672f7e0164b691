#!/bin/bash
# round 5, call 11: the decode step next to a process whose stream is HELD by a gate with work queued behind it (what a prefill instance
# looks like during a hold): empty kernels / GEMMs behind the gate, NULL / masked stream
OUT=gpurun_out/r05_s11; mkdir -p $OUT
step() { timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 400 2>&1 | grep "ms per decode" | cut -c1-90; }
echo "alone: $(step)" | tee $OUT/gate_neighbour_backlog.txt
for cfg in "masked 300 noop" "masked 40 gemm" "null 40 gemm" "masked 0 noop"; do
  timeout 120 python tools/gate_neighbour_probe.py spinner $(echo $cfg | cut -d" " -f1) 45 $(echo $cfg | cut -d" " -f2) $(echo $cfg | cut -d" " -f3) > $OUT/spinner.txt 2>&1 &
  SP=$!
  sleep 2
  echo "next to [$cfg]: $(step)" | tee -a $OUT/gate_neighbour_backlog.txt
  wait $SP
  grep spinner $OUT/spinner.txt | cut -c1-200 | tee -a $OUT/gate_neighbour_backlog.txt
done
