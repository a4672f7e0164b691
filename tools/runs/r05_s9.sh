#!/bin/bash
# round 5, call 9: why does a holding gate on a CU-masked stream slow the decode instance down?  the decode step alone, and next to a
# process that does nothing but hold gate kernels on its NULL stream / a created stream / a CU-masked stream
OUT=gpurun_out/r05_s9; mkdir -p $OUT
step() { timeout 300 python tools/decode_step_bench.py --model llama3-8b --batch 32 --ctx 1100 --steps 400 2>&1 | grep "ms per decode" | cut -c1-90; }
echo "alone: $(step)" | tee $OUT/gate_neighbour.txt
for kind in null created masked; do
  timeout 120 python tools/gate_neighbour_probe.py spinner $kind 45 > $OUT/spinner_$kind.txt 2>&1 &
  SP=$!
  sleep 2
  echo "next to a gate held on the $kind stream: $(step)" | tee -a $OUT/gate_neighbour.txt
  wait $SP
  grep spinner $OUT/spinner_$kind.txt | cut -c1-200 | tee -a $OUT/gate_neighbour.txt
done
