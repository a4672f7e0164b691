#!/bin/bash
# round 6, call 14: the narrow streaming workgroups by default (weights of <= 2^24 elements at <= 32 rows): decode step alone, 8B and the
# 70B TP = 8 rank's shapes, against SEMIPD_SL_NW=8 (the eight-wave form everywhere)
OUT=gpurun_out/r06_s14; mkdir -p $OUT
for M in llama3-8b llama3-70b-tp8-rank; do
  for B in 32 16; do
    for NW in 8 0 8 0; do
      SEMIPD_SL_NW=$NW timeout 300 python tools/decode_step_bench.py --model $M --batch $B --ctx 1100 2>&1 | grep "ms per decode step" | cut -c1-75 | sed "s/^/SL_NW=$NW /" | tee -a $OUT/steps.txt
    done
  done
done
