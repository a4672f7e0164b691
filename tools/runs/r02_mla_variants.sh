#!/bin/bash
OUT=gpurun_out/r02_mla; mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DMLS_TRACE"
for v in "" "-DMLS_NO_RESC" "-DMLS_NO_EXP_IN_PV" "-DMLS_ONE_TR" "-DMLS_NO_RESC -DMLS_ONE_TR -DMLS_NO_EXP_IN_PV"; do
  (cd semi-pd_amd/csrc && touch mla_decode_shared.hip && make CXXFLAGS="$BASE $v" > /dev/null 2>&1)
  echo "== variant [$v]"
  timeout 120 python tools/dbg_mls_trace.py 2>&1 | tail -9
done 2>&1 | tee $OUT/variants.txt
