#!/bin/bash
OUT=gpurun_out/r02_moe_x; mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result"
for v in "" "-DMTG_X_NOBAR" "-DMTG_X_NODMA" "-DMTG_X_NODMA -DMTG_X_NOBAR" "-DMTG_X_NOLDS" "-DMTG_X_NODMA -DMTG_X_NOBAR -DMTG_X_NOLDS"; do
  (cd semi-pd_amd/csrc && touch moe_tiled_gemm.hip && make CXXFLAGS="$BASE $v" > /dev/null 2>&1)
  echo "== variant [$v]"
  timeout 120 python tools/kbench_moe_stages.py 2>&1 | grep "T=8192"
done 2>&1 | tee $OUT/variants.txt
