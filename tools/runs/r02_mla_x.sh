#!/bin/bash
OUT=gpurun_out/r02_mla; mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result"
for v in "" "-DMLS_X_NOBAR" "-DMLS_X_NOEXP" "-DMLS_X_NODMA" "-DMLS_X_NODMA -DMLS_X_NOBAR" "-DMLS_X_NODMA -DMLS_X_NOBAR -DMLS_X_NOEXP"; do
  (cd semi-pd_amd/csrc && touch mla_decode_shared.hip && make CXXFLAGS="$BASE $v" > /dev/null 2>&1)
  echo "== variant [$v]"
  timeout 120 python tools/dbg_mla_l2.py 2>&1 | tail -3
done 2>&1 | tee $OUT/variants_x.txt
