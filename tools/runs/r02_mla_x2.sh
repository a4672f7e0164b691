#!/bin/bash
OUT=gpurun_out/r02_mla; mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result"
for v in ""; do
  (cd semi-pd_amd/csrc && touch mla_decode_shared.hip && make CXXFLAGS="$BASE $v" > /dev/null 2>&1)
  echo "== variant [$v]"
  timeout 120 python tools/dbg_mla_l2.py 2>&1 | tail -3
  timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "mla_decode_shared" 2>&1 | tail -2
  python tools/kbench_mla_quick.py 2>&1 | grep "H="
done 2>&1 | tee $OUT/variants_x2.txt
