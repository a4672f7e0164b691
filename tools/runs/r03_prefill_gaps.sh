#!/bin/bash
# round 3: idle time between the kernels of a 1024-token prefill batch (eager launches), default masks, 8 req/s
OUT=gpurun_out/r03_prefill_gaps; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_prof -- python $R/bench.py --no-cpu-baseline --no-static-split-wave --no-saturation-wave --rate-sweep "" --request-rate 8 --num-requests 48 --steps 1 --warmup 1 > $R/$OUT/bench.json 2> $R/$OUT/bench.err )
python tools/prefill_gaps.py /tmp/gaps_prof | tee $OUT/gaps.txt
