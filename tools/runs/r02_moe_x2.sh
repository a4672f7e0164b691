for f in 0 3; do echo "== SEMIPD_MOE_TILED_FORM=$f"; SEMIPD_MOE_TILED_FORM=$f python tools/kbench_moe_stages.py; done
SEMIPD_MOE_TILED_FORM=3 python -m pytest tests/test_gpu_ops.py -q -x -k "moe" 2>&1 | tail -2
