for f in 1 2 0; do echo "== tiled form $f"; SEMIPD_MOE_TILED_FORM=$f timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "moe_grouped_gemm_prefill or fused_experts_layer" -x 2>&1 | tail -2; done
for f in 1 2; do echo "== tiled form $f"; SEMIPD_MOE_TILED_FORM=$f timeout 300 python tools/kbench_moe_stages.py 2>&1 | grep "T="; done
