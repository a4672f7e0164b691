python tools/dbg_dsfp8.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_deepseek.py -x -q 2>&1 | tail -15
