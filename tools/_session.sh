python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py -m gpu -q 2>&1 | tail -1
bash tools/ab.sh 3 2>&1
