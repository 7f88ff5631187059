mkdir -p gpurun_out/s11
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_autograph_gpu.py tests/test_two_process_gpu.py tests/test_fused_mlp_gpu.py -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -8 > gpurun_out/s11/test.txt
tail -4 gpurun_out/s11/test.txt
