mkdir -p gpurun_out/s10
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_autograph_gpu.py tests/test_two_process_gpu.py -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -15 > gpurun_out/s10/test.txt
tail -4 gpurun_out/s10/test.txt
tools/exp/ab_vals.sh ISTNET_EXP_NNMULTI "0 1" > gpurun_out/s10/ab_nnmulti.txt 2>&1
cat gpurun_out/s10/ab_nnmulti.txt
