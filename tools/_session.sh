mkdir -p gpurun_out/r4j
python -m pytest tests/test_pw_kernels_gpu.py tests/test_fused_mlp_gpu.py -m gpu -q 2>&1 | tail -2
bash tools/ab.sh 3 2>&1 | tee gpurun_out/r4j/ab2.txt
bash tools/ab.sh 2 --workload istnet --steps 20 2>&1 | tee gpurun_out/r4j/ab_istnet.txt
