mkdir -p gpurun_out/s8
python -m pytest tests -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -15 > gpurun_out/s8/test.txt
tail -5 gpurun_out/s8/test.txt
tools/exp/ab_vals.sh ISTNET_EXP_INPLACE "0 1" > gpurun_out/s8/ab_inplace.txt 2>&1
cat gpurun_out/s8/ab_inplace.txt
