mkdir -p gpurun_out/s3
python -m pytest tests -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -60 > gpurun_out/s3/test_full.txt
python tools/exp/find_copies2.py > gpurun_out/s3/find_copies2.txt 2>&1
tools/exp/ab_vals.sh ISTNET_EXP_TAIL "0 1" > gpurun_out/s3/ab_tail.txt 2>&1
cat gpurun_out/s3/test_full.txt; tail -40 gpurun_out/s3/find_copies2.txt; cat gpurun_out/s3/ab_tail.txt
