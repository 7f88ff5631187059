mkdir -p gpurun_out/s4
for i in 1 2 3 4 5 6; do
  python -m pytest tests/test_autograph_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q --tb=short -p no:warnings 2>&1 | tail -30 > gpurun_out/s4/run_$i.txt
  tail -1 gpurun_out/s4/run_$i.txt
done
grep -l "failed" gpurun_out/s4/run_*.txt | head -3 | xargs -r -n1 cat
