mkdir -p gpurun_out/s2
python -m pytest tests/test_pipeline_gpu.py -m gpu -q -k "bn_momentum_schedule" 2>&1 | grep -v Warning | tail -40 > gpurun_out/s2/fail.txt
python -m pytest tests -m gpu -q --deselect tests/test_pipeline_gpu.py::test_captured_step_follows_bn_momentum_schedule 2>&1 | tail -8 > gpurun_out/s2/test_rest.txt
tools/exp/ab_vals.sh ISTNET_PW_TUNE "20:64 23:128 20:128" > gpurun_out/s2/ab_wgrad_tiles.txt 2>&1
cat gpurun_out/s2/fail.txt gpurun_out/s2/test_rest.txt gpurun_out/s2/ab_wgrad_tiles.txt
