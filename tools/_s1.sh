mkdir -p gpurun_out/r5b
python tools/exp/autograph_debug.py 2>&1 | grep -v amdgpu.ids
python -X faulthandler -m pytest tests/test_autograph_gpu.py tests/test_optim.py tests/test_conv_gpu.py -x -q > gpurun_out/r5b/test.txt 2>&1
tail -30 gpurun_out/r5b/test.txt
