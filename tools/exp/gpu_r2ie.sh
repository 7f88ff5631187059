#!/bin/bash
python -m pytest tests/test_pipeline_gpu.py tests/test_fused_mlp_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -2
ARGS="'bench.py','--no-roofline','--no-cpu-baseline','--no-unpipelined','--steps','50','--warmup','10'"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "interpolate launch  : "; python -c "import sys; sys.argv=[$ARGS]; import istnet_amd.pointnet2.fused_mlp as f; f.USE_INTERP_IN_EPILOGUE=False; import bench; bench.main()" 2>&1 | ms
  echo -n "interp. in epilogue : "; python -c "import sys; sys.argv=[$ARGS]; import bench; bench.main()" 2>&1 | ms
done
