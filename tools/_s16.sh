for rep in 1 2; do
for v in 256 192 320 384 512; do
python - <<PY 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FP_BWD_MID_WORKGROUPS=$v', round(d['ms_per_step'],4), d.get('windows_ms_per_step'))"
import sys
sys.argv = ['bench.py', '--no-roofline', '--no-cpu-baseline', '--no-eager-leg', '--no-unpipelined', '--steps', '50', '--warmup', '10', '--windows', '3']
import istnet_amd
from istnet_amd.pointnet2 import fused_mlp as f
f.FP_BWD_MID_WORKGROUPS = $v
import bench
bench.main()
PY
done; done
