"""A/B of a fused_mlp module switch inside the bench step:  python tools/exp/ab_flag.py NAME=0|1 [bench args...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import istnet_amd
from istnet_amd.pointnet2 import fused_mlp
name, val = sys.argv[1].split("=")
assert hasattr(fused_mlp, name), name
setattr(fused_mlp, name, bool(int(val)))
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
