"""A/B of a module-level switch inside the bench step:  python tools/exp/ab_modflag.py pkg.module NAME=0|1 [bench args...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import istnet_amd
mod = importlib.import_module(sys.argv[1])
name, val = sys.argv[2].split("=")
assert hasattr(mod, name), name
setattr(mod, name, bool(int(val)))
sys.argv = ["bench.py"] + sys.argv[3:]
import bench
bench.main()
