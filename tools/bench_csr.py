"""Inverse-list builds of one encoder pass (6 ball-query + 4 three_nn index tensors, B=32 N=1024 shell clouds):
key-range kernel, one launch for all ten, vs the one-workgroup-per-cloud kernels (ten launches)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CAM_RADII, shell_cloud  # noqa: E402
from istnet_amd import _native  # noqa: E402
from istnet_amd.pointnet2 import _ext  # noqa: E402

dev = "cuda:0"
xyz = shell_cloud(32, 1024, 0, dev)
levels, problems = [xyz], []
for lv, (npoint, rr) in enumerate(zip((512, 256, 128, 64), CAM_RADII)):
    cur = levels[-1]
    _, new = _ext.furthest_point_sampling_gather(cur, npoint)
    for r, s in zip(rr, (16, 32)):
        idx = _ext.ball_query(new, cur, r, s)
        if lv > 0:
            problems.append((idx, cur.shape[1]))
    levels.append(new)
for lv in range(4):
    _, idx = _ext.three_nn(levels[lv], levels[lv + 1])
    problems.append((idx, levels[lv + 1].shape[1]))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("# inverse-list builds of one encoder pass, B=32 N=1024 shell clouds; us per pass (all ten index tensors)")
print(f"key-range kernel, one launch      : {timeit(lambda: _ext.csr_multi(problems)):8.1f} us")
print(f"key-range kernel, ten launches    : {timeit(lambda: [_ext.csr_multi([p]) for p in problems]):8.1f} us")
assert _native.lib().istnet_pn2_set_tuning(2, 1) == 0
print(f"one workgroup per cloud (round 1) : {timeit(lambda: [_ext.csr_multi([p]) for p in problems]):8.1f} us")
_native.lib().istnet_pn2_set_tuning(2, 0)
for (idx, m) in problems:
    e = idx.numel() // 32
    print(f"  E={e:5d} m={m:4d}: {timeit(lambda: _ext.csr_multi([(idx, m)])):7.1f} us alone")
