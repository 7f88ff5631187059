"""Level-1 furthest-point sampling (512 of 1024, the encoder's 298-us chain) with 1 / 2 / 4 / 8 waves per cloud, plain and with the
tie tracking the chained form needs; B = 32 (training) and B = 1, 4 (the reference test loop's batches)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import shell_cloud
from istnet_amd import _native
from istnet_amd.pointnet2 import _ext
lib = _native.lib()
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n, m in ((1024, 512), (2048, 512)):
    for b in (32, 4, 1):
        xyz = shell_cloud(b, n, seed=0, device=dev)
        ref = None
        row = []
        for waves in (1, 2, 4, 8):
            if waves == 1:
                if n > 1024:
                    row.append("1 wave: n/a")
                    continue
                lib.istnet_pn2_set_tuning(0, 1025)
            else:
                lib.istnet_pn2_set_tuning(0, 1); lib.istnet_pn2_set_tuning(3, waves)
            idx, _ = _ext.furthest_point_sampling_gather(xyz, m)
            ref = idx if ref is None else ref
            assert torch.equal(idx, ref)
            t_plain = timeit(lambda: _ext.furthest_point_sampling_gather(xyz, m))
            t_track = timeit(lambda: _ext.furthest_point_sampling_chain(xyz, m, None, m // 2))
            row.append(f"{waves} wave{'s' if waves > 1 else ''}: {t_plain:6.1f} / {t_track:6.1f}")
        lib.istnet_pn2_set_tuning(0, 1024); lib.istnet_pn2_set_tuning(3, 4)
        print(f"n={n} m={m} B={b:2d}  us plain / tracked   " + "   ".join(row))
