"""Furthest point sampling of the encoder's four levels: every level scanned vs the chained form (children take the
prefix of their parent's picks when the parent reported no arg-max tie).  HIP-event timing, B=32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import shell_cloud
from istnet_amd.pointnet2 import _ext

dev = torch.device("cuda:0")
LEVELS = (512, 256, 128, 64)


def run_plain(xyz):
    cur = xyz
    for m in LEVELS:
        _, cur = _ext.furthest_point_sampling_gather(cur, m)
    return cur


def run_chain(xyz, track=True, track_from=0):
    """track_from: first level (0-based) whose run reports ties -- earlier levels scan without the bookkeeping and their
    children scan too (modules.FPS_CHAIN_MAX_TRACKED_N: above 1 024 points the 4-wave kernel pays more for the tracking
    than the level below saves)."""
    cur, tie = xyz, None
    for li, m in enumerate(LEVELS):
        nxt = LEVELS[li + 1] if li + 1 < len(LEVELS) else 0
        rounds = min(nxt, m) if (track and li >= track_from) else 0
        _, cur, tie = _ext.furthest_point_sampling_chain(cur, m, tie_in=tie, track_rounds=rounds)
    return cur, tie


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


for n, b in ((1024, 32), (2048, 64)):
    xyz = shell_cloud(b, n, seed=0, device=dev)
    a = run_plain(xyz)
    c, _ = run_chain(xyz)
    assert torch.equal(a, c)
    _, _, tie = _ext.furthest_point_sampling_chain(xyz, LEVELS[0], None, LEVELS[1])     # level 1's own report
    c1, _ = run_chain(xyz, track_from=1)
    assert torch.equal(a, c1)
    print(f"n={n} b={b}: every level scanned {timeit(lambda: run_plain(xyz)):.1f} us, chained {timeit(lambda: run_chain(xyz)):.1f} us, "
          f"chained from level 2 on (level 1 untracked) {timeit(lambda: run_chain(xyz, track_from=1)):.1f} us, "
          f"level 1 alone (the encoder's {LEVELS[0]} of {n}) plain {timeit(lambda: _ext.furthest_point_sampling_gather(xyz, LEVELS[0])):.1f} us / tracking "
          f"{LEVELS[1]} rounds {timeit(lambda: _ext.furthest_point_sampling_chain(xyz, LEVELS[0], None, LEVELS[1])):.1f} us; "
          f"(a run of {n // 2} of {n}: plain {timeit(lambda: _ext.furthest_point_sampling_gather(xyz, n // 2)):.1f} us / tracking {n // 4} rounds "
          f"{timeit(lambda: _ext.furthest_point_sampling_chain(xyz, n // 2, None, n // 4)):.1f} us); level-1 clouds with an arg-max tie before round {LEVELS[1]}: "
          f"{int((tie.cpu() < LEVELS[1]).sum())} of {b}")
