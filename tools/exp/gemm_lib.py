"""dW = a^T g with a huge reduction dim (the weight gradient of rgb_branch._PointMixFn): plain torch.matmul against
the manual split-K through torch.bmm over S slices of the points.  Measured on MI355X (fp32, hipBLASLt):
  m=18432  k=1024 n=2304: plain 1370 us (63 TF/s)   S=16:  659 us (132 TF/s)
  m=73728  k=256  n=576 : plain  666 us (33 TF/s)   S=16:  193 us (113 TF/s)
  m=294912 k=64   n=576 : plain  615 us (35 TF/s)   S=32:  227 us ( 96 TF/s)"""
import torch
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (m, k, n) in ((18432, 1024, 2304), (73728, 256, 576), (18432, 512, 1024), (294912, 64, 576)):
    a = torch.randn(m, k, device=dev)
    g = torch.randn(m, n, device=dev)
    fl = 2.0 * m * k * n
    t = timeit(lambda: a.t() @ g)
    line = f"m={m} k={k} n={n}: plain {t:7.1f} us {fl / t / 1e6:6.1f} TF |"
    ref = a.t() @ g
    for s in (4, 8, 16, 32, 64):
        f = lambda: torch.bmm(a.view(s, m // s, k).transpose(1, 2), g.view(s, m // s, n)).sum(0)
        t = timeit(f)
        err = float((f() - ref).abs().max() / ref.abs().max())
        line += f" S={s}: {t:7.1f} us {fl / t / 1e6:6.1f} TF ({err:.1e})"
    print(line)
