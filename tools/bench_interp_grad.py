"""istnet_interp_grad_csr_dy in isolation on the encoder's four feature-propagation levels (B = 32): the gradient of the
interpolated features with dY formed per element, gathered from global memory (key 22 = 0) or from LDS-staged rows (default)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istnet_amd, bench
from istnet_amd import _native
from istnet_amd.pointnet2 import _ext as E
lib = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = 32
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
for name, n, m, c in [("FP1", 1024, 512, 128), ("FP2", 512, 256, 256), ("FP3", 256, 128, 256), ("FP4", 128, 64, 512)]:
    unknown = bench.shell_cloud(B, n, seed=n).to(dev)
    known = E.furthest_point_sampling_gather(unknown, m)[1]
    idx, weight = E.three_nn_weights(unknown, known)
    off, ent = E.interp_csr(idx, m)
    y = torch.randn(B, c, n, generator=g).to(dev); d = torch.randn(B, c, n, generator=g).to(dev)
    bn = torch.stack([torch.ones(c), torch.zeros(c), torch.zeros(c), torch.ones(c)]).contiguous().to(dev)
    bw = torch.stack([torch.ones(c), torch.zeros(c), torch.zeros(c)]).contiguous().to(dev)
    outs, times = [], []
    for flag in (0, 1):
        lib.istnet_pw_set_tuning(22, flag)
        out = torch.empty(B, c, m, device=dev)
        f = lambda: lib.istnet_interp_grad_csr_dy(B, c, n, m, y.data_ptr(), d.data_ptr(), bn.data_ptr(), bw.data_ptr(),
                                                  weight.data_ptr(), off.data_ptr(), ent.data_ptr(), out.data_ptr(), st)
        assert f() == 0
        times.append(timeit(f)); outs.append(out)
    nbytes = 4.0 * B * (2 * c * n + c * m + 7 * n)
    print(f"{name}: n {n:5d} m {m:4d} C {c:4d}  global gather {times[0]:6.1f} us {nbytes / times[0] / 1e3:6.0f} GB/s   "
          f"LDS rows {times[1]:6.1f} us {nbytes / times[1] / 1e3:6.0f} GB/s   bit-identical: {torch.equal(outs[0], outs[1])}")
