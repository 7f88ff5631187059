"""The decoder's channel-mixing products (PSPUpsample split form: q = p Wr, dp = dq Wr^T, dWr = p^T dq) as the framework issues
them, under the BLAS back ends torch can pick: default (hipBLASLt heuristic), rocBLAS, TunableOp."""
import os, sys, time
import torch

dev = torch.device("cuda:0")
shapes = [("up_1", 32 * 24 * 24, 1024, 9 * 256), ("up_2", 32 * 48 * 48, 256, 9 * 64), ("up_3", 32 * 96 * 96, 64, 9 * 64)]


def bench(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def run(tag):
    tot = 0.0
    for name, m, k, n in shapes:
        p = torch.randn(m, k, device=dev)
        wr = torch.randn(k, n, device=dev)
        dq = torch.randn(m, n, device=dev)
        t1 = bench(lambda: p @ wr)
        t2 = bench(lambda: dq @ wr.t())
        t3 = bench(lambda: p.t() @ dq)
        gf = 2.0 * m * k * n / 1e9
        print(f"{tag:10s} {name}: fwd {t1:7.1f} us ({gf / t1 * 1e3:6.1f} TF)  dgrad {t2:7.1f} us ({gf / t2 * 1e3:6.1f} TF)  wgrad {t3:7.1f} us ({gf / t3 * 1e3:6.1f} TF)")
        tot += t1 + t2 + t3
    print(f"{tag:10s} total {tot / 1e3:.2f} ms")


mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")
elif mode == "hipblaslt":
    torch.backends.cuda.preferred_blas_library("cublaslt")
elif mode == "tunable":
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.set_max_tuning_duration(300)
    torch.cuda.tunable.set_filename("/tmp/tunableop_decoder.csv")
print("preferred:", torch.backends.cuda.preferred_blas_library())
run(mode)
