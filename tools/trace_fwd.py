"""Debug: per-workgroup phase timestamps of pw_fwd_kernel (trace build of the library, -DISTNET_TRACE).
    python tools/trace_fwd.py            (GPU box; builds gpurun_out/libtrace.so with hipcc)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
so = os.path.join(ROOT, "gpurun_out", "libtrace.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
src = [os.path.join(ROOT, "ist-net_amd", "csrc", f) for f in ("pw_mlp.hip", "pn2_index_ops.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                       "-shared", "-fvisibility=hidden", "-DISTNET_TRACE", "-I", os.path.join(ROOT, "include"), *src, "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = 32
P_ = ctypes.c_void_p
for (cin, cout, P) in [(128, 128, 1024), (128, 256, 2048), (64, 128, 4096), (16, 32, 16384), (256, 128, 1024)]:
    x = torch.randn(B, cin, P, device=dev); w = torch.randn(cout, cin, device=dev) * 0.1
    y = torch.empty(B, cout, P, device=dev); nt = lib.istnet_pw_stat_tiles(B, cout, P); part = torch.empty(2, cout, nt, device=dev)
    def run():
        lib.istnet_pw_forward(B, cin, cout, P, P_(x.data_ptr()), P_(w.data_ptr()), None, None, P_(y.data_ptr()),
                              P_(part[0].data_ptr()), P_(part[1].data_ptr()), P_(st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    tr = np.zeros((8, 8192), dtype=np.uint64)
    lib.istnet_pw_trace_read(tr.ctypes.data_as(P_))
    cfg = lib.istnet_pw_tile_cfg(B, cout, P)
    nwg = nt * ((cout + cfg // 1000 - 1) // (cfg // 1000))
    n = min(nwg, 8192)
    t = tr[:4, :n].astype(np.int64); t0 = t[0].min()
    cyc = tr[4:, :n].astype(np.int64)
    mhz = (cyc[2] - cyc[1]) / np.maximum((t[2] - t[1]) / 100.0, 1e-3)
    print(f"  shader clock during main loop: median {np.median(mhz):.0f} MHz; main loop cycles median {np.median(cyc[2]-cyc[1]):.0f}")
    us = (t - t0) / 100.0
    print(f"cin {cin} cout {cout} P {P} cfg {cfg} wgs {nwg}  event {e0.elapsed_time(e1)*1e3:.1f} us")
    print(f"  wg start: min {us[0].min():.1f} median {np.median(us[0]):.1f} p90 {np.percentile(us[0],90):.1f} max {us[0].max():.1f}")
    print(f"  prologue (start->first chunk staged): median {np.median(us[1]-us[0]):.2f} max {(us[1]-us[0]).max():.2f}")
    print(f"  main loop: median {np.median(us[2]-us[1]):.2f} max {(us[2]-us[1]).max():.2f}")
    print(f"  epilogue: median {np.median(us[3]-us[2]):.2f} max {(us[3]-us[2]).max():.2f}")
    print(f"  wg end: median {np.median(us[3]):.1f} max {us[3].max():.1f}")
