"""Which shapes / switches make the backward-only capture of graphed.AutoGraph die?  One child process per case."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def child(b, n, sw):
    import torch, warnings
    import bench
    from istnet_amd import graphed
    from istnet_amd.pointnet2 import fused_mlp
    import istnet_amd.modules as M
    for kv in sw.split(","):
        if kv:
            k, v = kv.split("=")
            mod = M if hasattr(M, k) else fused_mlp
            setattr(mod, k, v == "1")
    model = bench.make_model(torch.device("cuda:0"))
    pts = bench.shell_cloud(b, n, 0, "cuda:0")
    for it in range(4):
        model.zero_grad()
        out = model(pts)
        out.square().mean().backward()
    torch.cuda.synchronize()
    print("OK", b, n, sw, graphed.STATS, fused_mlp.FALLBACKS, flush=True)

if len(sys.argv) > 1:
    child(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "")
else:
    cases = [(32, 1024, ""), (4, 1024, ""), (4, 512, ""), (2, 256, ""), (3, 512, ""),
             (4, 512, "USE_SCALE_STREAMS_BWD=0"), (4, 512, "USE_DEFERRED_WGRAD=0"), (4, 512, "USE_FP_SKIP_STREAM=0"),
             (4, 512, "USE_GEOMETRY_STREAM=0"), (4, 512, "USE_SCALE_STREAMS=0,USE_SCALE_STREAMS_BWD=0")]
    for b, n, sw in cases:
        r = subprocess.run([sys.executable, __file__, str(b), str(n), sw], capture_output=True, text=True)
        last = [l for l in r.stdout.splitlines() if l.startswith("OK")]
        print(f"B={b} n={n} {sw or '-':40s} rc={r.returncode}", last[-1] if last else r.stderr.strip().splitlines()[-1:] , flush=True)
