"""Bisect the capture_end SIGSEGV of ISTNET_WORLD_EXTRACTOR_STREAM=1: which ingredient of the full-model
step kills hipStreamEndCapture when the world-space encoder runs on a stream of its own?  Toy sizes, one process per
variant.    python tools/exp/world_stream_bisect.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {
    "baseline (no world stream)": {"ISTNET_WORLD_EXTRACTOR_STREAM": "0"},
    "world stream": {},
    "world stream, forward only": {"BISECT_FWD_ONLY": "1"},
    "world stream, no geometry stream": {"ISTNET_GEOMETRY_STREAM": "0"},
    "world stream, heads wgrad not deferred": {"BISECT_NO_HEADS_DEFER": "1"},
    "world stream, no rgb stream": {"BISECT_NO_RGB_STREAM": "1"},
    "world stream, no FPS chain": {"BISECT_NO_FPS_CHAIN": "1"},
}


def run():
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from istnet_amd import ist_net, modules
    from istnet_amd.ist_net import point_branch_side_streams
    from istnet_amd.optim import FlatAdam, layout_hints
    from istnet_amd.pointnet2 import fused_mlp
    point_branch_side_streams(False)
    if os.environ.get("BISECT_NO_HEADS_DEFER"):
        fused_mlp.USE_DEFERRED_WGRAD_HEADS = False
    if os.environ.get("BISECT_NO_RGB_STREAM"):
        ist_net.USE_RGB_STREAM = False
    if os.environ.get("BISECT_NO_FPS_CHAIN"):
        modules.USE_FPS_CHAIN = False
    dev = torch.device("cuda:0")
    model = bench.make_istnet(dev, seed=0)
    batch = bench.istnet_batch(4, 256, seed=0, device=dev, hw=64)
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
    if os.environ.get("BISECT_FWD_ONLY"):
        def fwd_bwd():
            return model(batch)["pred_rotation"].sum()
    else:
        fwd_bwd = bench.make_istnet_fwd_bwd(model, batch)
    step = bench.make_graphed_step(fwd_bwd, opt, 1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print("captured and replayed")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        for name, env in VARIANTS.items():
            e = dict(os.environ, ISTNET_WORLD_EXTRACTOR_STREAM="1")
            e.update(env)
            p = subprocess.run([sys.executable, "-X", "faulthandler", __file__, "child"], capture_output=True, text=True, env=e)
            ok = p.returncode == 0 and "captured and replayed" in p.stdout
            why = "" if ok else " | ".join(l.strip() for l in p.stderr.splitlines() if "Fatal" in l or "Error" in l)[-160:]
            print(f"{name:42s} rc={p.returncode:4d} {'OK' if ok else why}")
