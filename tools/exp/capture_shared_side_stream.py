"""hipStreamEndCapture SIGSEGV, minimal form (found by tools/exp/world_stream_bisect.py): ONE side stream G that serves two
different streams of the same capture -- first the capture's origin stream M, later a stream W forked from M -- with
every dependency joined.  A legal DAG; on this stack (ROCm 7.2.0, torch 2.10.0+rocm7.0) capture_end dies inside the
runtime instead of returning a graph or an error.  `separate` gives each consumer its own side stream: works.
    python tools/exp/capture_shared_side_stream.py            # runs both variants in child processes
"""
import subprocess
import sys

import torch


def run(variant):
    x = torch.randn(1 << 20, device="cuda")
    W, G = torch.cuda.Stream(), torch.cuda.Stream()
    G2 = G if variant == "shared" else torch.cuda.Stream()

    def step():
        M = torch.cuda.current_stream()
        G.wait_stream(M)                       # side stream serves the origin stream ...
        with torch.cuda.stream(G):
            a = x * 2
        e1 = torch.cuda.Event(); e1.record(G); M.wait_event(e1)
        b = a + 1
        W.wait_stream(M)                       # ... a second stream is forked from the origin ...
        with torch.cuda.stream(W):
            c = x * 3
            G2.wait_stream(W)                  # ... and is served by the SAME side stream
            with torch.cuda.stream(G2):
                d = c * 2
            e2 = torch.cuda.Event(); e2.record(G2); W.wait_event(e2)
            f = d + 1
        M.wait_stream(W)
        return b + f

    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ref = step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    g.replay(); torch.cuda.synchronize()
    print("captured and replayed, max err", float((out - ref).abs().max()))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in ("separate", "shared"):
            p = subprocess.run([sys.executable, "-X", "faulthandler", __file__, v], capture_output=True, text=True)
            msg = p.stdout.strip().splitlines()[-1] if p.returncode == 0 else \
                " | ".join(l.strip() for l in p.stderr.splitlines() if "Fatal" in l or "Error" in l or "capture_end" in l)[-200:]
            print(f"{v:9s} rc={p.returncode:4d}  {msg}")
