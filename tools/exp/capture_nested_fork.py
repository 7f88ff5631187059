"""Which fork / join topology makes hipStreamEndCapture segfault (ROCm 7.2.0, torch 2.10.0+rocm7.0)?
M = capture origin, W = stream forked from M, G = stream forked from W (nested fork).  One child process per variant.
    python tools/exp/capture_nested_fork.py
"""
import subprocess
import sys

import torch

VARIANTS = ["nested_join_parent", "nested_join_parent_ws", "nested_join_origin_too", "nested_prefork_origin",
            "nested_join_origin_only", "flat_two_forks", "first_part_only"]


def run(variant):
    x = torch.randn(1 << 20, device="cuda")
    W, G = torch.cuda.Stream(), torch.cuda.Stream()

    def step():
        M = torch.cuda.current_stream()
        if variant == "first_part_only":           # fork + event join on the origin only
            G.wait_stream(M)
            with torch.cuda.stream(G):
                a = x * 2
            e = torch.cuda.Event(); e.record(G); M.wait_event(e)
            return a + 1
        if variant == "flat_two_forks":            # W and G both forked from M, G also waits for W; both joined into M
            W.wait_stream(M); G.wait_stream(M)
            with torch.cuda.stream(W):
                c = x * 3
            G.wait_stream(W)
            with torch.cuda.stream(G):
                d = c * 2
            M.wait_stream(W); M.wait_stream(G)
            return d + 1
        if variant == "nested_prefork_origin":
            G.wait_stream(M)                       # G is a child of the origin BEFORE it serves W
        W.wait_stream(M)
        with torch.cuda.stream(W):
            c = x * 3
            G.wait_stream(W)                       # nested fork
            with torch.cuda.stream(G):
                d = c * 2
            if variant == "nested_join_parent_ws":
                W.wait_stream(G)
            elif variant != "nested_join_origin_only":
                e = torch.cuda.Event(); e.record(G); W.wait_event(e)
            f = c + 1 if variant == "nested_join_origin_only" else d + 1
        M.wait_stream(W)
        if variant in ("nested_join_origin_too", "nested_join_origin_only"):
            M.wait_stream(G)
        return f + d if variant == "nested_join_origin_only" else f

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
        for v in VARIANTS:
            p = subprocess.run([sys.executable, "-X", "faulthandler", __file__, v], capture_output=True, text=True)
            msg = p.stdout.strip().splitlines()[-1] if p.returncode == 0 else \
                " | ".join(l.strip() for l in p.stderr.splitlines() if "Fatal" in l or "Error" in l)[-160:]
            print(f"{v:24s} rc={p.returncode:4d}  {msg}")
