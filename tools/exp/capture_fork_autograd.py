"""Reproducer for the hipStreamEndCapture SIGSEGV of DESIGN.md 7 (rounds 2-3): which stream topology kills capture_end?

Each variant runs in its own process (a SIGSEGV must not take the others down) and captures ONE training step of a toy
two-branch model into a HIP graph:
  plain     branch B on a forked stream, joined before the loss; autograd replays its backward there (torch only)
  cb_cur    + a third stream W forked from the stream of B's backward node, joined by an end-of-backward callback
              (Engine.queue_callback) into the CURRENT stream of the callback (the ambient stream of backward())
  cb_first  + the same, but joined into the stream recorded by the FIRST backward node that ran (fused_mlp._Deferred.mains
              before round 4): when that node ran on the forked stream, W's join lands on a stream the engine has
              already joined -> the capture ends with an unjoined fork
    python tools/exp/capture_fork_autograd.py            # runs all variants, prints one line each
"""
import subprocess
import sys

import torch

VARIANTS = ("plain", "cb_cur", "cb_first")


def run(variant):
    dev = torch.device("cuda:0")
    wa, wb = (torch.randn(256, 256, device=dev, requires_grad=True) for _ in range(2))
    x = torch.randn(64, 256, device=dev)
    fork, wside = torch.cuda.Stream(), torch.cuda.Stream()
    state = {}

    class Tap(torch.autograd.Function):          # stands for a fused node that defers work to stream W in backward
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            cur = torch.cuda.current_stream()
            state.setdefault("first", cur)
            if variant != "plain":
                wside.wait_stream(cur)
                with torch.cuda.stream(wside):
                    state["junk"] = g * 2.0          # "deferred weight gradient"
                if not state.get("armed"):
                    state["armed"] = True

                    def join():
                        target = torch.cuda.current_stream() if variant == "cb_cur" else state["first"]
                        target.wait_stream(wside)
                        state["armed"] = False
                    torch.autograd.Variable._execution_engine.queue_callback(join)
            return g

    def step():
        cur = torch.cuda.current_stream()
        a = (x @ wa).relu()
        fork.wait_stream(cur)
        with torch.cuda.stream(fork):
            b = Tap.apply((x @ wb).relu())           # created last: its backward runs FIRST, and on `fork`
        cur.wait_stream(fork)
        state.pop("first", None)
        (a.sum() + b.sum()).backward()

    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    wa.grad = wb.grad = None
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    print("captured and replayed")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in VARIANTS:
            p = subprocess.run([sys.executable, "-X", "faulthandler", __file__, v], capture_output=True, text=True)
            tail = (p.stdout.strip().splitlines() or [""])[-1] if p.returncode == 0 else \
                " | ".join(l for l in p.stderr.strip().splitlines() if "Error" in l or "Fatal" in l or "capture_end" in l)[-300:]
            print(f"{v:9s} rc={p.returncode:4d}  {tail}")
