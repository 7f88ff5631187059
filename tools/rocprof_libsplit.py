"""Per-library split and a wall-clock timeline of the TIMED steps of a bench.py run traced with
``rocprofv3 --kernel-trace`` (rocpd sqlite .db) -- sees every kernel of the process, MIOpen's backward-data /
backward-weights convolutions included (the torch-profiler table of round 3 dropped them).

    python tools/rocprof_libsplit.py <results.db> <timed steps> <ms per step from the bench line> [bin_ms]

Only kernels that start inside the last ``steps * ms_per_step`` of the trace are counted (the timed region of bench.py:
warm-up and capture come before it), so every figure is per timed step.
"""
import sqlite3
import sys


def library(name):
    n = name
    if "at::native" in n or "at::cuda" in n or n.startswith("void at::") or "elementwise_kernel" in n:
        return "ATen (framework)"
    if "anonymous namespace" in n or n.startswith("istnet") or "pw_" in n[:40] or "fps_" in n[:60]:
        return "libistnet_pn2 (this repo)"
    if n.startswith("igemm_") or "miopen" in n.lower() or "SubTensorOp" in n or "ck::" in n or "_ZN2ck" in n \
            or n.startswith("Op") or "gridwise" in n or "naive_conv" in n or "batchnorm" in n.lower() or "transpose_" in n.lower():
        return "MIOpen"
    if n.startswith("Cijk_") or "rocblas" in n.lower() or "hipblaslt" in n.lower():
        return "hipBLASLt / rocBLAS"
    if "rccl" in n.lower() or "nccl" in n.lower():
        return "RCCL"
    if "copyBuffer" in n or "fillBuffer" in n or "rocclr" in n:
        return "ROCclr copy / fill"
    return "other"


def miopen_kind(name):
    for tag in ("igemm_fwd", "igemm_bwd", "igemm_wrw"):
        if name.startswith(tag):
            return tag
    return name.split("(")[0][:48]


def main():
    path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    bin_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
    sel = "name, start, end" + (f", {qcol}" if qcol else ", 0")
    rows = list(db.execute(f"select {sel} from kernels order by start"))
    t_end = max(r[2] for r in rows)
    t0 = t_end - int(steps * ms * 1e6)
    rows = [r for r in rows if r[1] >= t0]
    span = (t_end - min(r[1] for r in rows)) / 1e6
    print(f"# source: {path}")
    print(f"# timed region: last {steps} steps x {ms:.3f} ms = {steps * ms:.1f} ms of the trace; {len(rows)} dispatches in it "
          f"({len(rows) / steps:.0f} per step), first to last kernel {span:.1f} ms")
    # ---- per library ----
    lib = {}
    for name, s, e, q in rows:
        k = library(name)
        c, t = lib.get(k, (0, 0))
        lib[k] = (c + 1, t + (e - s))
    tot = sum(t for _, t in lib.values())
    print(f"\n## device time per step by library (sum of kernel durations {tot / 1e6 / steps:.2f} ms per step on {ms:.2f} ms of wall "
          f"clock: {tot / 1e6 / steps / ms:.2f} kernels in flight on average)")
    print(f"{'ms/step':>9} {'share':>7} {'launches/step':>14}  library")
    for k, (c, t) in sorted(lib.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e6 / steps:9.3f} {t / tot * 100:6.1f}% {c / steps:14.1f}  {k}")
    # ---- MIOpen by direction ----
    mi = {}
    for name, s, e, q in rows:
        if library(name) == "MIOpen":
            k = miopen_kind(name)
            c, t = mi.get(k, (0, 0))
            mi[k] = (c + 1, t + (e - s))
    print("\n## MIOpen kernels per step")
    for k, (c, t) in sorted(mi.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e6 / steps:9.3f} ms {c / steps:7.1f} launches  {k}")
    # ---- top kernels ----
    kern = {}
    for name, s, e, q in rows:
        c, t = kern.get(name, (0, 0))
        kern[name] = (c + 1, t + (e - s))
    print("\n## kernels above 0.5 % of the device time")
    for name, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][1]):
        if t < 0.005 * tot:
            break
        print(f"{t / 1e6 / steps:9.3f} ms {c / steps:7.1f}/step avg {t / c / 1e3:8.1f} us  [{library(name)[:10]}] {name[:110]}")
    # ---- timeline of ONE step (the last one): busy fraction per queue and union, per bin ----
    s0 = t_end - int(ms * 1e6)
    last = [r for r in rows if r[1] >= s0]
    queues = sorted({r[3] for r in last})
    nb = int(ms / bin_ms) + 1
    print(f"\n## last step, {bin_ms:g} ms bins: busy fraction per hardware queue ({', '.join(str(q) for q in queues)}) | union | "
          "library with most device time in the bin")
    for b in range(nb):
        lo, hi = s0 + int(b * bin_ms * 1e6), s0 + int((b + 1) * bin_ms * 1e6)
        per_q, by_lib, ivs = {q: 0 for q in queues}, {}, []
        for name, s, e, q in last:
            ov = min(e, hi) - max(s, lo)
            if ov > 0:
                per_q[q] += ov
                by_lib[library(name)] = by_lib.get(library(name), 0) + ov
                ivs.append((max(s, lo), min(e, hi)))
        ivs.sort()
        union, cur_s, cur_e = 0, None, None
        for s, e in ivs:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            union += cur_e - cur_s
        width = hi - lo
        top = max(by_lib.items(), key=lambda kv: kv[1])[0] if by_lib else "-"
        print(f"{b * bin_ms:6.1f} ms  " + " ".join(f"{min(per_q[q] / width, 9.99):4.2f}" for q in queues)
              + f" | {union / width:4.2f} | {top}")


if __name__ == "__main__":
    main()
