"""Timeline of the LAST step in a rocprofv3 --kernel-trace CSV (graph replays included): per hardware queue, busy fraction and
dominant kernel per 0.5 ms; and the per-kernel totals of that step.   python tools/trace_timeline.py <kernel_trace.csv> [step_ms]"""
import csv, sys, collections
path = sys.argv[1]
step_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 34.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
end = max(r[1] for r in rows)
# last step: kernels starting within the last step_ms (the adam kernel ends a step)
adam = [r for r in rows if "adam_step_kernel" in r[3]]
t_hi = adam[-1][1]
t_lo = adam[-2][1] if len(adam) > 1 else t_hi - int(step_ms * 1e6)
step = [r for r in rows if t_lo < r[0] <= t_hi]


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    return n.split("(")[0][:46]


print(f"# last step: {(t_hi - t_lo) / 1e6:.2f} ms, {len(step)} dispatches on queues {sorted({r[2] for r in step})}")
agg = collections.defaultdict(lambda: [0, 0])
for a, b, q, n in step:
    agg[short(n)][0] += 1
    agg[short(n)][1] += b - a
print("   total_us  calls  kernel")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t / 1e3:10.1f} {c:6d}  {n}")
queues = sorted({r[2] for r in step})
nb = int((t_hi - t_lo) / 5e5) + 1
print("# per 0.5 ms: per queue busy fraction and dominant kernel")
for bi in range(nb):
    lo, hi = t_lo + bi * 500000, t_lo + (bi + 1) * 500000
    cells = []
    for q in queues:
        busy, by = 0, {}
        for a, b, qq, n in step:
            if qq != q or b <= lo or a >= hi:
                continue
            d = min(b, hi) - max(a, lo)
            busy += d
            by[short(n)] = by.get(short(n), 0) + d
        top = max(by.items(), key=lambda kv: kv[1])[0] if by else ""
        cells.append(f"{busy / 5e5:4.2f} {top[:30]:30s}")
    print(f"{bi * 0.5:5.1f} | " + " | ".join(cells))
