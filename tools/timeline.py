"""Timeline of the last graph replay in a rocprofv3 --kernel-trace db: per-stream busy time, idle gaps, and the
kernel sequence of the busiest stream with start offsets.
    python tools/timeline.py <db> [dispatches_per_step] [--list]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 401
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
strm = qcol or "stream_id"      # hardware queue: a replayed graph reports stream 0 for every kernel
rows = list(db.execute(f"select start, end, name, {strm} from kernels order by start"))
rows = rows[-per:]
t0 = rows[0][0]; t1 = max(r[1] for r in rows)
print(f"# columns: {cols}")
print(f"# last {per} dispatches span {(t1 - t0) / 1e3:.1f} us")
streams = {}
for s, e, n, q in rows:
    streams.setdefault(q, []).append((s, e, n))
for q, ks in streams.items():
    busy = sum(e - s for s, e, _ in ks)
    print(f"stream/queue {q}: {len(ks)} kernels, busy {busy / 1e3:.1f} us, first {(ks[0][0] - t0) / 1e3:.1f} last end {(ks[-1][1] - t0) / 1e3:.1f}")
# union busy time over all streams and gaps
ev = sorted((s, e) for s, e, _, _ in rows)
cur_s, cur_e = ev[0]; union = 0; gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"union busy {union / 1e3:.1f} us, idle {sum(gaps) / 1e3:.1f} us in {len(gaps)} gaps (median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us)")
if "--list" in sys.argv:
    for s, e, n, q in rows:
        short = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f}  q{q}  {short}")
