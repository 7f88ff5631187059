"""Every dispatch of the LAST step of a rocprofv3 --kernel-trace run of bench.py (rocpd .db), in start order: offset from the
step's first kernel, duration, queue, short name.  (The tracer serialises dispatches, so durations are each kernel alone.)

    python tools/step_kernel_list.py <results.db> <dispatches per step>
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
rows = list(db.execute(f"select name, start, end, {qcol or 0} from kernels order by start"))
if per > 0:
    rows = rows[-per:]
else:       # per = 0: the last whole step, delimited by the optimizer launches
    ends = [i for i, r in enumerate(rows) if "adam_step_kernel" in r[0]]
    rows = rows[ends[-2] + 1:ends[-1] + 1]
    print(f"# {len(rows)} dispatches, {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us from the first start to the last end")
    busy = {}
    for name, s0, e0, q in rows:
        busy[q] = busy.get(q, 0) + (e0 - s0)
    print("# busy per queue (us):", {f"q{q}": round(v / 1e3, 1) for q, v in sorted(busy.items())})
t0 = rows[0][1]
for name, s, e, q in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {short}")
