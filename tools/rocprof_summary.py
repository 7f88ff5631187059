"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) as a per-kernel stats table.

    python tools/rocprof_summary.py gpurun_out/prof0/enc_results.db [steps] > profiles/r01_x.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = list(db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f"# source: {sys.argv[1]}")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {n} dispatches"
          + (f" ({steps} steps incl. warm-up: {tot / 1e6 / steps:.3f} ms, {n / steps:.0f} dispatches per step)" if steps else ""))
    print(f"{'pct':>6} {'calls':>7} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  name")
    for name, cnt, total, avg, mn, mx in rows:
        print(f"{total / tot * 100:6.2f} {cnt:7d} {total / 1e3:11.1f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}  {name[:140]}")


if __name__ == "__main__":
    main()
