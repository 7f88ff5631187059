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
    # steps delimited by the optimizer launches: everything between the first and the last adam_step_kernel is whole steps
    # (set-up copies / fills and the capture's warm-up outside that window do not count as per-step dispatches)
    adam = [r[0] for r in db.execute("select start from kernels where name like '%adam_step_kernel%' order by start")]
    if len(adam) > 2:
        inside = list(db.execute("select count(*), sum(end-start) from kernels where start > ? and start <= ?", (adam[0], adam[-1])))[0]
        k = len(adam) - 1
        print(f"# between the first and the last of {len(adam)} optimizer launches: {inside[0] / k:.1f} dispatches and "
              f"{inside[1] / 1e3 / k:.1f} us of kernel time per step")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {n} dispatches"
          + (f" ({steps} steps incl. warm-up: {tot / 1e6 / steps:.3f} ms, {n / steps:.0f} dispatches per step)" if steps else ""))
    print(f"{'pct':>6} {'calls':>7} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  name")
    for name, cnt, total, avg, mn, mx in rows:
        print(f"{total / tot * 100:6.2f} {cnt:7d} {total / 1e3:11.1f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}  {name[:140]}")


def by_grid(path, pattern, steps):
    db = sqlite3.connect(path)
    rows = list(db.execute(
        "select name, grid_x, grid_y, grid_z, count(*), avg(end-start), sum(end-start) from kernels "
        "where name like ? group by name, grid_x, grid_y, grid_z order by 7 desc", (f"%{pattern}%",)))
    print(f"# per-shape breakdown of kernels matching '{pattern}' (grid in threads)")
    for name, gx, gy, gz, cnt, avg, total in rows:
        short = name.split("(")[0][-60:]
        print(f"{total / 1e3 / steps:9.1f} us/step  calls/step {cnt / steps:5.1f}  avg {avg / 1e3:8.2f} us  grid ({gx},{gy},{gz})  {short}")


if __name__ == "__main__" and len(sys.argv) > 3:
    by_grid(sys.argv[1], sys.argv[3], int(sys.argv[2]))
elif __name__ == "__main__":
    main()
