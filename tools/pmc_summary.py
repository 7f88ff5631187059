"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite db.
    python tools/pmc_summary.py <db> [name-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
if "--schema" in sys.argv:
    print(cols); print([r[1] for r in db.execute("pragma table_info(pmc_events)")]); sys.exit()
rows = db.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, avg(value), count(*) from counters_collection "
                  "where kernel_name like ? group by kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name", (f"%{pat}%",)).fetchall()
agg = {}
for name, gx, gy, gz, cn, val, cnt in rows:
    agg.setdefault((name.replace("void (anonymous namespace)::","").split("(")[0][:40], gx, gy, gz), {})[cn] = val
for k, d in sorted(agg.items()):
    line = " ".join(f"{c}={v:.3g}" for c, v in sorted(d.items()))
    if "SQ_WAVE_CYCLES" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        wc = d["SQ_WAVE_CYCLES"] * 4  # quad-cycles -> cycles, summed over waves
        line += f" | mfma_busy/wave_cycles={d['SQ_VALU_MFMA_BUSY_CYCLES'] / wc:.2f}"
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in d: line += f" {c[3:]}={d[c] * 4 / wc:.2f}"
    print(k, line)
