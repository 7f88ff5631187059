#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh <tag>
# HBM traffic of the GEMM kernels, two separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass)
TAG=${1:-x}
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $C -d /root/repo/gpurun_out/pmc_${TAG}_$C -o p -- python /root/repo/bench.py --eager --no-roofline --no-cpu-baseline --steps 3 --warmup 2 > /root/repo/gpurun_out/pmc_${TAG}_$C.log 2>&1)
done
python - <<PY
import sqlite3
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(f"gpurun_out/pmc_${TAG}_{c}/p_results.db")
    for name, val, cnt in db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (c,)):
        res.setdefault(name, {})[c] = (val, cnt)
rows = sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[0] * kv[1].get("FETCH_SIZE", (0, 1))[1]))
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per-launch averages, raw counter units (KiB)")
print("# gfx950 note (MI355X_MICROARCH.md HBM): FETCH_SIZE under-reports wide coalesced reads by 2x; WRITE_SIZE uncalibrated")
import json, re
out = {}
for name, d in rows[:40]:
    f = d.get("FETCH_SIZE", (0, 0)); w = d.get("WRITE_SIZE", (0, 0))
    print(f"{f[0]:12.1f} KiB fetch  {w[0]:12.1f} KiB write  launches {f[1]:5d}  {name[:120]}")
    short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
    # bytes per launch: FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024
    out[short] = {"fetch_kib_raw": f[0], "write_kib_raw": w[0], "launches": f[1],
                  "hbm_bytes_per_launch": f[0] * 1024 * 2 + w[0] * 1024}
import sys, os
sys.path.insert(0, os.getcwd())
import istnet_amd
from istnet_amd.roofline import kernel_source_hash
json.dump({"kernel_source_sha256": kernel_source_hash(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --eager",
           "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported",
           "kernels": out}, open("gpurun_out/pmc_${TAG}_traffic.json", "w"), indent=1)
PY
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE      # keep gpurun_out under the 64 MiB copy-back limit
