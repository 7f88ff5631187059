"""Per-kernel SQ / GRBM counter summary of rocprofv3 --pmc passes (tools/pmc_sq.sh).
    python tools/pmc_sq_summary.py <pass1.db> <pass2.db> <out-prefix>
Counter units (MI355X_MICROARCH.md, per-instruction constants; calibrated on this repo's kernels, round 6):
* SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (FPS level 1: 32 waves x 296 us: 5.52e6 =
  32 x 172 k quad-cycles = 2.33 GHz);
* SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs: exactly 64 per v_mfma_f32_32x32x2_f32 (= 64 x flop / 4096);
* SQ_BUSY_CU_CYCLES counts quad-cycles summed over the SIMDs that hold at least one wave (FPS, one wave per workgroup: equal to
  SQ_WAVE_CYCLES; pw_bwd_mid_kernel, 8 waves on the 4 SIMDs of 128 CUs: 4 x 128 x the kernel's quad-cycles) -- so
  MFMA-busy / (4 x SQ_BUSY_CU_CYCLES) is the share of OCCUPIED SIMD-cycles with the matrix pipe busy;
* GRBM_GUI_ACTIVE is summed over the 8 XCDs and includes several microseconds around a short dispatch under the profiler: it
  is only used for the effective clock of long kernels (GRBM / 8 / duration); chip-level shares use the traced duration.
"""
import json
import os
import re
import sqlite3
import sys


def load(path):
    out = {}
    if not os.path.exists(path):
        return out
    db = sqlite3.connect(path)
    q = ("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name")
    for name, cn, total, cnt in db.execute(q):
        short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
        d = out.setdefault(short, {})
        d[cn] = total
        d["launches"] = max(d.get("launches", 0), cnt)
    # kernel durations of the same run (kernel-trace)
    try:
        for name, total_ns, cnt in db.execute("select name, sum(end - start), count(*) from kernels group by name"):
            short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]
            if short in out:
                out[short]["ns"] = total_ns
    except sqlite3.Error:
        pass
    return out


p1, p2, prefix = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
names = sorted(p1, key=lambda k: -p1[k].get("GRBM_GUI_ACTIVE", 0))
# effective clock under the profiler: the longest kernel's wave cycles per wave over its traced duration
CLK = 2.33
long_k = [k for k in names if p1[k].get("ns") and p2.get(k, {}).get("SQ_WAVES") and p1[k]["ns"] / p1[k]["launches"] > 1e5]
if long_k:
    k = long_k[0]
    CLK = 4.0 * p1[k]["SQ_WAVE_CYCLES"] / p2[k]["SQ_WAVES"] / (p1[k]["ns"] / p1[k]["launches"])
lines = ["# rocprofv3 --kernel-trace --pmc (two separate passes) over `bench.py --eager --steps 3 --warmup 2`: the encoder step, launch by launch",
         "# mfma_chip  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x traced duration x %.2f GHz, the clock under the profiler): share of ALL SIMD-cycles of the dispatch with the matrix pipe busy" % CLK,
         "# mfma_cu    = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): the same over the SIMDs that held a wave (SQ_BUSY_CU_CYCLES: quad-cycles per occupied SIMD)",
         "# per-wave shares (of SQ_WAVE_CYCLES): wait_inst = SQ_WAIT_INST_ANY (issue stall: dependency / pipe), wait_any = SQ_WAIT_ANY (s_waitcnt / barrier), active = SQ_ACTIVE_INST_ANY",
         "# valu/mfma = SQ_INSTS_VALU / MFMA count (SQ_VALU_MFMA_BUSY_CYCLES / 64): VALU instructions issued per matrix instruction (the MFMA itself included)",
         "%-46s %6s %9s %9s %8s %9s %8s %7s %9s" % ("kernel", "calls", "mfma_chip", "mfma_cu", "wait_inst", "wait_any", "active", "clk_GHz", "valu/mfma")]
js = {}
unit = 1.0      # SQ_BUSY_CU_CYCLES: quad-cycles per occupied SIMD -> 4 x value = SIMD-cycles (see the header)
for k in names:
    d, e = p1[k], p2.get(k, {})
    grbm, mfma, wc = d.get("GRBM_GUI_ACTIVE"), d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("SQ_WAVE_CYCLES")
    if not grbm or wc is None:
        continue
    chip = mfma / (1024.0 * d["ns"] * CLK) if (mfma is not None and d.get("ns")) else None
    cu = mfma / (4.0 * unit * d["SQ_BUSY_CU_CYCLES"]) if (mfma is not None and d.get("SQ_BUSY_CU_CYCLES")) else None
    clk = grbm / 8.0 / d["ns"] if (d.get("ns") and d["ns"] / d["launches"] > 5e4) else None
    nm = mfma / 64.0 if mfma else 0
    vm = e.get("SQ_INSTS_VALU") / nm if (nm and e.get("SQ_INSTS_VALU")) else None
    f = lambda v, fmt="%.3f": ("-" if v is None else fmt % v)
    lines.append("%-46s %6d %9s %9s %8s %9s %8s %7s %9s" % (
        k[:46], d["launches"], f(chip), f(cu), f(d.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None),
        f(d.get("SQ_WAIT_ANY", 0) / wc if wc else None), f(d.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else None),
        f(clk, "%.2f"), f(vm, "%.1f")))
    js[k] = {"launches": d["launches"], "mfma_busy_frac": chip, "mfma_busy_frac_of_busy_cus": cu,
             "wait_inst_share": d.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None,
             "wait_any_share": d.get("SQ_WAIT_ANY", 0) / wc if wc else None,
             "active_inst_share": d.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else None,
             "effective_clock_ghz": clk, "valu_per_mfma": vm,
             "raw": {**{c: v for c, v in d.items() if c not in ("launches",)}, **{c: v for c, v in e.items() if c not in ("launches", "ns")}}}
lines.append("# clk_GHz: GRBM_GUI_ACTIVE / 8 XCDs / traced duration, kernels longer than 50 us only")
open(prefix + ".txt", "w").write("\n".join(lines) + "\n")
sys.path.insert(0, os.getcwd())
import istnet_amd  # noqa: E402,F401
from istnet_amd.roofline import kernel_source_hash  # noqa: E402
json.dump({"kernel_source_sha256": kernel_source_hash(),
           "source": "rocprofv3 --kernel-trace --pmc, two separate passes over bench.py --eager (tools/pmc_sq.sh)",
           "definitions": lines[1:5], "kernels": js}, open(prefix + ".json", "w"), indent=1)
print("\n".join(lines[:40]))
