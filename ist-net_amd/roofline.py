"""Live roofline measurement for bench.py.

Times every launch of the fused per-point MLP GEMM kernels inside real training steps with HIP
events recorded on the stream the kernels are launched on, finds the kernel (exact template
instance, as rocprofv3 names it) with the largest total time, and reports its achieved fp32 MFMA
rate against the chip peak of /opt/skills/guides/MI355X_MICROARCH.md (157.3 TFLOP/s for
v_mfma_f32_32x32x2_f32; there is no TF32 on gfx950).

Algorithmic work per launch: 2 * B*P * Cin * Cout flop for one layer GEMM
(forward, dgrad and wgrad alike), with B*P = points x samples processed by the launch.
"""
import torch

from . import _native

PEAK_MFMA_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def kernel_source_hash():
    """sha256 over csrc/*.hip (sorted by name): stamps a PMC summary with the kernel sources it was measured on."""
    import hashlib
    import os
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()


def pmc_traffic(kernel_name, path):
    """HBM bytes per launch of `kernel_name` from a committed rocprofv3 PMC summary (tools/pmc_traffic.sh).  PMC
    counters cannot be read from inside the process, so the summary carries the hash of the kernel sources it was
    collected on (kernel_source_hash): a summary of other sources is STALE and reported as traffic = null with the
    reason, never as a number.  (None, None) when the file or the kernel is missing."""
    import json
    import os
    if not os.path.exists(path):
        return None, None
    data = json.load(open(path))
    rec = data.get("kernels", {}).get(kernel_name)
    if not rec:
        return None, None
    if data.get("kernel_source_sha256") != kernel_source_hash():
        return None, "stale: %s was collected on other kernel sources (re-run tools/pmc_traffic.sh)" % os.path.basename(path)
    return rec["hbm_bytes_per_launch"], os.path.basename(path)


def pmc_sq(kernel_name, path):
    """MFMA-pipe busy fraction of `kernel_name` from a committed rocprofv3 SQ-counter summary (tools/pmc_sq.sh):
    SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) over the launches of one profiled run.  Same staleness rule as
    pmc_traffic: a summary collected on other kernel sources is reported as null with the reason."""
    import json
    import os
    if not path or not os.path.exists(path):
        return None, None
    data = json.load(open(path))
    rec = data.get("kernels", {}).get(kernel_name)
    if not rec:
        return None, None
    if data.get("kernel_source_sha256") != kernel_source_hash():
        return None, "stale: %s was collected on other kernel sources (re-run tools/pmc_sq.sh)" % os.path.basename(path)
    return rec, os.path.basename(path)


def measure_replayed(capture, replays=5, only=None, errors=None):
    """Times every GEMM launch INSIDE replays of the captured step: ``capture()`` must capture the step into HIP graphs
    while _native.TIMING is active and return a function that replays them once.  Each launch is bracketed by two
    marker kernels on its launch stream that store the device wall clock (_native.timed), so the durations are those of
    the replayed step -- the side streams and the deferred weight gradients running beside the timed kernel -- which is
    what a rocprofv3 kernel trace of the bench shows (plus the two launch gaps inside the bracket).
    Returns per-kernel [ms, flops, bytes, launches] summed over the replays, or None if the capture fails."""
    buf = torch.zeros(8192, dtype=torch.int64, device="cuda")
    _native.TIMING, _native.TIMING_IN_GRAPH, _native.TIMING_BUF, _native.TIMING_ONLY = [], True, buf, only
    try:
        replay = capture()
        records = list(_native.TIMING)
    except Exception as exc:
        if errors is not None:
            errors.append(f"{type(exc).__name__}: {exc}")
        torch.cuda.synchronize()
        return None
    finally:
        _native.TIMING, _native.TIMING_IN_GRAPH, _native.TIMING_BUF, _native.TIMING_ONLY = None, False, None, None
    per_kernel, samples, empty = {}, [], []
    for _ in range(replays):
        buf.zero_()
        torch.cuda.synchronize()
        replay()
        torch.cuda.synchronize()
        t = buf.cpu()
        for name, flops, nbytes, i, _unused in records:
            t0, t1 = int(t[2 * i]), int(t[2 * i + 1])
            if t0 == 0 or t1 < t0:
                continue            # a launch that this replay did not execute
            (empty if name == "" else samples).append((name, flops, nbytes, (t1 - t0) / 1e5))   # 100 MHz ticks -> ms
    if not samples:
        return None
    empty.sort(key=lambda r: r[3])
    # an empty bracket is ONE kernel boundary (marker -> marker) under the step's load; a bracket around a launch has two
    # (marker -> kernel, kernel -> marker)
    boundary = empty[len(empty) // 2][3] if empty else 0.0
    overhead = 2.0 * boundary
    for name, flops, nbytes, ms in samples:
        k = per_kernel.setdefault(name, [0.0, 0.0, 0.0, 0])
        k[0] += max(ms - overhead, 1e-4)
        k[1] += flops
        k[2] += nbytes
        k[3] += 1
    per_kernel["__bracket_overhead_us__"] = overhead * 1e3
    return per_kernel


def _eager_events(step, steps):
    step()
    torch.cuda.synchronize()
    _native.TIMING = []
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        records = _native.TIMING
    finally:
        _native.TIMING = None
    per_kernel = {}
    for name, flops, nbytes, start, end in records:
        k = per_kernel.setdefault(name, [0.0, 0.0, 0.0, 0])
        k[0] += start.elapsed_time(end)
        k[1] += flops
        k[2] += nbytes
        k[3] += 1
    return per_kernel


def measure(step, steps=5, traffic_file=None, capture=None, steps_per_replay=1, sq_file=None):
    """`step` runs one full training step eagerly.  Returns the `roofline` object of the bench JSON line.

    Pass 1 (HIP events around every GEMM launch of instrumented eager steps, single stream) finds the kernel with the
    largest total time and gives the isolated rate.  Pass 2, when ``capture`` is given (see measure_replayed), times THAT
    kernel inside replays of the captured step, where it shares the chip with the other streams -- the number a
    rocprofv3 kernel trace of the bench shows; it is the one reported as ``achieved``."""
    eager = _eager_events(step, steps)
    if not eager:
        return None
    name = max(eager.items(), key=lambda kv: kv[1][0])[0]
    obj = _roofline_object(eager, steps, traffic_file,
                           "HIP events on the launch stream around each launch, %d instrumented eager steps" % steps)
    sq, sq_src = pmc_sq(name, sq_file)
    if sq is not None:
        # the share of the SIMD-cycles of the CUs the kernel occupies with the matrix pipe busy (the definition the round-5 review
        # asked for), and the same over ALL SIMD-cycles of the dispatch (comparable with `frac`, which is flop / time / peak)
        obj["mfma_busy_frac"] = sq.get("mfma_busy_frac_of_busy_cus")
        obj["mfma_busy_frac_chip"] = sq.get("mfma_busy_frac")
        obj["mfma_busy_source"] = sq_src
        obj["mfma_busy_note"] = ("mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): matrix pipe busy over the "
                                 "SIMD-cycles of the occupied CUs; mfma_busy_frac_chip = the same over 1024 SIMDs x the traced "
                                 "duration; rocprofv3 --pmc run of the encoder step, counters in their own pass (profiles/%s)" % sq_src)
    elif sq_src is not None:
        obj["mfma_busy_frac"], obj["mfma_busy_source"] = None, sq_src
    replayed = measure_replayed(capture, steps, only=name) if capture is not None else None
    # the dominant FAMILY on the same clock as the dominant kernel: its launches bracketed inside replays of the step
    fam_base = obj["dominant_family"]["kernel"].split("<")[0]
    fam_errors = []
    fam_replayed = (measure_replayed(capture, steps, only=lambda n: n.split("<")[0] == fam_base, errors=fam_errors)
                    if capture is not None else None)
    if fam_errors:
        obj["dominant_family"]["replay_timing_error"] = fam_errors[0][:300]
    if fam_replayed:
        fam_replayed.pop("__bracket_overhead_us__", None)
        fms = sum(v[0] for v in fam_replayed.values())
        fflops = sum(v[1] for v in fam_replayed.values())
        fbytes = sum(v[2] for v in fam_replayed.values())
        fl = sum(v[3] for v in fam_replayed.values())
        if fms > 0 and fl:
            d = obj["dominant_family"]
            d.update({"achieved_eager_isolated": d["achieved"], "frac_eager_isolated": d["frac"],
                      "achieved": fflops / (fms * 1e-3) / 1e12,
                      "frac": fflops / (fms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS,
                      "algorithmic_gbs": fbytes / (fms * 1e-3) / 1e9,
                      "us_per_step_in_replay": fms * 1e3 / (steps * steps_per_replay),
                      "launches_per_step": fl / (steps * steps_per_replay),
                      "timing": "device wall-clock markers around every launch of the family inside replays of the captured "
                                "step (the clock of roofline.achieved); *_eager_isolated: HIP events, instrumented eager steps"})
    if replayed and name in replayed:
        overhead_us = replayed.pop("__bracket_overhead_us__", None)
        ms, flops, nbytes, launches = replayed[name]
        achieved = flops / (ms * 1e-3) / 1e12
        obj.update({
            "achieved_eager_isolated": obj["achieved"], "frac_eager_isolated": obj["frac"],
            "avg_launch_us_eager_isolated": obj["avg_launch_us"],
            "achieved": achieved, "frac": achieved / PEAK_MFMA_F32_TFLOPS, "avg_launch_us": ms * 1e3 / launches,
            "launches_per_step": launches / (steps * steps_per_replay),
            "algorithmic_gbs": nbytes / (ms * 1e-3) / 1e9, "bracket_overhead_us": overhead_us,
            "timing": ("device wall-clock markers on the launch stream around each launch of this kernel INSIDE %d replayed "
                       "steps of the captured graph(s) (HIP cannot read events recorded during capture), minus two kernel "
                       "boundaries as measured by empty brackets placed in the same graph; the other GEMM kernels (all_gemm_*) and the "
                       "*_eager_isolated fields: HIP events in instrumented eager steps" % (steps * steps_per_replay))})
    return obj


def _roofline_object(per_kernel, steps, traffic_file, timing):
    if not per_kernel:
        return None
    name, (ms, flops, nbytes, launches) = max(per_kernel.items(), key=lambda kv: kv[1][0])
    achieved = flops / (ms * 1e-3) / 1e12
    total_ms = sum(v[0] for v in per_kernel.values())
    traffic, traffic_src = pmc_traffic(name, traffic_file) if traffic_file else (None, None)
    # kernel FAMILIES (template instances of one kernel summed): the dominant instance is 7 % of the kernel time, the
    # dominant family two to three times that
    fam = {}
    for kname, v in per_kernel.items():
        f = fam.setdefault(kname.split("<")[0], [0.0, 0.0, 0.0, 0])
        for i in range(4):
            f[i] += v[i]
    fname, (fms, fflops, fbytes, flaunches) = max(fam.items(), key=lambda kv: kv[1][0])
    all_tflops = sum(v[1] for v in per_kernel.values()) / (total_ms * 1e-3) / 1e12
    return {
        "bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_MFMA_F32_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
        "traffic_source": traffic_src,
        "launches_per_step": launches / steps, "avg_launch_us": ms * 1e3 / launches,
        "flop_per_launch_avg": flops / launches, "algorithmic_bytes_per_launch_avg": nbytes / launches,
        "algorithmic_gbs": nbytes / (ms * 1e-3) / 1e9,
        "all_gemm_kernels_ms_per_step": total_ms / steps,
        # what the step EXECUTES per step in its GEMM launches (layer 0 of an SA scale runs over the source points, level 1 on
        # compact columns, no coordinate gradient): less than SURVEY 8d's nominal 140.6 GFLOP, which step_flops_frac divides by
        "executed_gemm_gflop_per_step": sum(v[1] for v in per_kernel.values()) / steps / 1e9,
        "all_gemm_kernels_tflops": all_tflops,
        "all_gemm_kernels_frac": all_tflops / PEAK_MFMA_F32_TFLOPS,        # plain FLOP/s over the fp32 MFMA peak
        "dominant_family": {"kernel": fname + "<*>" if any(k.startswith(fname + "<") for k in per_kernel) else fname,
                            "share_of_gemm_time": fms / total_ms, "launches_per_step": flaunches / steps,
                            "achieved": fflops / (fms * 1e-3) / 1e12,
                            "frac": fflops / (fms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS,
                            "algorithmic_gbs": fbytes / (fms * 1e-3) / 1e9,
                            "timing": "HIP events, instrumented eager steps (isolated launches)"},
        # NOT a FLOP/s fraction: every timed launch against a two-roof MODEL -- the time the roofs allow (per kernel
        # instance the larger of flops / MFMA peak and algorithmic bytes / HBM peak) over the time taken, eager and isolated
        "all_gemm_kernels_two_roof_model_frac": sum(
            max(v[1] / (PEAK_MFMA_F32_TFLOPS * 1e12), v[2] / (PEAK_HBM_GBS * 1e9)) for v in per_kernel.values())
        / (total_ms * 1e-3),
        "timing": timing,
    }
