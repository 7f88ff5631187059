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


def pmc_traffic(kernel_name, path):
    """HBM bytes per launch of `kernel_name` from a committed rocprofv3 PMC summary (tools/pmc_traffic.sh);
    None when the file or the kernel is missing.  PMC counters cannot be read from inside the process."""
    import json
    import os
    if not os.path.exists(path):
        return None, None
    data = json.load(open(path))
    rec = data.get("kernels", {}).get(kernel_name)
    return (rec["hbm_bytes_per_launch"], os.path.basename(path)) if rec else (None, None)


def measure(step, steps=5, traffic_file=None):
    """`step` runs one full training step.  Returns the `roofline` object of the bench JSON line."""
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
        ms = start.elapsed_time(end)
        k = per_kernel.setdefault(name, [0.0, 0.0, 0.0, 0])
        k[0] += ms
        k[1] += flops
        k[2] += nbytes
        k[3] += 1
    if not per_kernel:
        return None
    name, (ms, flops, nbytes, launches) = max(per_kernel.items(), key=lambda kv: kv[1][0])
    achieved = flops / (ms * 1e-3) / 1e12
    total_ms = sum(v[0] for v in per_kernel.values())
    traffic, traffic_src = pmc_traffic(name, traffic_file) if traffic_file else (None, None)
    return {
        "bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_MFMA_F32_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
        "traffic_source": traffic_src,
        "launches_per_step": launches / steps, "avg_launch_us": ms * 1e3 / launches,
        "flop_per_launch_avg": flops / launches, "algorithmic_bytes_per_launch_avg": nbytes / launches,
        "algorithmic_gbs": nbytes / (ms * 1e-3) / 1e9,
        "all_gemm_kernels_ms_per_step": total_ms / steps,
        "all_gemm_kernels_tflops": sum(v[1] for v in per_kernel.values()) / (total_ms * 1e-3) / 1e12,
        "timing": "HIP events on the launch stream around each launch, %d instrumented steps" % steps,
    }
