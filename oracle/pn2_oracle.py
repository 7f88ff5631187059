"""CPU oracle for the nine ``pointnet2._ext`` functions (TEST INFRASTRUCTURE ONLY).

This module plays the role of the reference's pybind module ``pointnet2._ext``
(model/pointnet2/_ext_src/src/bindings.cpp:11-24) on CPU tensors, backed by the
C restatement in ``pn2_oracle.c``.  Host-side behaviour (contiguity / dtype
checks, zero-initialised outputs, ``temp`` = 1e10) follows the four reference
``.cpp`` host files, cited per function.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.  Parity status: see the header of pn2_oracle.c
("parity unpinned" for kernel semantics).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpn2_oracle.so")


def build(force=False):
    """Compile pn2_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pn2_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _chk_f(t, name):
    _check(t.is_contiguous(), f"{name} must be a contiguous tensor")
    _check(t.dtype == torch.float32, f"{name} must be a float tensor")
    _check(t.device.type == "cpu", f"oracle: {name} must be a CPU tensor")


def _chk_i(t, name):
    _check(t.is_contiguous(), f"{name} must be a contiguous tensor")
    _check(t.dtype == torch.int32, f"{name} must be an int tensor")
    _check(t.device.type == "cpu", f"oracle: {name} must be a CPU tensor")


def set_convention(c):
    """Distance convention of FPS / ball query / three_nn: 0 un-contracted (default; what the product and the golden
    vectors use), 1 / 2 the FMA-contracted forms (see pn2_oracle.c).  Returns the previous value."""
    prev = lib().oracle_get_convention()
    if lib().oracle_set_convention(int(c)) != 0:
        raise ValueError(f"oracle: unknown distance convention {c}")
    return prev


def set_threads(n):
    """OpenMP team size of the oracle's batch-parallel loops (cpu_baseline thread sweep)."""
    lib().oracle_set_threads(int(n))


def opt_n_threads(work_size):
    return lib().oracle_opt_n_threads(int(work_size))


# sampling.cpp:20-43
def gather_points(points, idx):
    _chk_f(points, "points"); _chk_i(idx, "idx")
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.zeros(b, c, m, dtype=torch.float32)
    lib().oracle_gather_points(b, c, n, m, _p(points), _p(idx), _p(out))
    return out


# sampling.cpp:45-68
def gather_points_grad(grad_out, idx, n):
    _chk_f(grad_out, "grad_out"); _chk_i(idx, "idx")
    b, c, m = grad_out.shape
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().oracle_gather_points_grad(b, c, int(n), m, _p(grad_out), _p(idx), _p(out))
    return out


# sampling.cpp:70-91
def furthest_point_sampling(points, nsamples):
    _chk_f(points, "points")
    b, n, _ = points.shape
    out = torch.zeros(b, nsamples, dtype=torch.int32)
    tmp = torch.full((b, n), 1e10, dtype=torch.float32)
    lib().oracle_furthest_point_sampling(b, n, int(nsamples), _p(points), _p(tmp), _p(out))
    return out


# interpolate.cpp:19-45
def three_nn(unknowns, knows):
    _chk_f(unknowns, "unknowns"); _chk_f(knows, "knows")
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    idx = torch.zeros(b, n, 3, dtype=torch.int32)
    dist2 = torch.zeros(b, n, 3, dtype=torch.float32)
    lib().oracle_three_nn(b, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx))
    return [dist2, idx]


# interpolate.cpp:47-74
def three_interpolate(points, idx, weight):
    _chk_f(points, "points"); _chk_i(idx, "idx"); _chk_f(weight, "weight")
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().oracle_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


# interpolate.cpp:75-104
def three_interpolate_grad(grad_out, idx, weight, m):
    _chk_f(grad_out, "grad_out"); _chk_i(idx, "idx"); _chk_f(weight, "weight")
    b, c, n = grad_out.shape
    out = torch.zeros(b, c, int(m), dtype=torch.float32)
    lib().oracle_three_interpolate_grad(b, c, n, int(m), _p(grad_out), _p(idx), _p(weight), _p(out))
    return out


# ball_query.cpp:13-37
def ball_query(new_xyz, xyz, radius, nsample):
    _chk_f(new_xyz, "new_xyz"); _chk_f(xyz, "xyz")
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.zeros(b, m, int(nsample), dtype=torch.int32)
    lib().oracle_query_ball_point(b, n, m, ctypes.c_float(radius), int(nsample),
                                  _p(new_xyz), _p(xyz), _p(idx))
    return idx


# group_points.cpp:17-40
def group_points(points, idx):
    _chk_f(points, "points"); _chk_i(idx, "idx")
    b, c, n = points.shape
    npoints, nsample = idx.shape[1], idx.shape[2]
    out = torch.zeros(b, c, npoints, nsample, dtype=torch.float32)
    lib().oracle_group_points(b, c, n, npoints, nsample, _p(points), _p(idx), _p(out))
    return out


# group_points.cpp:42-65
def group_points_grad(grad_out, idx, n):
    _chk_f(grad_out, "grad_out"); _chk_i(idx, "idx")
    b, c, npoints, nsample = grad_out.shape
    out = torch.zeros(b, c, int(n), dtype=torch.float32)
    lib().oracle_group_points_grad(b, c, int(n), npoints, nsample, _p(grad_out), _p(idx), _p(out))
    return out
