"""ctypes binding of libistnet_pn2.so (include/istnet_pn2.h).

There is NO fallback: if the library is missing or a launch fails, the caller gets a RuntimeError.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libistnet_pn2.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "istnet_pn2.h")
ABI_VERSION = 1

_i, _f, _p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
# name -> argtypes (restype is always int); mirrors include/istnet_pn2.h
SIGNATURES = {
    "istnet_pn2_furthest_point_sampling": [_i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_gather_points": [_i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_gather_points_grad": [_i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_query_ball_point": [_i, _i, _i, _f, _i, _p, _p, _p, _p],
    "istnet_pn2_group_points": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_group_points_grad": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_three_nn": [_i, _i, _i, _p, _p, _p, _p, _p],
    "istnet_pn2_three_interpolate": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "istnet_pn2_three_interpolate_grad": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
}

_lib = None


def declared_symbols():
    """Every function name declared in include/istnet_pn2.h."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    return sorted(set(re.findall(r"\b(istnet_pn2_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python __graft_entry__.py build`). There is no CPU or PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        handle.istnet_pn2_abi_version.restype = ctypes.c_int
        handle.istnet_pn2_target.restype = ctypes.c_char_p
        got = handle.istnet_pn2_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libistnet_pn2.so ABI {got} != expected {ABI_VERSION}; rebuild")
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status} (hipError_t / ISTNET_PN2_EINVAL=100001)")
