"""Builds libistnet_pn2.so (the C-ABI library declared in include/istnet_pn2.h) with hipcc for gfx950.

In-tree build: the .so lands in ist-net_amd/lib/ so it travels with the source snapshot.
``-ffp-contract=off`` is part of the numerical contract of the index ops.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libistnet_pn2.so")
ARCH = "gfx950"
# -ffp-contract per source.  "off" for the index ops is part of their numerical contract (the
# squared distances that decide an index are IEEE f32 in the reference's source order, no FMA contraction).
FP_CONTRACT = {"pn2_index_ops.hip": "off", "preproc.hip": "off"}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libistnet_pn2.so cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    newest = max(os.path.getmtime(p) for p in sources() + [
        os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))])
    return os.path.getmtime(LIB_PATH) < newest


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    common = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
              "-Wno-unused-function"]
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    objs, procs = [], []
    for src in sources():       # one translation unit per file, compiled in parallel, each with its own FP contract
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        contract = os.environ.get("ISTNET_FP_CONTRACT_" + os.path.basename(src)[:-4].upper(),
                                  FP_CONTRACT.get(os.path.basename(src), "off"))
        extra = os.environ.get("ISTNET_HIPCC_FLAGS", "").split()      # experiments (e.g. -DISTNET_BWD_SMALL_WAVES=3)
        cmd = common + extra + [f"-ffp-contract={contract}", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = common + ["-shared", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
