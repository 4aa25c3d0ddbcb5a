"""Builds mulan_amd/libmulan_hip.so (gfx950) from mulan_amd/csrc/*.hip with hipcc.

In-tree build so the shared object travels with the repo snapshot to the GPU box; hipcc
cross-compiles gfx950 without a GPU.  Usage: `python -m mulan_amd.build [--force]`.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libmulan_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wno-unused-result"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libmulan_hip.so")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, extra_flags=(), variant=""):
    """variant (dev A/B builds: `python -m mulan_amd.build --variant b -DMULAN_X=1`): objects under csrc/_obj_<variant>/,
    library libmulan_hip_<variant>.so, selected at run time with MULAN_HIP_LIB=<path> (mulan_amd/lib.py)"""
    hipcc = _hipcc()
    global OBJ, LIB
    if variant:
        OBJ = os.path.join(HERE, "csrc", "_obj_" + variant)
        LIB = os.path.join(HERE, f"libmulan_hip_{variant}.so")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "mulan_hip.h"))     # csrc/common.h includes it
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, *extra_flags, "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if verbose and (r.stdout or r.stderr):
                sys.stderr.write(r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stderr)
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: " + r.stderr)
    return LIB


def build_abi_driver(out=None):
    """tests/abi_driver.cpp: the torch-free C++ caller of the C ABI (host code only), linked against the in-tree library"""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "abi_driver.cpp")
    out = out or os.path.join(root, "tests", "_bin", "abi_driver")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if _stale(out, [src, os.path.join(root, "include", "mulan_hip.h"), LIB]):
        cmd = [_hipcc(), "-O1", "-I", os.path.join(root, "include"), src, "-o", out, "-L", HERE, "-lmulan_hip",
               "-Wl,-rpath," + HERE, "-Wl,-rpath,$ORIGIN/../../mulan_amd"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("abi_driver: " + " ".join(cmd) + "\n" + r.stderr)
    return out


if __name__ == "__main__":
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build_library(force="--force" in sys.argv, variant=variant,
                        extra_flags=tuple(a for a in sys.argv[1:] if a.startswith("-D"))))
