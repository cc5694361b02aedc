"""Build recipe for the native parts: librt_amd.so (HIP, gfx950) and the C++ host.

hipcc cross-compiles gfx950 without a GPU, so this runs anywhere the ROCm
toolchain is installed.  Outputs stay in-tree (git-ignored) next to the sources.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "librt_amd.so")
HOST_BIN = os.path.join(HERE, "bin", "raytracer")

HIP_SOURCES = ["rt_shim.hip", "trace.hip", "wavefront.hip", "persistent.hip", "lbvh.hip"]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # the trace arithmetic must round like the reference's scalar x86 code:
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_lib(force=False, verbose=False, extra_flags=()):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "rt_internal.h"), os.path.join(CSRC, "rt_device.h"), os.path.join(CSRC, "rt_traverse.h"), os.path.join(ROOT, "include", "rt_shim.h"), __file__]
    if not force and not extra_flags and _newer(LIB, deps):
        return LIB
    cmd = [_hipcc()] + HIP_FLAGS + list(extra_flags) + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


def build_host(force=False, verbose=False):
    srcs = [os.path.join(HOST, s) for s in sorted(os.listdir(HOST)) if s.endswith(".cpp")] if os.path.isdir(HOST) else []
    if not srcs:
        return None
    hdrs = [os.path.join(HOST, s) for s in os.listdir(HOST) if s.endswith(".h")]
    if not force and _newer(HOST_BIN, srcs + hdrs + [LIB]):
        return HOST_BIN
    os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include")] + srcs + [
        "-o", HOST_BIN, "-L", HERE, "-lrt_amd", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return HOST_BIN


def build_all(force=False, verbose=False):
    build_lib(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
