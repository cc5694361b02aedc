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
OBJHOST_LIB = os.path.join(HERE, "libobjhost.so")
GUARD_LIB = os.path.join(HERE, "variants", "guard.so")
SHIM_ASAN = os.path.join(HERE, "variants", "shim_asan")

HIP_SOURCES = ["rt_shim.hip", "trace.hip", "wavefront.hip", "shade.hip", "lbvh.hip"]
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


def _lib_deps():
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    return srcs, srcs + hdrs + [os.path.join(ROOT, "include", "rt_shim.h"), __file__]


TRAVERSAL_SOURCES = ["wavefront.hip", "rt_traverse.h", "rt_device.h", "rt_internal.h", "lbvh.hip"]


def source_hash():
    """sha256 over the sources that decide what the traversal kernel fetches (the kernel itself, its device headers and
    the hierarchy builder): the key that ties a committed counter measurement (profiles/hbm_traffic.json) to the
    kernels it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in TRAVERSAL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def build_lib(force=False, verbose=False, extra_flags=(), out=LIB):
    srcs, deps = _lib_deps()
    # the up-to-date shortcut only knows the default flags of each output file: extra flags on the library's own
    # path would otherwise return a stale build without them
    if not force and _newer(out, deps) and not (extra_flags and out == LIB):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [_hipcc()] + HIP_FLAGS + list(extra_flags) + srcs + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_guard(force=False, verbose=False):
    """The guarded variant (-DRT_GUARD=1: every index that comes out of device data is checked against its
    array before use).  Built with the library so that it cannot rot; kernel changes are tried on it first
    (pytest --rt-lib raytracer_amd/variants/guard.so, bench.py --lib ...)."""
    return build_lib(force=force, verbose=verbose, extra_flags=("-DRT_GUARD=1",), out=GUARD_LIB)


def build_shim_asan(force=False, verbose=False):
    """variants/shim_asan: tests/shim_asan_driver.cpp linked with the library's own sources, the HOST side compiled with the
    address and undefined-behaviour sanitizers (device code as always).  A GPU test runs it: the shim's buffer, vector and
    thread handling under the sanitizers, through the C ABI."""
    out = SHIM_ASAN
    srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES] + [os.path.join(ROOT, "tests", "shim_asan_driver.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "rt_shim.h")]
    if not force and _newer(out, deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flags = [f for f in HIP_FLAGS if f not in ("-fPIC", "-shared", "-O3")]
    cmd = [_hipcc()] + flags + ["-O2", "-g", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
                                "-Xarch_host", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + srcs + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out


def build_objhost(force=False, verbose=False):
    """libobjhost.so: include/rt_objhost.h, the host's OBJ loader behind a C ABI (no GPU code)."""
    srcs = [os.path.join(HOST, "objhost_c.cpp"), os.path.join(HOST, "objloader.cpp")]
    deps = srcs + [os.path.join(HOST, "objloader.h"), os.path.join(HOST, "maths.h"), os.path.join(ROOT, "include", "rt_objhost.h")]
    if not force and _newer(OBJHOST_LIB, deps):
        return OBJHOST_LIB
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-fPIC", "-shared"] + srcs + ["-o", OBJHOST_LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OBJHOST_LIB


def build_host(force=False, verbose=False):
    srcs = [os.path.join(HOST, s) for s in sorted(os.listdir(HOST)) if s.endswith(".cpp") and s != "objhost_c.cpp"] if os.path.isdir(HOST) else []
    if not srcs:
        return None
    hdrs = [os.path.join(HOST, s) for s in os.listdir(HOST) if s.endswith(".h")]
    if not force and _newer(HOST_BIN, srcs + hdrs + [LIB]):
        return HOST_BIN
    os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
    # multigpu.cpp calls the HIP runtime and RCCL directly (streams, buffers, the all-gather of the slices)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    # (plain g++: the HIP headers want to be told the platform, as hipcc tells them)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__=1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include")] + srcs + [
        "-o", HOST_BIN, "-L", HERE, "-lrt_amd", "-L", os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64",
        "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE, "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return HOST_BIN


def build_all(force=False, verbose=False):
    build_lib(force=force, verbose=verbose)
    build_guard(force=force, verbose=verbose)
    build_shim_asan(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)
    build_objhost(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
