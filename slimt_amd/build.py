"""Build libslimt_hip.so (gfx950 only) in-tree with hipcc.

The library is the product: HIP kernels + the extern "C" boundary declared in
include/slimt_hip.h. It is built into slimt_amd/lib/ so that it travels with
the source tree (no JIT cache, no site-packages install).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
# SLIMT_HIP_LIB: build/load an alternative library file (kernel experiments only)
LIB_PATH = os.environ.get("SLIMT_HIP_LIB") or os.path.join(LIB_DIR, "libslimt_hip.so")

SOURCES = ["kernels.hip", "gemm_tile.hip", "decode_kernels.hip", "decode_fused.hip", "encode_fused.hip", "encode_wide.hip", "encode_tall.hip", "shortlist.hip",
           "engine.cpp"]
HEADERS = ["kernels.h", "engine.h", "device_common.h", "shortlist_device.h", "decode_attention_packed.inl.h",
           os.path.join(ROOT, "include", "slimt_hip.h")]

# -ffp-contract=off: the float epilogues are written operation by operation
# (separate mul/add like intgemm's callbacks); fused ops are explicit fmaf.
FLAGS = [
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "--offload-arch=gfx950",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-gpu-flush-denormals-to-zero",
    "-Wall",
    "-Wno-unused-result",
    "-x", "hip",
]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS
    ] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


OBJ_DIR = os.path.join(LIB_DIR, "obj")


def _compile_one(args):
    src, obj, cmd, verbose = args
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (compiled in parallel, rebuilt only when the source or a header
    changed), then one link: an edit to one kernel file costs that file's compile time."""
    if not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    extra = os.environ.get("SLIMT_HIPCC_EXTRA", "").split()  # experiments only
    # (a stable tag: the interpreter's string hash changes per process and left one set of objects per build behind)
    import zlib
    tag = ("_x%08x" % zlib.crc32(" ".join(extra).encode())) if extra else ""
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    cflags = [f for f in FLAGS if f != "-shared"]
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + tag + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_hdr, os.path.getmtime(src)):
            jobs.append((src, obj, [hipcc()] + cflags + extra + ["-c", src, "-I", CSRC, "-I",
                                                                  os.path.join(ROOT, "include"), "-o", obj], verbose))
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(_compile_one, jobs))
    link = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return LIB_PATH


HOST_DIR = os.path.join(HERE, "host")
HOST_TEST = os.path.join(LIB_DIR, "slimt_hip_host_test")
HOST_SOURCES = ["Io.cc", "QMM.cc", "Model.cc", "Shortlist.cc", "Service.cc", "Transformer.cc", "host_test.cc"]


def build_host(force: bool = False) -> str:
    """The C++ host mirror of the reference interface (slimt::qmm, Model/Worker)
    + its test driver, linked against libslimt_hip.so like slimt itself would be."""
    build()
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, h) for h in os.listdir(HOST_DIR)] + [LIB_PATH]
    if not force and os.path.exists(HOST_TEST) and all(
            os.path.getmtime(d) <= os.path.getmtime(HOST_TEST) for d in deps):
        return HOST_TEST
    rocm_lib = "/opt/rocm/lib"
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", HOST_DIR,
           "-I", os.path.join(ROOT, "include")] + srcs + [
        "-L", LIB_DIR, "-lslimt_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + rocm_lib,
        "-Wl,-rpath," + rocm_lib, "-pthread", "-o", HOST_TEST]
    subprocess.check_call(cmd)
    return HOST_TEST


HOST_LIB = os.path.join(LIB_DIR, "libslimt_hip_host.so")
HOST_LIB_SOURCES = ["Io.cc", "Model.cc", "Shortlist.cc", "Service.cc", "service_capi.cc"]


def build_host_lib(force: bool = False) -> str:
    """libslimt_hip_host.so: the C++ Service behind include/slimt_hip_service.h (what a binding
    links next to libslimt_hip.so)."""
    build()
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_LIB_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, h) for h in os.listdir(HOST_DIR)] + [
        LIB_PATH, os.path.join(ROOT, "include", "slimt_hip_service.h")]
    if not force and os.path.exists(HOST_LIB) and all(
            os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    rocm_lib = "/opt/rocm/lib"
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", HOST_DIR,
           "-I", os.path.join(ROOT, "include")] + srcs + [
        "-L", LIB_DIR, "-lslimt_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + rocm_lib,
        "-Wl,-rpath," + rocm_lib, "-pthread", "-o", HOST_LIB]
    subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv))
    print(build_host_lib(force="--force" in sys.argv))
