"""The CMake face of the backend (CMakeLists.txt + cmake/SlimtHip.cmake = the WITH_HIP
provider fragment a slimt checkout would include, reference CMakeLists.txt:16-19,125-144):
configure, build the C++ mirror of the reference interface against the in-tree
libslimt_hip.so and run the resulting test driver (a mode that needs no GPU)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_cmake_builds_the_host_mirror_and_it_runs(tmp_path, oracle):
    from slimt_amd import build
    lib = build.build()
    bdir = tmp_path / "build"
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    r = subprocess.run(["cmake", "-S", ROOT, "-B", str(bdir), f"-DSLIMT_HIP_PREBUILT={lib}"] + gen,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", str(bdir), "-j", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    exe = bdir / "slimt_hip_host_test"
    assert exe.exists()
    # the batch-forming queue through the CMake-built binary against the restatement
    reqs = [[3, 5, 5, 2], [7, 1], [4, 4, 4]]
    case, out = tmp_path / "case.bin", tmp_path / "out.bin"
    with open(case, "wb") as f:
        f.write(struct.pack("<7If", 1, 1, 1, 16, 8, 1, len(reqs), 1.5))
        for segs in reqs:
            f.write(struct.pack("<I", len(segs)))
            for n in segs:
                f.write(struct.pack("<I", n) + np.zeros(n, np.uint32).tobytes())
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(lib) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe), "--batcher", str(case), str(out)], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0, r.stderr
    raw = out.read_bytes()
    got, off = [], 0
    while off < len(raw):
        n, ml = struct.unpack_from("<2I", raw, off)
        off += 8
        got.append([tuple(struct.unpack_from("<2I", raw, off + 8 * i)) for i in range(n)])
        off += 8 * n
    assert got == oracle.batcher_generate(reqs, 16, 8, 1.5)


def test_cmake_fragment_names_the_reference_hooks():
    text = open(os.path.join(ROOT, "cmake", "SlimtHip.cmake")).read()
    for needle in ("WITH_HIP", "SLIMT_PRIVATE_LIBS", "SLIMT_COMPILE_DEFINITIONS", "SLIMT_HAS_HIP", "gfx950"):
        assert needle in text
