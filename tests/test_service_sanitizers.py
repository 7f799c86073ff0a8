"""host/Service's threading under ThreadSanitizer and AddressSanitizer, on the CPU: the service is
linked against a TEST DOUBLE of the engine's C ABI (tests/support/fake_hip_engine.cc: a sentence's
"translation" is its tokens reversed, completed on a helper thread) and driven by several client
threads over two replicas with three double-buffered workers each; then the failure modes (refused
request, failing batch, workers that cannot be built, a service with none left)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "slimt_amd", "host")
SUPPORT = os.path.join(ROOT, "tests", "support")


def _build(out, sanitizer):
    srcs = [os.path.join(HOST, f) for f in ("Service.cc", "Model.cc", "Shortlist.cc", "Io.cc")] + \
           [os.path.join(SUPPORT, f) for f in ("fake_hip_engine.cc", "service_stress.cc")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={sanitizer}", "-I", HOST,
           "-I", os.path.join(ROOT, "include")] + srcs + ["-pthread", "-o", out]
    subprocess.check_call(cmd)


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_service_threading_under_sanitizers(sanitizer):
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "service_stress")
        _build(exe, sanitizer)
        env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
                   UBSAN_OPTIONS="halt_on_error=1")
        res = subprocess.run([exe, "4", "25"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ThreadSanitizer" not in res.stderr and "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr
    out = res.stdout
    assert " 0 wrong" in out and "rejected empty" in out and "engine failure reported" in out
    assert "survived: 2 sentences" in out
    assert "half the workers retired: 40 sentences translated" in out
    assert "dead service: " in out and "fake: out of device memory" in out
