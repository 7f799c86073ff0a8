"""CPU-side checks of the C-ABI boundary: the shared library loads, exports
every entry point include/slimt_hip.h declares, its host-side functions work,
and its compute entry points FAIL LOUDLY without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "slimt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slimt_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from slimt_amd import build, capi
    path = build.build()
    assert os.path.exists(path)
    dll = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(dll, n), f"{n} declared in slimt_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == names
    assert capi.lib().slimt_hip_abi_version() == 3


_CHILD_ENV = r"""
import ctypes, os, sys
sys.path.insert(0, sys.argv[1])
libc = ctypes.CDLL(None)
libc.getenv.restype = ctypes.c_char_p
libc.getenv.argtypes = [ctypes.c_char_p]
assert libc.getenv(b"GPU_MAX_HW_QUEUES") is None
from slimt_amd import capi
L = capi.lib()
# loading the library runs nothing that writes the environment (it used to setenv in a constructor)
assert libc.getenv(b"GPU_MAX_HW_QUEUES") is None and "GPU_MAX_HW_QUEUES" not in os.environ
assert L.slimt_hip_hw_queues() == 0
assert L.slimt_hip_request_hw_queues(0) < 0 and b"not in 1..1024" in L.slimt_hip_last_error()
assert capi.request_hw_queues(16) is True
assert libc.getenv(b"GPU_MAX_HW_QUEUES") == b"16" and L.slimt_hip_hw_queues() == 16
assert capi.request_hw_queues(32) is True and L.slimt_hip_hw_queues() == 16  # a value already chosen stays
capi.device_count()                                                          # the library's first HIP call
assert capi.request_hw_queues(32) is False                                   # too late: says so, changes nothing
assert L.slimt_hip_hw_queues() == 16
print("ok")
"""

_CHILD_ORDER = r"""
import sys
sys.path.insert(0, sys.argv[1])
def runtimes():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
from slimt_amd import capi
if sys.argv[2] == "torch_first":
    import torch
    capi.lib()
else:
    capi.lib()
    import torch
r = runtimes()
assert len(r) == 1, r
print("ok", r[0])
"""


def _run_child(code, *args):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    return subprocess.run([sys.executable, "-c", code, ROOT, *args], capture_output=True, text=True, timeout=300, env=env)


def test_library_leaves_the_environment_alone():
    """No load-time side effects: GPU_MAX_HW_QUEUES is written by slimt_hip_request_hw_queues only, only when
    the process has not chosen a value, and not after the library's first HIP call."""
    r = _run_child(_CHILD_ENV)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("order", ["torch_first", "ours_first"])
def test_one_hip_runtime_in_either_import_order(order):
    """PyTorch bundles a HIP runtime; the process must end up with exactly one mapped whichever of
    `import torch` / loading libslimt_hip.so comes first (capi._preload_hip_runtime)."""
    pytest.importorskip("torch")
    r = _run_child(_CHILD_ORDER, order)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_service_library_exports_every_declared_symbol():
    """include/slimt_hip_service.h (the batching service's C ABI) against libslimt_hip_host.so; without a
    GPU a service cannot be created (its workers need a device), but the failure is an error code."""
    from slimt_amd import build, capi
    path = build.build_host_lib()
    assert os.path.exists(path)
    text = open(os.path.join(ROOT, "include", "slimt_hip_service.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(slimt_hip_(?:service|result)_[a-z0-9_]+)\s*\(", text)))
    assert names == ["slimt_hip_result_destroy", "slimt_hip_result_view", "slimt_hip_service_create",
                     "slimt_hip_service_destroy", "slimt_hip_service_last_error", "slimt_hip_service_translate"]
    dll = capi.host_lib()
    for n in names:
        assert hasattr(dll, n), f"{n} declared in slimt_hip_service.h but not exported"
    out = ctypes.c_void_p()
    assert dll.slimt_hip_service_create(None, None, 0, ctypes.byref(out)) != 0
    assert b"null argument" in dll.slimt_hip_service_last_error()
    assert dll.slimt_hip_result_view(None, None, None, None, None, None, None, None) != 0


def test_no_oracle_in_product():
    """The product path must never import/link the oracle or /root/reference."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slimt_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hh", ".cc")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "slimt_oracle" not in src and "from oracle" not in src, f
                assert "import oracle" not in src, f
                assert "/root/reference" not in src, f


def test_host_side_prepare_functions(oracle):
    from slimt_amd import capi
    r = np.random.Generator(np.random.PCG64(0))
    w = r.normal(0, 0.2, size=(40, 64)).astype(np.float32)
    w[0, :4] = [0.5 / 100, 1.5 / 100, -2.5 / 100, 3.0]  # ties + saturation
    got = capi.prepare_weight_transposed(w, 100.0)
    want = oracle.quantize(w, 100.0)
    assert np.array_equal(got, want)
    q = r.integers(-127, 128, size=(24, 128)).astype(np.int8)
    assert np.array_equal(capi.prepare_weight_quantized_transposed(q, 128, 24), q)


def test_compute_fails_loudly_without_gpu():
    from slimt_amd import capi, synth
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    x = np.zeros((2, 64), dtype=np.float32)
    W = np.zeros((16, 64), dtype=np.int8)
    with pytest.raises(capi.SlimtHipError):
        capi.affine(x, W, None, 1.0, 1.0)
    m = synth.make_model("micro")
    with pytest.raises(capi.SlimtHipError) as e:
        capi.Model(m)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_bin_roundtrip():
    from slimt_amd import synth
    m = synth.make_model("micro", eos_bias=2.0)
    buf = synth.write_bin(m)
    back = synth.model_from_bin(buf, enc_layers=m.enc_layers, dec_layers=m.dec_layers)
    assert back.D == m.D and back.F == m.F and back.V == m.V
    assert set(back.params) == set(m.params)
    for k, p in m.params.items():
        q = back.params[k]
        assert (q.rows, q.cols, q.kind) == (p.rows, p.cols, p.kind)
        assert np.array_equal(np.asarray(q.data).reshape(-1), np.asarray(p.data).reshape(-1))
        assert np.float32(q.mult) == np.float32(p.mult)
    # 256-byte aligned payload start (Io.cc:151-153)
    import struct
    n = struct.unpack_from("<Q", buf, 8)[0]
    assert n == len(m.params) + 1


def test_loader_rejects_malformed_bins(tmp_path):
    """io::load_items (host/Io.cc; slimt/Io.cc:114-161) must refuse a file whose payloads do
    not cover their shapes instead of letting the uploads read past the mapping."""
    import struct
    import subprocess
    from slimt_amd import build, synth
    exe = build.build_host()
    m = synth.make_model("micro", eos_bias=0.0)
    good = synth.write_bin(m)

    def run(blob):
        f = tmp_path / "m.bin"
        f.write_bytes(blob)
        return subprocess.run([exe, "--load", str(f)], capture_output=True, text=True, timeout=60)

    r = run(good)
    assert r.returncode == 0 and "loaded" in r.stdout, r.stderr
    r = run(good[: len(good) // 2])
    assert r.returncode == 1 and "truncated" in r.stderr
    # an f32 item whose shape claims more elements than its payload holds
    n = struct.unpack_from("<Q", good, 8)[0]
    headers = [list(struct.unpack_from("<QQQQ", good, 16 + 32 * i)) for i in range(n)]
    names_off = 16 + 32 * n
    shapes_off = names_off + sum(h[0] for h in headers)
    off = shapes_off
    victim = None
    for i, h in enumerate(headers):
        if h[1] == synth.TYPE_F32 and victim is None:
            victim = (i, off)
        off += 4 * h[2]
    i, soff = victim
    bad = bytearray(good)
    dims = list(struct.unpack_from("<%di" % headers[i][2], good, soff))
    dims[-1] *= 3
    struct.pack_into("<%di" % headers[i][2], bad, soff, *dims)
    r = run(bytes(bad))
    assert r.returncode == 1 and "shorter than its shape" in r.stderr, r.stderr
    bad = bytearray(good)
    dims[-1] = -4
    struct.pack_into("<%di" % headers[i][2], bad, soff, *dims)
    r = run(bytes(bad))
    assert r.returncode == 1 and "non-positive" in r.stderr, r.stderr


def test_model_create_from_bin_parses_before_it_needs_a_device():
    """slimt_hip_model_create_from_bin (the `View model` of Transformer::Transformer, Transformer.cc:87-94):
    a malformed container is refused by the parser; a good one gets as far as the device check."""
    import struct
    from slimt_amd import capi, synth
    m = synth.make_model("micro", eos_bias=0.0)
    good = synth.write_bin(m)
    for blob, why in ((good[:40], "truncated"), (good[: len(good) // 2], "truncated"),
                      (struct.pack("<Q", 2) + good[8:], "version"),
                      (good[:8] + struct.pack("<Q", 1 << 30) + good[16:], "implausible")):
        with pytest.raises(capi.SlimtHipError) as e:
            capi.Model.from_bin(blob, m.enc_layers, m.dec_layers, m.H)
        assert why in str(e.value), str(e.value)
    if capi.device_count() > 0:
        pytest.skip("a GPU is present: the good container loads (tests/test_gpu_engine.py covers it)")
    with pytest.raises(capi.SlimtHipError) as e:
        capi.Model.from_bin(good, m.enc_layers, m.dec_layers, m.H)
    assert "no HIP device" in str(e.value) or "device" in str(e.value).lower()  # past the parser: the device check
