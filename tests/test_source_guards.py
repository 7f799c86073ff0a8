"""Guards against traps that have bitten this code base, checked on the sources (CPU, no compiler).

1. `__builtin_bit_cast(T, v.y)` of a vector ELEMENT reads element 0 with this ROCm's clang (it bit twice: round 3
   in the decoder's epilogue constants, round 4 in an experiment -- every tiny11 test failed until the element was
   copied first). A cast of a whole vector, or of a scalar copy, is fine.
2. The context-count guard: the library counts the contexts a process holds per device (past 22 the hardware queues
   are time-sliced) -- on a box without a GPU the count is simply 0 and out-of-range devices are refused.
"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# an element access as the LAST thing of the operand: .x/.y/.z/.w, .s0-.sf, .lo/.hi/.even/.odd, or [i] on a name that was
# declared as an ext_vector_type in the same file
_ELEMENT = re.compile(r"__builtin_bit_cast\s*\(\s*[\w: ]+?,\s*([^()]*?(?:\.(?:[xyzw]{1,4}|s[0-9a-fA-F]|lo|hi|even|odd)))\s*\)")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def device_sources():
    pats = ["slimt_amd/csrc/*.hip", "slimt_amd/csrc/*.h", "slimt_amd/csrc/*.cpp", "tools/probes/*.hip"]
    return sorted(f for p in pats for f in glob.glob(os.path.join(ROOT, p)))


def test_the_pattern_itself_catches_the_trap():
    assert _ELEMENT.search("const float c = __builtin_bit_cast(float, q.y);")
    assert _ELEMENT.search("x = __builtin_bit_cast( float , frag.cp4 . w );".replace(" . ", "."))
    assert _ELEMENT.search("__builtin_bit_cast(int, v.s3)")
    assert not _ELEMENT.search("__builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, 0, 0))")
    assert not _ELEMENT.search("const int y = q.y; f = __builtin_bit_cast(float, y);")


def test_no_bit_cast_of_a_vector_element_in_device_code():
    files = device_sources()
    assert len(files) >= 10
    hits = []
    for f in files:
        text = _strip_comments(open(f).read())
        for m in _ELEMENT.finditer(text):
            line = text.count("\n", 0, m.start()) + 1
            hits.append(f"{os.path.relpath(f, ROOT)}:{line}: {m.group(0)}")
    assert not hits, ("__builtin_bit_cast of a vector element reads element 0 with this compiler: copy the element to a "
                      "scalar first\n" + "\n".join(hits))


def test_context_count_without_a_device():
    from slimt_amd import capi
    assert capi.contexts_on_device(0) == 0
    L = capi.lib()
    import ctypes
    n = ctypes.c_int(-5)
    assert L.slimt_hip_contexts_on_device(-1, ctypes.byref(n)) < 0
    assert L.slimt_hip_contexts_on_device(1 << 20, ctypes.byref(n)) < 0
    assert L.slimt_hip_contexts_on_device(0, None) < 0
