"""The drop-in in the reference's own form (integration/): the Provider::Hip specialisations of
slimt/QMM.hh:24-44, the engine hooks and the two patches that wire them into a slimt checkout
must keep matching the reference. CPU only, and only where the reference checkout exists (this
container): its files are copied to a temporary directory, patched and compiled there -- nothing
of the reference is committed or travels to the GPU box, where this file skips.

What is checked, and what cannot be:
  * both patches apply cleanly to the reference's files (every hunk's context matches);
  * slimt/QMM.cc, patched, compiles with the reference's own headers + include/slimt_hip.h and
    defines the five slimt::qmm::* functions and the five detail::*<Provider::Hip> specialisations;
  * slimt/hip/Engine.cc (Model::forward, the model / shortlist constructors on the device) compiles
    against the reference's own Input.hh / Types.hh / Tensor.hh, and the patched Transformer.hh
    parses;
  * both objects call only slimt_hip_* symbols that include/slimt_hip.h declares and the built
    library exports;
  * the patched Transformer.cc / Model.cc / Shortlist.cc call only functions Engine.hh declares.
Not compiled: the patched Transformer.cc, Model.{hh,cc} and Shortlist.{hh,cc} -- they include
slimt/Vocabulary.hh, which needs sentencepiece_processor.h (un-vendored, absent from this image;
writing a stand-in is not allowed). Objects are compiled, not linked: slimt/Tensor.cc pulls in
TensorOps.cc, which needs cblas.h or ruy."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
PATCHES = ["0001-qmm-provider-hip.patch", "0002-engine-hooks.patch"]

pytestmark = pytest.mark.skipif(
    not os.path.isdir(os.path.join(REFERENCE, "slimt")) or shutil.which("patch") is None or shutil.which("g++") is None,
    reason="needs the reference checkout (this container only), patch and g++")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "slimt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(slimt_hip_[a-z0-9_]+)\s*\(", text))


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    """A scratch copy of the reference's slimt/ + top-level CMakeLists.txt with both patches applied and
    integration/slimt/ copied in."""
    work = tmp_path_factory.mktemp("slimt_with_hip")
    shutil.copytree(os.path.join(REFERENCE, "slimt"), work / "slimt")
    shutil.copy(os.path.join(REFERENCE, "CMakeLists.txt"), work / "CMakeLists.txt")
    for name in PATCHES:
        with open(os.path.join(ROOT, "integration", "patches", name)) as f:
            r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "--fuzz=0"], stdin=f, cwd=work,
                               capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, f"{name} no longer applies to the reference:\n{r.stdout}{r.stderr}"
        assert "offset" not in r.stdout and "fuzz" not in r.stdout, r.stdout  # hunks sit exactly where they were cut
    shutil.copytree(os.path.join(ROOT, "integration", "slimt"), work / "slimt", dirs_exist_ok=True)
    return work


def compile_object(work, source, out):
    r = subprocess.run(["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-DSLIMT_HAS_HIP", "-I", str(work), "-I",
                        os.path.join(ROOT, "include"), "-c", str(work / source), "-o", str(work / out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    nm = subprocess.run(["nm", "-C", str(work / out)], capture_output=True, text=True, timeout=60)
    assert nm.returncode == 0
    defined, undefined = set(), set()
    for line in nm.stdout.splitlines():
        m = re.match(r"^(?:[0-9a-f]+)?\s+([A-Za-z])\s+(.*)$", line)
        if m:
            (undefined if m.group(1) == "U" else defined).add(m.group(2))
    return defined, undefined


def without_hip(text):
    """The file as the preprocessor sees it without SLIMT_HAS_HIP (the hooks nest nothing)."""
    out, state = [], "code"
    for line in text.splitlines():
        if state == "code" and line.strip() == "#ifdef SLIMT_HAS_HIP":
            state = "hip"
        elif state == "hip" and line.strip() == "#else":
            state = "else"
        elif state in ("hip", "else") and line.strip() == "#endif":
            state = "code"
        elif state != "hip":
            out.append(line)
    assert state == "code"
    return "\n".join(out)


def test_patches_touch_only_the_files_they_name(patched):
    touched = set()
    for name in PATCHES:
        touched |= set(re.findall(r"^\+\+\+ b/(\S+)", open(os.path.join(ROOT, "integration", "patches", name)).read(), flags=re.M))
    assert touched == {"slimt/QMM.hh", "slimt/QMM.cc", "CMakeLists.txt", "slimt/CMakeLists.txt", "slimt/Transformer.hh",
                       "slimt/Transformer.cc", "slimt/Model.hh", "slimt/Model.cc", "slimt/Shortlist.hh", "slimt/Shortlist.cc"}
    cm = (patched / "CMakeLists.txt").read_text()
    assert "option(WITH_HIP" in cm and "SLIMT_HAS_HIP" in cm and "cmake/SlimtHip.cmake" in cm
    assert "hip/Engine.cc" in (patched / "slimt" / "CMakeLists.txt").read_text()
    # every added line of the class-level hooks sits under SLIMT_HAS_HIP: other providers build as before
    for name in ("Transformer.hh", "Transformer.cc", "Model.hh", "Model.cc", "Shortlist.hh", "Shortlist.cc"):
        before = open(os.path.join(REFERENCE, "slimt", name)).read()
        after = (patched / "slimt" / name).read_text()
        assert re.sub(r"\s+", "", without_hip(after)) == re.sub(r"\s+", "", before), name


def test_provider_hip_compiles_against_the_reference_headers(patched):
    defined, undefined = compile_object(patched, "slimt/QMM.cc", "qmm.o")
    public = [s for s in defined if re.match(r"slimt::qmm::(affine|affine_with_select|dot|prepare_weight_transposed|"
                                             r"prepare_weight_quantized_transposed)\(", s)]
    assert len(public) == 5, sorted(public)
    # Provider::Hip is the enum's fifth value: the specialisations the public functions forward to
    special = [s for s in defined if "slimt::qmm::detail::" in s and "<(slimt::qmm::detail::Provider)4>" in s]
    assert len(special) == 5, sorted(special)
    called = {s for s in undefined if s.startswith("slimt_hip_")}
    assert called == {"slimt_hip_affine", "slimt_hip_affine_select", "slimt_hip_prepare_weight_transposed",
                      "slimt_hip_prepare_weight_quantized_transposed", "slimt_hip_last_error"}
    assert called <= declared_symbols()


def test_engine_hooks_compile_against_the_reference_headers(patched):
    defined, undefined = compile_object(patched, "slimt/hip/Engine.cc", "engine.o")
    for fn in ("create_model", "create_shortlist", "generate", "forward"):
        assert any(s.startswith(f"slimt::hip::{fn}(") for s in defined), fn
    called = {s for s in undefined if s.startswith("slimt_hip_")}
    assert {"slimt_hip_model_create_from_bin", "slimt_hip_translate", "slimt_hip_translate_generated",
            "slimt_hip_shortlist_create", "slimt_hip_shortlist_generate", "slimt_hip_ctx_create_budget"} <= called
    assert called <= declared_symbols()
    # ... and the built library exports every one of them
    import ctypes
    from slimt_amd import build
    dll = ctypes.CDLL(build.build())
    for s in called:
        assert hasattr(dll, s), s
    # the patched class declaration that needs no third-party header parses with its new member
    r = subprocess.run(["g++", "-std=c++20", "-DSLIMT_HAS_HIP", "-I", str(patched), "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", "-x", "c++", str(patched / "slimt" / "Transformer.hh")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    # without the definition the same files are the reference's own again (the header still parses)
    r = subprocess.run(["g++", "-std=c++20", "-I", str(patched), "-fsyntax-only", "-x", "c++",
                        str(patched / "slimt" / "Transformer.hh")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_call_sites_use_what_engine_hh_declares(patched):
    header = (patched / "slimt" / "hip" / "Engine.hh").read_text()
    declared = set(re.findall(r"^\w[\w:<>\* ]*?\b(\w+)\(", header, flags=re.M)) | set(re.findall(r"using (\w+) =", header))
    used = set()
    for name in ("Transformer.hh", "Transformer.cc", "Model.hh", "Model.cc", "Shortlist.hh", "Shortlist.cc"):
        used |= set(re.findall(r"\bhip::(\w+)", (patched / "slimt" / name).read_text()))
    assert used == {"create_model", "create_shortlist", "generate", "forward", "ModelHandle", "ShortlistHandle"}
    assert used <= declared, (used, declared)
