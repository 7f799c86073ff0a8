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
    declared = (set(re.findall(r"^\w[\w:<>\*& ]*?\b(\w+)\(", header, flags=re.M)) | set(re.findall(r"using (\w+) =", header))
                | set(re.findall(r"^class (\w+) \{", header, flags=re.M)))
    used = set()
    for name in ("Transformer.hh", "Transformer.cc", "Model.hh", "Model.cc", "Shortlist.hh", "Shortlist.cc"):
        used |= set(re.findall(r"\bhip::(\w+)", (patched / "slimt" / name).read_text()))
    assert used == {"create_model", "create_shortlist", "generate", "forward", "ModelHandle", "ShortlistHandle"}
    assert used <= declared, (used, declared)


def test_contexts_die_with_their_model_not_with_the_thread(patched, tmp_path):
    """the contexts Model::forward runs on point into the model, so the handle that owns the model
    owns them too. A recording double of the C ABI (test code: counts and orders the calls, computes nothing)
    is linked with the real Engine.cc: a worker thread that outlives the model must destroy nothing afterwards,
    every context must go before its model, and concurrent callers end with one context each."""
    double = tmp_path / "abi_double.cc"
    double.write_text(r"""
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "slimt/hip/Engine.hh"
struct slimt_hip_model { std::atomic<int> live_ctx{0}; bool alive = true; };
struct slimt_hip_ctx { slimt_hip_model* model; };
static std::atomic<int> g_ctx_built{0}, g_ctx_destroyed{0}, g_bad{0};
extern "C" {
const char* slimt_hip_last_error(void) { return ""; }
int slimt_hip_model_create_from_bin(const void*, size_t, const slimt_hip_dims*, int, slimt_hip_model** out) { *out = new slimt_hip_model; return 0; }
int slimt_hip_model_destroy(slimt_hip_model* m) { if (!m) return 0; if (m->live_ctx.load() != 0) g_bad++; m->alive = false; return 0; }
int slimt_hip_ctx_create_budget(slimt_hip_model* m, size_t, size_t, size_t, void*, slimt_hip_ctx** out) {
  if (!m->alive) g_bad++; m->live_ctx++; g_ctx_built++; *out = new slimt_hip_ctx{m}; return 0; }
int slimt_hip_ctx_destroy(slimt_hip_ctx* c) { if (!c) return 0; if (!c->model->alive) g_bad++; c->model->live_ctx--; g_ctx_destroyed++; delete c; return 0; }
int slimt_hip_translate(slimt_hip_ctx* c, const uint32_t*, const uint32_t*, size_t B, size_t, const uint32_t*, size_t, float, uint32_t,
                        uint32_t*, uint32_t* out_len, float*) { if (!c->model->alive) g_bad++; for (size_t b = 0; b < B; ++b) out_len[b] = 0; return 0; }
int slimt_hip_translate_generated(slimt_hip_ctx*, slimt_hip_shortlist*, const uint32_t*, const uint32_t*, size_t, size_t, float, uint32_t,
                                  uint32_t*, uint32_t*, float*) { return -1; }
int slimt_hip_shortlist_create(const void*, size_t, size_t, size_t, int, int, int, slimt_hip_shortlist**) { return -1; }
int slimt_hip_shortlist_destroy(slimt_hip_shortlist*) { return 0; }
int slimt_hip_shortlist_generate(slimt_hip_shortlist*, const uint32_t*, const uint32_t*, size_t, size_t, uint32_t*, size_t*) { return -1; }
}
int main() {
  using namespace slimt;
  std::atomic<bool> go_on{true};
  std::atomic<int> ready{0};
  {
    hip::ModelHandle model = hip::create_model(View{nullptr, 0}, 6, 2, 8);
    std::vector<std::thread> workers;
    for (int w = 0; w < 4; ++w)
      workers.emplace_back([&] {
        auto lease = model.acquire(8, 16);  // four concurrent callers
        ready++;
        while (ready.load() < 4) std::this_thread::yield();
        model.release(lease);
        auto again = model.acquire(16, 16);  // a larger batch: the pooled context is rebuilt, not leaked
        model.release(again);
      });
    for (auto& t : workers) t.join();
    // a thread that is still alive when the model goes
    std::thread late([&] { while (go_on.load()) std::this_thread::yield(); });
    hip::ModelHandle moved = std::move(model);
    if (model || !moved) g_bad++;
    {
      hip::ModelHandle gone = std::move(moved);
    }  // model destroyed here, with every pooled context before it
    if (g_ctx_built.load() != g_ctx_destroyed.load()) g_bad++;
    go_on = false;
    late.join();
  }
  std::printf("built %d destroyed %d bad %d\n", g_ctx_built.load(), g_ctx_destroyed.load(), g_bad.load());
  return g_bad.load() == 0 && g_ctx_built.load() >= 4 ? 0 : 1;
}
""")
    # Engine.cc's forward() also references Input / Tensor (slimt/Tensor.cc pulls in TensorOps.cc: cblas or ruy, absent):
    # nothing of that is called here, so those references stay unresolved at link time
    r = subprocess.run(["g++", "-std=c++20", "-O1", "-DSLIMT_HAS_HIP", "-pthread", "-I", str(patched), "-I",
                        os.path.join(ROOT, "include"), str(patched / "slimt" / "hip" / "Engine.cc"), str(double),
                        "-Wl,--unresolved-symbols=ignore-all", "-o", str(tmp_path / "lifetime")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    run = subprocess.run([str(tmp_path / "lifetime")], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, run.stdout + run.stderr
