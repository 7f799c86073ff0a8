"""ctypes binding of include/slimt_hip.h (slimt_amd/lib/libslimt_hip.so).

Thin: numpy arrays in/out, every call goes through the C ABI exactly as a
cgo/JNI/C++ host would. There is no fallback: if the library is missing or a
call fails, SlimtHipError is raised.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
from typing import Optional, Sequence

import numpy as np

from . import build as _build

SYMBOLS = [
    "slimt_hip_abi_version", "slimt_hip_last_error", "slimt_hip_device_count",
    "slimt_hip_set_device", "slimt_hip_affine", "slimt_hip_affine_select",
    "slimt_hip_affine_acc_i32", "slimt_hip_prepare_weight_transposed",
    "slimt_hip_prepare_weight_quantized_transposed", "slimt_hip_layer_norm",
    "slimt_hip_softmax", "slimt_hip_highway", "slimt_hip_sdpa",
    "slimt_hip_request_hw_queues", "slimt_hip_hw_queues", "slimt_hip_model_create_from_bin",
    "slimt_hip_model_create", "slimt_hip_model_destroy", "slimt_hip_model_info",
    "slimt_hip_ctx_create", "slimt_hip_ctx_create_budget", "slimt_hip_ctx_destroy", "slimt_hip_contexts_on_device", "slimt_hip_ctx_stream",
    "slimt_hip_ctx_synchronize", "slimt_hip_ctx_set_decode_mode", "slimt_hip_ctx_set_encode_rows", "slimt_hip_ctx_plan", "slimt_hip_translate", "slimt_hip_translate_device",
    "slimt_hip_encode", "slimt_hip_decode_begin", "slimt_hip_decode_step",
    "slimt_hip_profile_enable", "slimt_hip_profile_read", "slimt_hip_profile_reset",
    "slimt_hip_debug_decode_stamps", "slimt_hip_debug_kv_formats", "slimt_hip_debug_kv_narrow_limit", "slimt_hip_debug_kv_tight_limit", "slimt_hip_debug_kv_tight_watch", "slimt_hip_debug_kv_centres", "slimt_hip_model_set_kv_centres", "slimt_hip_debug_kv_watch", "slimt_hip_debug_break_shortlist_handoff", "slimt_hip_debug_cross_attention",
    "slimt_hip_debug_occupancy_trace", "slimt_hip_model_set_decoder_budget",
    "slimt_hip_model_set_kv_cache_policy",
    "slimt_hip_model_set_xcd_affinity", "slimt_hip_model_device",
    "slimt_hip_model_set_kv_cache_format", "slimt_hip_model_set_adaptive_decoder_rows",
    "slimt_hip_shortlist_create", "slimt_hip_shortlist_destroy", "slimt_hip_shortlist_info",
    "slimt_hip_shortlist_generate", "slimt_hip_shortlist_generate_device",
    "slimt_hip_translate_device_generated", "slimt_hip_translate_generated", "slimt_hip_translate_async_generated",
    "slimt_hip_translate_async", "slimt_hip_host_alloc", "slimt_hip_host_free",
    "slimt_hip_encode_embedded", "slimt_hip_decode_begin_from", "slimt_hip_decode_step_states",
    "slimt_hip_translate_many_rows", "slimt_hip_translate_many_device", "slimt_hip_translate_many_async",
    "slimt_hip_debug_kv_recalibrations", "slimt_hip_translate_many_device_generated", "slimt_hip_translate_many_async_generated",
]

K_NONE, K_GEMM_ENC, K_GEMM_DEC, K_LOGITS, K_ATTN_ENC, K_ATTN_DEC, K_SSRU, K_DECODE_FUSED, K_ENCODE_FUSED = range(9)
KERNEL_NAMES = {K_GEMM_ENC: "gemm_enc", K_GEMM_DEC: "gemm_dec", K_LOGITS: "logits_argmax",
                K_ATTN_ENC: "attn_enc", K_ATTN_DEC: "attn_dec", K_SSRU: "ssru"}


class SlimtHipError(RuntimeError):
    pass


class _Param(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("rows", C.c_int32),
                ("cols", C.c_int32), ("data", C.c_void_p), ("bytes", C.c_uint64)]


class _Batch(C.Structure):
    """slimt_hip_batch (include/slimt_hip.h): one batch of a merged launch."""
    _fields_ = [("src_ids", C.c_void_p), ("lengths", C.c_void_p), ("B", C.c_size_t), ("S", C.c_size_t), ("shortlist", C.c_void_p),
                ("n_shortlist", C.c_size_t), ("out_ids", C.c_void_p), ("out_len", C.c_void_p), ("align", C.c_void_p)]


class _Dims(C.Structure):
    _fields_ = [("encoder_layers", C.c_int32), ("decoder_layers", C.c_int32),
                ("num_heads", C.c_int32)]


_lib = None


def library_path() -> str:
    return _build.LIB_PATH


def _elf_soname(path: str):
    """DT_SONAME of a 64-bit little-endian ELF shared object (None when it has none or is not one)."""
    import struct
    try:
        with open(path, "rb") as f:
            head = f.read(64)
            if len(head) < 64 or head[:4] != b"\x7fELF" or head[4] != 2 or head[5] != 1:
                return None
            e_shoff, = struct.unpack_from("<Q", head, 0x28)
            e_shentsize, e_shnum = struct.unpack_from("<HH", head, 0x3A)
            sections = []
            for i in range(e_shnum):
                f.seek(e_shoff + i * e_shentsize)
                sh = f.read(e_shentsize)
                sh_type, = struct.unpack_from("<I", sh, 4)
                sh_offset, sh_size, sh_link = struct.unpack_from("<QQI", sh, 0x18)
                sections.append((sh_type, sh_offset, sh_size, sh_link))
            for sh_type, off, size, link in sections:
                if sh_type != 6:  # SHT_DYNAMIC
                    continue
                f.seek(off)
                dyn = f.read(size)
                for j in range(0, len(dyn) - 15, 16):
                    tag, val = struct.unpack_from("<qQ", dyn, j)
                    if tag == 14:  # DT_SONAME: offset into the linked string table
                        f.seek(sections[link][1] + val)
                        return f.read(256).split(b"\0", 1)[0].decode()
                    if tag == 0:
                        break
    except (OSError, struct.error, IndexError, UnicodeDecodeError):
        return None
    return None


def mapped_hip_runtimes():
    """Paths of the libamdhip64 images this process has mapped (one, unless something loaded a second runtime)."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        return []


HIP_RUNTIME_SONAME = "libamdhip64.so.7"  # what libslimt_hip.so NEEDs


def _preload_hip_runtime() -> None:
    """One HIP runtime per process, whatever the import order. libslimt_hip.so NEEDs `libamdhip64.so.7` by
    SONAME: the dynamic loader binds that to a copy the process has mapped already (PyTorch imported first: its
    bundled one), else to ROCm's through the RUNPATH. PyTorch's own libraries ask for theirs by FILE, so with
    ROCm's copy mapped first a later `import torch` maps a second runtime next to it (and finds no GPU). When no
    runtime is mapped yet and an installed PyTorch bundles one WITH THAT SONAME (a PyTorch built against another ROCm
    major would be a second runtime, the very thing this avoids), map that one now: a later `import torch` then finds
    its own file already loaded, and both sides share it. Nothing is imported, no HIP call is made.
    SLIMT_HIP_NO_PRELOAD=1 switches this off (processes that never import torch)."""
    if os.environ.get("SLIMT_HIP_NO_PRELOAD") == "1" or mapped_hip_runtimes():
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for d in (spec.submodule_search_locations or []) if spec else []:
        cand = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(cand) and _elf_soname(cand) == HIP_RUNTIME_SONAME:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return


def lib():
    """Load libslimt_hip.so (must have been built: __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise SlimtHipError(
            f"{path} is missing: build it with `python -m slimt_amd.build` "
            "(there is no CPU fallback)")
    _preload_hip_runtime()
    L = C.CDLL(path)
    if os.environ.get("SLIMT_HIP_LIB"):
        # an alternative library (A/B of builds, tools/ab_lib.sh): an OLDER one may lack the newest entry points -- they
        # then fail when called, not when the library is loaded
        class _Tolerant:
            def __init__(self, lib_):
                object.__setattr__(self, "_l", lib_)

            def __getattr__(self, name):
                try:
                    return getattr(self._l, name)
                except AttributeError:
                    def missing(*a, **k):
                        raise SlimtHipError(f"{name} is not exported by {path}")
                    missing.argtypes = missing.restype = None
                    object.__setattr__(self, name, missing)
                    return missing
        L = _Tolerant(L)
    if len(mapped_hip_runtimes()) > 1:  # (someone mapped another copy by file name before or behind us)
        import warnings
        warnings.warn("two HIP runtimes are mapped in this process (%s): the GPU may be invisible to one of them"
                      % ", ".join(mapped_hip_runtimes()), RuntimeWarning, stacklevel=2)
    vp, f32, sz, i32, u32 = C.c_void_p, C.c_float, C.c_size_t, C.c_int, C.c_uint32
    L.slimt_hip_abi_version.restype = i32
    L.slimt_hip_last_error.restype = C.c_char_p
    L.slimt_hip_device_count.argtypes = [vp]
    L.slimt_hip_set_device.argtypes = [i32]
    L.slimt_hip_affine.argtypes = [vp, sz, sz, vp, sz, vp, f32, f32, vp]
    L.slimt_hip_affine_select.argtypes = [vp, sz, sz, vp, sz, vp, f32, f32, vp, sz, vp]
    L.slimt_hip_affine_acc_i32.argtypes = [vp, sz, sz, vp, sz, f32, vp]
    L.slimt_hip_prepare_weight_transposed.argtypes = [vp, vp, f32, sz, sz]
    L.slimt_hip_prepare_weight_quantized_transposed.argtypes = [vp, vp, sz, sz]
    L.slimt_hip_layer_norm.argtypes = [vp, vp, vp, f32, sz, sz, vp]
    L.slimt_hip_softmax.argtypes = [vp, sz, sz, vp]
    L.slimt_hip_highway.argtypes = [vp, vp, vp, sz, vp]
    L.slimt_hip_sdpa.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz, sz, vp, vp]
    L.slimt_hip_model_create.argtypes = [vp, sz, vp, i32, vp]
    L.slimt_hip_model_create_from_bin.argtypes = [vp, sz, vp, i32, vp]
    L.slimt_hip_request_hw_queues.argtypes = [i32]
    L.slimt_hip_hw_queues.restype = i32
    L.slimt_hip_model_destroy.argtypes = [vp]
    L.slimt_hip_model_info.argtypes = [vp, vp, vp, vp, vp]
    L.slimt_hip_ctx_create.argtypes = [vp, sz, sz, vp, vp]
    L.slimt_hip_ctx_destroy.argtypes = [vp]
    L.slimt_hip_contexts_on_device.argtypes = [i32, vp]
    L.slimt_hip_ctx_stream.argtypes = [vp, vp]
    L.slimt_hip_ctx_synchronize.argtypes = [vp]
    L.slimt_hip_ctx_set_decode_mode.argtypes = [vp, i32]
    L.slimt_hip_ctx_set_encode_rows.argtypes = [vp, i32]
    L.slimt_hip_ctx_plan.argtypes = [vp, sz, vp, vp]
    L.slimt_hip_translate.argtypes = [vp, vp, vp, sz, sz, vp, sz, f32, u32, vp, vp, vp]
    L.slimt_hip_translate_async.argtypes = [vp, vp, vp, sz, sz, vp, sz, f32, u32, vp, vp, vp]
    L.slimt_hip_host_alloc.argtypes = [sz, vp]
    L.slimt_hip_host_free.argtypes = [vp]
    L.slimt_hip_translate_device.argtypes = [vp, vp, vp, sz, sz, vp, sz, f32, u32, vp, vp, vp, i32]
    L.slimt_hip_encode.argtypes = [vp, vp, vp, sz, sz, vp, vp, vp]
    L.slimt_hip_decode_begin.argtypes = [vp, vp, sz]
    L.slimt_hip_encode_embedded.argtypes = [vp, vp, vp, sz, sz, vp]
    L.slimt_hip_decode_begin_from.argtypes = [vp, vp, vp, sz, sz, vp, sz]
    L.slimt_hip_decode_step_states.argtypes = [vp, vp, vp, vp, vp, vp]
    L.slimt_hip_decode_step.argtypes = [vp, vp, vp, vp, vp]
    L.slimt_hip_profile_enable.argtypes = [vp, i32]
    L.slimt_hip_profile_read.argtypes = [vp, vp, vp, vp, vp]
    L.slimt_hip_profile_reset.argtypes = [vp]
    L.slimt_hip_debug_decode_stamps.argtypes = [vp, i32, vp, sz]
    L.slimt_hip_debug_occupancy_trace.argtypes = [vp, sz]
    L.slimt_hip_debug_kv_formats.argtypes = [vp, vp, sz, vp]
    L.slimt_hip_debug_kv_narrow_limit.argtypes = [vp, i32]
    L.slimt_hip_debug_kv_watch.argtypes = [vp, vp, vp, vp]
    L.slimt_hip_debug_kv_tight_limit.argtypes = [vp, i32]
    L.slimt_hip_debug_kv_tight_watch.argtypes = [vp, vp, vp, vp]
    L.slimt_hip_debug_kv_recalibrations.argtypes = [vp, vp, i32]
    L.slimt_hip_debug_kv_centres.argtypes = [vp, vp, sz, vp]
    L.slimt_hip_model_set_kv_centres.argtypes = [vp, vp, sz]
    L.slimt_hip_debug_break_shortlist_handoff.argtypes = [vp, i32, u32]
    L.slimt_hip_debug_cross_attention.argtypes = [vp, i32, i32, vp, vp, vp]
    L.slimt_hip_model_set_decoder_budget.argtypes = [vp, i32]
    L.slimt_hip_model_set_kv_cache_policy.argtypes = [vp, i32]
    L.slimt_hip_model_set_xcd_affinity.argtypes = [vp, i32]
    L.slimt_hip_model_set_kv_cache_format.argtypes = [vp, i32]
    L.slimt_hip_model_set_adaptive_decoder_rows.argtypes = [vp, i32]
    L.slimt_hip_ctx_create_budget.argtypes = [vp, sz, sz, sz, vp, vp]
    L.slimt_hip_shortlist_create.argtypes = [vp, sz, sz, sz, i32, i32, i32, vp]
    L.slimt_hip_shortlist_destroy.argtypes = [vp]
    L.slimt_hip_shortlist_info.argtypes = [vp, vp, vp]
    L.slimt_hip_shortlist_generate.argtypes = [vp, vp, vp, sz, sz, vp, vp]
    L.slimt_hip_shortlist_generate_device.argtypes = [vp, vp, vp, vp, sz, sz, vp, vp]
    L.slimt_hip_translate_device_generated.argtypes = [vp, vp, vp, vp, sz, sz, f32, u32, vp, vp, vp, i32]
    L.slimt_hip_translate_generated.argtypes = [vp, vp, vp, vp, sz, sz, f32, u32, vp, vp, vp]
    L.slimt_hip_translate_async_generated.argtypes = [vp, vp, vp, vp, sz, sz, f32, u32, vp, vp, vp]
    L.slimt_hip_translate_many_rows.argtypes = [vp, sz]
    L.slimt_hip_translate_many_rows.restype = sz
    L.slimt_hip_translate_many_device.argtypes = [vp, vp, sz, sz, f32, u32, i32]
    L.slimt_hip_translate_many_async.argtypes = [vp, vp, sz, sz, f32, u32]
    L.slimt_hip_translate_many_device_generated.argtypes = [vp, vp, vp, sz, sz, f32, u32, i32]
    L.slimt_hip_translate_many_async_generated.argtypes = [vp, vp, vp, sz, sz, f32, u32]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("slimt_hip_abi_version",):
            fn.restype = C.c_int
    _lib = L
    return L


def translate_many_rows(sizes) -> int:
    """Global sentences a merged launch of these batch sizes occupies in its context (max_batch)."""
    a = (C.c_size_t * len(sizes))(*[int(x) for x in sizes])
    return int(lib().slimt_hip_translate_many_rows(a, len(sizes)))


def contexts_on_device(device: int = 0) -> int:
    """slimt_hip_contexts_on_device: contexts (streams) this process holds on `device`; past 22 the device's hardware
    queues are time-sliced and the library says so once on stderr."""
    n = C.c_int(0)
    if lib().slimt_hip_contexts_on_device(int(device), C.byref(n)):
        raise SlimtHipError(lib().slimt_hip_last_error().decode())
    return int(n.value)


def request_hw_queues(n: int = 32) -> bool:
    """slimt_hip_request_hw_queues: ask the HIP runtime for `n` hardware queues (GPU_MAX_HW_QUEUES, unless the
    process has set it) -- one per concurrent translate worker's stream; the default of four makes 20 workers
    take turns. Only before the library's first HIP call: returns False when it came too late."""
    rc = lib().slimt_hip_request_hw_queues(int(n))
    if rc < 0:
        raise SlimtHipError(lib().slimt_hip_last_error().decode())
    return rc == 0


def _chk(rc: int) -> None:
    if rc != 0:
        raise SlimtHipError(f"slimt_hip error {rc}: {lib().slimt_hip_last_error().decode()}")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().slimt_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


# ---- op level (slimt::qmm / TensorOps) ---------------------------------------

def affine(x, W_nk, bias, a_quant: float, b_quant: float):
    """qmm::affine; bias=None gives qmm::dot. W_nk: int8 [N][K]."""
    x = _f32(x)
    W = np.ascontiguousarray(W_nk, dtype=np.int8)
    N, K = W.shape
    M = x.size // K
    b = None if bias is None else _f32(bias).reshape(-1)
    y = np.empty(x.shape[:-1] + (N,), dtype=np.float32)
    _chk(lib().slimt_hip_affine(_p(x), M, K, _p(W), N, _p(b), a_quant, b_quant, _p(y)))
    return y


def dot(x, W_nk, a_quant: float, b_quant: float):
    return affine(x, W_nk, None, a_quant, b_quant)


def affine_with_select(x, W_nk, bias, a_quant, b_quant, indices):
    x = _f32(x)
    W = np.ascontiguousarray(W_nk, dtype=np.int8)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    N, K = W.shape
    M = x.size // K
    b = _f32(bias).reshape(-1)
    y = np.empty(x.shape[:-1] + (idx.size,), dtype=np.float32)
    _chk(lib().slimt_hip_affine_select(_p(x), M, K, _p(W), N, _p(b), a_quant, b_quant, _p(idx),
                                       idx.size, _p(y)))
    return y


def affine_acc_i32(x, W_nk, a_quant: float):
    x = _f32(x)
    W = np.ascontiguousarray(W_nk, dtype=np.int8)
    N, K = W.shape
    M = x.size // K
    acc = np.empty((M, N), dtype=np.int32)
    _chk(lib().slimt_hip_affine_acc_i32(_p(x), M, K, _p(W), N, a_quant, _p(acc)))
    return acc


def prepare_weight_transposed(weights, quant_mult: float):
    w = _f32(weights)
    rows, cols = w.shape
    out = np.empty((rows, cols), dtype=np.int8)
    _chk(lib().slimt_hip_prepare_weight_transposed(_p(w), _p(out), quant_mult, cols, rows))
    return out


def prepare_weight_quantized_transposed(w_nk, rows: int, cols: int):
    w = np.ascontiguousarray(w_nk, dtype=np.int8)
    out = np.empty_like(w)
    _chk(lib().slimt_hip_prepare_weight_quantized_transposed(_p(w), _p(out), rows, cols))
    return out


def layer_norm(x, scale, bias, eps: float = 1e-6):
    x = _f32(x)
    s, b = _f32(scale).reshape(-1), _f32(bias).reshape(-1)
    cols = x.shape[-1]
    y = np.empty_like(x)
    _chk(lib().slimt_hip_layer_norm(_p(x), _p(s), _p(b), eps, x.size // cols, cols, _p(y)))
    return y


def softmax(x):
    x = _f32(x)
    cols = x.shape[-1]
    y = np.empty_like(x)
    _chk(lib().slimt_hip_softmax(_p(x), x.size // cols, cols, _p(y)))
    return y


def highway(x, y, g):
    x, y, g = _f32(x), _f32(y), _f32(g)
    out = np.empty_like(x)
    _chk(lib().slimt_hip_highway(_p(x), _p(y), _p(g), x.size, _p(out)))
    return out


def sdpa(q, k, v, mask, want_attn: bool = True):
    q, k, v, mask = _f32(q), _f32(k), _f32(v), _f32(mask)
    B, H, Tq, dh = q.shape
    S = k.shape[2]
    out = np.empty_like(q)
    attn = np.empty((B, H, Tq, S), dtype=np.float32) if want_attn else None
    _chk(lib().slimt_hip_sdpa(_p(q), _p(k), _p(v), _p(mask), B, H, Tq, S, dh, _p(out), _p(attn)))
    return out, attn


# ---- engine level --------------------------------------------------------------

class Model:
    """Device-resident slimt::Transformer (weights on one GPU)."""

    def __init__(self, model, device: int = 0):
        """model: slimt_amd.synth.Model (or anything with .params/.H/...)."""
        keep = []
        arr = (_Param * len(model.params))()
        for i, p in enumerate(model.params.values()):
            buf = np.frombuffer(p.payload(), dtype=np.uint8).copy()
            name = p.name.encode()
            keep += [buf, name]
            arr[i] = _Param(name, 0 if p.kind == "f32" else 1, p.rows, p.cols,
                            buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        dims = _Dims(model.enc_layers, model.dec_layers, model.H)
        h = C.c_void_p()
        _chk(lib().slimt_hip_model_create(C.cast(arr, C.c_void_p), len(model.params),
                                          C.byref(dims), device, C.byref(h)))
        self.h = h
        self.device = device
        self.D, self.F, self.H, self.V = model.D, model.F, model.H, model.V
        self.Le, self.Ld = model.enc_layers, model.dec_layers

    @classmethod
    def from_bin(cls, blob: bytes, enc_layers: int, dec_layers: int, heads: int, device: int = 0) -> "Model":
        """slimt_hip_model_create_from_bin: the Marian .bin container as Transformer::Transformer receives it
        (Transformer.cc:87-94); the library locates the items itself."""
        self = cls.__new__(cls)
        buf = bytes(blob)
        dims = _Dims(enc_layers, dec_layers, heads)
        h = C.c_void_p()
        _chk(lib().slimt_hip_model_create_from_bin(buf, len(buf), C.byref(dims), device, C.byref(h)))
        self.h = h
        self.device = device
        d, f, v, hh = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _chk(lib().slimt_hip_model_info(h, C.byref(d), C.byref(f), C.byref(v), C.byref(hh)))
        self.D, self.F, self.V, self.H = d.value, f.value, v.value, hh.value
        self.Le, self.Ld = enc_layers, dec_layers
        return self

    def set_kv_cache_policy(self, policy: int):
        """0 = per launch (default), 1 = temporal, 2 = non-temporal K/V cache loads in the decoder."""
        _chk(lib().slimt_hip_model_set_kv_cache_policy(self.h, policy))

    def set_xcd_affinity(self, xcds: int):
        """0 = off (default); 1 / 2 / 4 = a batch's decoder tiles are claimed on that many home XCDs."""
        _chk(lib().slimt_hip_model_set_xcd_affinity(self.h, xcds))

    def set_kv_cache_format(self, fmt: int):
        """0 = packed K/V cache where supported, 20 bits per value where a sentence's accumulators fit and 24 elsewhere
        (default); 1 = always f32; 2 = packed, always 24 bits; 3 = f32 and the reference's literal dequantise-then-attend
        sequence (stage-wise decoder: for checking)."""
        _chk(lib().slimt_hip_model_set_kv_cache_format(self.h, fmt))

    def debug_kv_watch(self):
        """(switched to the 24-bit form for good, sentence-layers that needed it so far, sentence-layers cached so far)."""
        sw, w, t = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        _chk(lib().slimt_hip_debug_kv_watch(self.h, C.byref(sw), C.byref(w), C.byref(t)))
        return bool(sw.value), int(w.value), int(t.value)

    def debug_kv_narrow_limit(self, limit: int):
        """Accumulators must lie in [-limit, limit) for the 20-bit cache form (default and maximum 2**19)."""
        _chk(lib().slimt_hip_debug_kv_narrow_limit(self.h, int(limit)))

    def debug_kv_tight_limit(self, limit: int):
        """Signed accumulators must lie in [-limit, limit) for the 16-bit cache form (default and maximum 2**15; 0 = never tried)."""
        _chk(lib().slimt_hip_debug_kv_tight_limit(self.h, int(limit)))

    def set_kv_centres(self, centres):
        """The 16-bit cache form's per-column centres, int32 [Ld][2][D] (else calibrated from the first large batch)."""
        c = np.ascontiguousarray(centres, dtype=np.int32)
        _chk(lib().slimt_hip_model_set_kv_centres(self.h, c.ctypes.data_as(C.c_void_p), c.size))

    def debug_kv_centres(self, Ld: int, D: int):
        """The centres [Ld][2][D] once they exist, else None."""
        out, ready = np.zeros((Ld, 2, D), dtype=np.int32), C.c_int(0)
        _chk(lib().slimt_hip_debug_kv_centres(self.h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(ready)))
        return out if ready.value else None

    def debug_kv_tight_watch(self):
        """(mask of decoder layers that stopped trying the 16-bit form, missed[4], submitted[4])."""
        off, missed, sub = C.c_uint(0), (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
        _chk(lib().slimt_hip_debug_kv_tight_watch(self.h, C.byref(off), missed, sub))
        return int(off.value), [int(x) for x in missed], [int(x) for x in sub]

    def debug_kv_recalibrations(self, max_recalibrations: int = -1) -> int:
        """Generations of K/V centres started after the first (a layer's watch tripped and the engine re-calibrated instead
        of switching it off); max_recalibrations >= 0 sets how many it may start (default 2)."""
        g = C.c_int(0)
        _chk(lib().slimt_hip_debug_kv_recalibrations(self.h, C.byref(g), max_recalibrations))
        return int(g.value)

    def set_adaptive_decoder_rows(self, on: bool):
        """Decode mode 0: 8 or 4 sentences per decoder workgroup while CUs would idle (default on)."""
        _chk(lib().slimt_hip_model_set_adaptive_decoder_rows(self.h, 1 if on else 0))

    def set_decoder_budget(self, workgroups: int):
        """Admission of persistent decoders: ~`workgroups` decoder workgroups at a time (0 = no limit)."""
        _chk(lib().slimt_hip_model_set_decoder_budget(self.h, workgroups))

    def close(self):
        if getattr(self, "h", None):
            lib().slimt_hip_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Pinned:
    """A growing pinned host buffer (slimt_hip_host_alloc) seen as numpy arrays."""

    def __init__(self):
        self.ptr = C.c_void_p()
        self.nbytes = 0

    def array(self, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if n > self.nbytes:
            self.free()
            want = max(n, 2 * self.nbytes, 4096)
            _chk(lib().slimt_hip_host_alloc(want, C.byref(self.ptr)))
            self.nbytes = want
        raw = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(self.nbytes,))
        return raw[:n].view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            lib().slimt_hip_host_free(self.ptr)
            self.ptr = C.c_void_p()
            self.nbytes = 0


class Context:
    """One worker's stream + workspace (mirrors one slimt Async worker)."""

    def __init__(self, model: Model, max_batch: int, max_source_length: int, stream: int = 0,
                 max_tokens: int = 0):
        """max_tokens > 0: a token-budget workspace (slimt_hip_ctx_create_budget): any batch with
        B <= max_batch, S <= max_source_length and B * S <= max_tokens."""
        self.model = model
        h = C.c_void_p()
        if max_tokens:
            _chk(lib().slimt_hip_ctx_create_budget(model.h, max_batch, max_source_length, max_tokens,
                                                   C.c_void_p(stream) if stream else None, C.byref(h)))
        else:
            _chk(lib().slimt_hip_ctx_create(model.h, max_batch, max_source_length,
                                            C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.B = self.S = 0
        self.N = model.V
        self._pinned = {}  # name -> _Pinned (translate_pinned)

    def close(self):
        if getattr(self, "h", None):
            lib().slimt_hip_ctx_destroy(self.h)
            self.h = None
            for b in getattr(self, "_pinned", {}).values():
                b.free()
            self._pinned = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        _chk(lib().slimt_hip_ctx_stream(self.h, C.byref(s)))
        return s.value or 0

    def synchronize(self):
        _chk(lib().slimt_hip_ctx_synchronize(self.h))

    def set_decode_mode(self, mode: int):
        """0 = auto (persistent fused decoder when supported), 1 = step-wise launches."""
        _chk(lib().slimt_hip_ctx_set_decode_mode(self.h, mode))

    def set_encode_rows(self, rows: int):
        """Rows per workgroup of the persistent D = 256 encoder: 0 = auto, 32, 64."""
        _chk(lib().slimt_hip_ctx_set_encode_rows(self.h, rows))

    def plan(self, S: int):
        """(encoder_fused, decoder_fused) for source length S in the current mode."""
        e, d = C.c_int(0), C.c_int(0)
        _chk(lib().slimt_hip_ctx_plan(self.h, S, C.byref(e), C.byref(d)))
        return bool(e.value), bool(d.value)

    def translate(self, ids, lengths, shortlist=None, limit_factor: float = 1.5, eos_id: int = 0,
                  want_align: bool = False):
        """Model::forward. Returns out_ids [B,Tmax], out_len [B], align|None."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        B, S = ids.shape
        Tmax = int(np.float32(limit_factor) * np.float32(S))
        T = max(Tmax, 1)
        sl = None if shortlist is None else np.ascontiguousarray(shortlist, dtype=np.uint32)
        out_ids = np.zeros((B, T), dtype=np.uint32)
        out_len = np.zeros(B, dtype=np.uint32)
        align = np.zeros((B, T, S), dtype=np.float32) if want_align else None
        _chk(lib().slimt_hip_translate(self.h, _p(ids), _p(lengths), B, S, _p(sl),
                                       0 if sl is None else sl.size, limit_factor, eos_id,
                                       _p(out_ids), _p(out_len), _p(align)))
        return out_ids, out_len, align

    def pinned_buffers(self, B: int, S: int, limit_factor: float = 1.5, want_align: bool = False):
        """This context's pinned staging arrays for a [B,S] batch: (ids, lengths, out_ids, out_len, align|None)."""
        T = max(int(np.float32(limit_factor) * np.float32(S)), 1)
        pin = lambda name: self._pinned.setdefault(name, _Pinned())
        return (pin("ids").array(np.uint32, (B, S)), pin("len").array(np.uint32, (B,)),
                pin("out").array(np.uint32, (B, T)), pin("ol").array(np.uint32, (B,)),
                pin("al").array(np.float32, (B, T, S)) if want_align else None)

    def translate_async(self, bufs, shortlist=None, generator=None, limit_factor: float = 1.5, eos_id: int = 0):
        """slimt_hip_translate_async[_generated] on arrays from pinned_buffers() (already filled);
        synchronize() before reading the outputs. `generator`: a ShortlistGenerator -- the batch's
        lexical shortlist is then generated on this context's stream (Model.cc:117-120)."""
        p_ids, p_len, p_out, p_ol, p_al = bufs
        B, S = p_ids.shape
        if generator is not None:
            _chk(lib().slimt_hip_translate_async_generated(self.h, generator.h, _p(p_ids), _p(p_len), B, S,
                                                           limit_factor, eos_id, _p(p_out), _p(p_ol), _p(p_al)))
            return
        sl = None if shortlist is None else np.ascontiguousarray(shortlist, dtype=np.uint32)
        _chk(lib().slimt_hip_translate_async(self.h, _p(p_ids), _p(p_len), B, S, _p(sl), 0 if sl is None else sl.size,
                                             limit_factor, eos_id, _p(p_out), _p(p_ol), _p(p_al)))

    def translate_pinned(self, ids, lengths, shortlist=None, limit_factor: float = 1.5, eos_id: int = 0,
                         want_align: bool = False, generator=None):
        """translate() through this context's pinned staging buffers and slimt_hip_translate_async:
        the persistent kernels then read and write host memory themselves, no copy is queued (host
        pipelines with several contexts: copies of one stream wait behind other streams' kernels).
        Returns copies of out_ids [B,Tmax], out_len [B], align|None."""
        ids = np.asarray(ids)
        B, S = ids.shape
        bufs = self.pinned_buffers(B, S, limit_factor, want_align)
        bufs[0][...] = ids
        bufs[1][...] = lengths
        self.translate_async(bufs, shortlist, generator, limit_factor, eos_id)
        self.synchronize()
        return bufs[2].copy(), bufs[3].copy(), (bufs[4].copy() if want_align else None)

    def translate_generated(self, generator, ids, lengths, limit_factor: float = 1.5, eos_id: int = 0,
                            want_align: bool = False):
        """Model::forward with its shortlist step (slimt_hip_translate_generated): host arrays, blocking."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        B, S = ids.shape
        T = max(int(np.float32(limit_factor) * np.float32(S)), 1)
        out_ids = np.zeros((B, T), dtype=np.uint32)
        out_len = np.zeros(B, dtype=np.uint32)
        align = np.zeros((B, T, S), dtype=np.float32) if want_align else None
        _chk(lib().slimt_hip_translate_generated(self.h, generator.h, _p(ids), _p(lengths), B, S, limit_factor,
                                                 eos_id, _p(out_ids), _p(out_len), _p(align)))
        return out_ids, out_len, align

    def translate_device(self, d_ids: int, d_lengths: int, B: int, S: int, d_shortlist: int,
                         n_shortlist: int, limit_factor: float, eos_id: int, d_out_ids: int,
                         d_out_len: int, d_align: int = 0, steps_hint: int = 0):
        """Device pointers (ints) in and out; asynchronous when steps_hint > 0."""
        vp = C.c_void_p
        _chk(lib().slimt_hip_translate_device(
            self.h, vp(d_ids), vp(d_lengths), B, S, vp(d_shortlist) if n_shortlist else None,
            n_shortlist, limit_factor, eos_id, vp(d_out_ids), vp(d_out_len),
            vp(d_align) if d_align else None, steps_hint))

    def translate_many_device(self, batches, S: int, limit_factor: float, eos_id: int, steps_hint: int = 0, generator=None):
        """slimt_hip_translate_many_device: `batches` = [(d_ids, d_lengths, B, d_shortlist, n_shortlist, d_out_ids,
        d_out_len, d_align[, S_j])] of device pointers (ints; 0 = none) -- ONE encoder and ONE decoder launch for all of them;
        S_j: that batch's own padded length (<= S; default S)."""
        arr = (_Batch * len(batches))()
        for j, b in enumerate(batches):
            d_ids, d_len, B, d_sl, n_sl, d_out, d_ol, d_al = b[:8]
            arr[j] = _Batch(d_ids, d_len, B, b[8] if len(b) > 8 else 0, d_sl if n_sl else None, n_sl, d_out, d_ol, d_al or None)
        if generator is not None:
            _chk(lib().slimt_hip_translate_many_device_generated(self.h, generator.h, arr, len(batches), S, limit_factor, eos_id, steps_hint))
            return
        _chk(lib().slimt_hip_translate_many_device(self.h, arr, len(batches), S, limit_factor, eos_id, steps_hint))

    def translate_many_async(self, bufs_list, shortlist=None, limit_factor: float = 1.5, eos_id: int = 0, generator=None):
        """slimt_hip_translate_many_async on a list of pinned buffer tuples (ids, lengths, out_ids, out_len, align|None),
        one shortlist (host array) or none for all; synchronize() before reading the outputs."""
        sl = None if shortlist is None else np.ascontiguousarray(shortlist, dtype=np.uint32)
        arr = (_Batch * len(bufs_list))()
        S = max(b[0].shape[1] for b in bufs_list)  # the launch's padded length; a batch may be padded to fewer tokens
        for j, (p_ids, p_len, p_out, p_ol, p_al) in enumerate(bufs_list):
            arr[j] = _Batch(p_ids.ctypes.data, p_len.ctypes.data, p_ids.shape[0], p_ids.shape[1], None if sl is None else sl.ctypes.data,
                            0 if sl is None else sl.size, p_out.ctypes.data, p_ol.ctypes.data,
                            None if p_al is None else p_al.ctypes.data)
        self._many_keep = (arr, sl)
        if generator is not None:  # every batch's own lexical shortlist, generated inside the encoder launch
            _chk(lib().slimt_hip_translate_many_async_generated(self.h, generator.h, arr, len(bufs_list), S, limit_factor, eos_id))
            return
        _chk(lib().slimt_hip_translate_many_async(self.h, arr, len(bufs_list), S, limit_factor, eos_id))

    def translate_device_generated(self, gen: "ShortlistGenerator", d_ids: int, d_lengths: int, B: int,
                                   S: int, limit_factor: float, eos_id: int, d_out_ids: int,
                                   d_out_len: int, d_align: int = 0, steps_hint: int = 0):
        """Shortlist generation + translate, all on this context's stream."""
        vp = C.c_void_p
        _chk(lib().slimt_hip_translate_device_generated(
            self.h, gen.h, vp(d_ids), vp(d_lengths), B, S, limit_factor, eos_id, vp(d_out_ids),
            vp(d_out_len), vp(d_align) if d_align else None, steps_hint))

    def encode(self, ids, lengths, want_embed=False, want_layers=False):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        B, S = ids.shape
        D, Le = self.model.D, self.model.Le
        emb = np.empty((B, S, D), dtype=np.float32) if want_embed else None
        layers = np.empty((Le, B, S, D), dtype=np.float32) if want_layers else None
        out = np.empty((B, S, D), dtype=np.float32)
        _chk(lib().slimt_hip_encode(self.h, _p(ids), _p(lengths), B, S, _p(emb), _p(layers), _p(out)))
        self.B, self.S = B, S
        return out, emb, layers

    def decode_begin(self, shortlist=None):
        sl = None if shortlist is None else np.ascontiguousarray(shortlist, dtype=np.uint32)
        _chk(lib().slimt_hip_decode_begin(self.h, _p(sl), 0 if sl is None else sl.size))
        self.N = self.model.V if sl is None else sl.size

    def decode_step(self, prev=None, want_attn=True, want_states=True):
        """Decoder::step. Returns logits [B,N], attn [B,H,1,S], states [Ld,B,D]."""
        B, S, m = self.B, self.S, self.model
        pv = None if prev is None else np.ascontiguousarray(prev, dtype=np.uint32)
        logits = np.empty((B, self.N), dtype=np.float32)
        attn = np.empty((B, m.H, 1, S), dtype=np.float32) if want_attn else None
        states = np.empty((m.Ld, B, m.D), dtype=np.float32) if want_states else None
        _chk(lib().slimt_hip_decode_step(self.h, _p(pv), _p(logits), _p(attn), _p(states)))
        return logits, attn, states

    def debug_decode_stamps(self, step: int):
        """Read the previous run's phase stamps (ticks of 10 ns) and arm `step`."""
        out = np.zeros(64, dtype=np.uint64)
        _chk(lib().slimt_hip_debug_decode_stamps(self.h, step, _p(out), 64))
        return out

    def debug_cross_attention(self, layer: int, yq, B: int, S: int, heads: int, literal: bool = False):
        """The decoder's cross-attention proper on projected queries yq [B, D] over the current batch's f32 K/V cache
        (decode_begin first): (joined heads [B, D], probabilities [B, H, S]); literal = the reference's own sequence."""
        yq = np.ascontiguousarray(yq, dtype=np.float32)
        out = np.empty_like(yq)
        attn = np.empty((B, heads, S), dtype=np.float32)
        _chk(lib().slimt_hip_debug_cross_attention(self.h, layer, 1 if literal else 0, _p(yq), _p(out), _p(attn)))
        return out, attn

    def debug_break_shortlist_handoff(self, broken: bool, poll_limit: int = 1 << 24):
        """The waiters of an in-launch shortlist look for a publication that never comes (tests: the timeout path)."""
        _chk(lib().slimt_hip_debug_break_shortlist_handoff(self.h, 1 if broken else 0, int(poll_limit)))

    def debug_kv_formats(self, layers: int, max_batch: int):
        """[layers][B] uint8 of the last batch: 0 = its cache is in the 20-bit form, 1 = 24-bit; None when the batch's
        caches are all in one form."""
        out = np.zeros(layers * max_batch, dtype=np.uint8)
        b = C.c_size_t(0)
        _chk(lib().slimt_hip_debug_kv_formats(self.h, _p(out), out.size, C.byref(b)))
        if b.value == 0:
            return None
        return out[: layers * b.value].reshape(layers, b.value).copy()

    def profile_enable(self, kernel_id: int):
        _chk(lib().slimt_hip_profile_enable(self.h, kernel_id))

    def profile_reset(self):
        _chk(lib().slimt_hip_profile_reset(self.h))

    def profile_read(self):
        n = C.c_uint64(0)
        ms, macs, wbytes = C.c_double(0), C.c_double(0), C.c_double(0)
        _chk(lib().slimt_hip_profile_read(self.h, C.byref(n), C.byref(ms), C.byref(macs),
                                          C.byref(wbytes)))
        return {"launches": n.value, "total_ms": ms.value, "int8_macs": macs.value,
                "weight_bytes": wbytes.value}


class ShortlistGenerator:
    """slimt::ShortlistGenerator (Shortlist.hh:38-90) on the device."""

    def __init__(self, blob: bytes, source_vocab: int, target_vocab: int, shared: bool = False,
                 check: bool = False, device: int = 0):
        self.h = C.c_void_p()
        self.target_vocab = target_vocab
        buf = bytes(blob)
        _chk(lib().slimt_hip_shortlist_create(buf, len(buf), source_vocab, target_vocab, int(shared),
                                              int(check), device, C.byref(self.h)))
        f, b = C.c_uint64(), C.c_uint64()
        _chk(lib().slimt_hip_shortlist_info(self.h, C.byref(f), C.byref(b)))
        self.frequent, self.best = int(f.value), int(b.value)

    def generate(self, ids, lengths) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        B, S = ids.shape
        out = np.zeros((self.target_vocab,), dtype=np.uint32)
        n = C.c_size_t()
        _chk(lib().slimt_hip_shortlist_generate(self.h, _p(ids), _p(lengths), B, S, _p(out), C.byref(n)))
        return out[: int(n.value)].copy()

    def generate_device(self, ctx: "Context", d_ids: int, d_lengths: int, B: int, S: int, d_out: int,
                        d_n: int) -> None:
        _chk(lib().slimt_hip_shortlist_generate_device(self.h, ctx.h, d_ids, d_lengths, B, S, d_out, d_n))

    def close(self):
        if self.h:
            lib().slimt_hip_shortlist_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass



# ---- the batching service (include/slimt_hip_service.h, libslimt_hip_host.so) -----------------------
_host_lib = None


def host_lib():
    """Load libslimt_hip_host.so (host/Service behind a C ABI; built by __graft_entry__.build())."""
    global _host_lib
    if _host_lib is not None:
        return _host_lib
    lib()  # libslimt_hip.so first: the host library links against it
    path = _build.HOST_LIB
    if not os.path.exists(path):
        raise SlimtHipError(f"{path} is missing: build it with `python -m slimt_amd.build`")
    H = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    H.slimt_hip_service_last_error.restype = C.c_char_p
    H.slimt_hip_service_create.argtypes = [vp, vp, sz, vp]
    H.slimt_hip_service_destroy.argtypes = [vp]
    H.slimt_hip_service_translate.argtypes = [vp, vp, vp, sz, vp]
    H.slimt_hip_result_view.argtypes = [vp] * 8
    H.slimt_hip_result_destroy.argtypes = [vp]
    _host_lib = H
    return H


class _ServiceConfig(C.Structure):
    _fields_ = [("max_words", C.c_uint64), ("wrap_length", C.c_uint64), ("limit_factor", C.c_float),
                ("workers_per_device", C.c_uint32), ("pad_id", C.c_uint32), ("eos_id", C.c_uint32),
                ("alignments", C.c_int32), ("lexical_shortlist", C.c_void_p), ("lexical_shortlist_bytes", C.c_uint64),
                ("source_vocab", C.c_uint64), ("target_vocab", C.c_uint64), ("shortlist_shared_vocab", C.c_int32),
                ("shortlist_check", C.c_int32), ("shortlist", C.c_void_p), ("n_shortlist", C.c_uint64),
                ("merge_batches", C.c_uint64), ("merge_words", C.c_uint64)]


class ServiceResult:
    """One request's result (slimt_hip_result): flat arrays, views valid while this object lives.
    targets[target_offsets[i]:target_offsets[i+1]] = sentence i's target ids (EOS included);
    alignment(i) = its [target tokens, source tokens] matrix."""

    def __init__(self, handle, source_lengths):
        self.h = handle
        n = C.c_size_t()
        ptrs = [C.c_void_p() for _ in range(6)]
        rc = host_lib().slimt_hip_result_view(self.h, C.byref(n), *[C.byref(p) for p in ptrs])
        if rc:
            raise SlimtHipError(host_lib().slimt_hip_service_last_error().decode())
        self.n = n.value

        def arr(p, dtype, count):
            if not count or not p.value:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))

        self.target_offsets = arr(ptrs[1], np.uint64, self.n + 1)
        self.targets = arr(ptrs[0], np.uint32, int(self.target_offsets[-1]) if self.n else 0)
        self.padded_length = arr(ptrs[2], np.uint32, self.n)
        self.batch = arr(ptrs[3], np.uint64, self.n)
        self.align_offsets = arr(ptrs[5], np.uint64, self.n + 1)
        self.alignments = arr(ptrs[4], np.float32, int(self.align_offsets[-1]) if self.n else 0)
        self.source_lengths = source_lengths

    def target(self, i: int) -> np.ndarray:
        return self.targets[int(self.target_offsets[i]):int(self.target_offsets[i + 1])]

    def alignment(self, i: int) -> np.ndarray:
        a, b = int(self.align_offsets[i]), int(self.align_offsets[i + 1])
        rows = int(self.target_offsets[i + 1] - self.target_offsets[i])
        return self.alignments[a:b].reshape(rows, -1) if b > a else np.zeros((rows, 0), np.float32)

    def close(self):
        if getattr(self, "h", None):
            host_lib().slimt_hip_result_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchService:
    """host/Service over its C ABI: tokenised sentences in, target ids + alignment rows out; token-budget
    batches, double-buffered pinned workers, per-batch lexical shortlist on the device."""

    def __init__(self, models, max_words: int = 8192, wrap_length: int = 128, limit_factor: float = 1.5,
                 workers_per_device: int = 6, pad_id: int = 0, eos_id: int = 0, alignments: bool = True,
                 lexical_shortlist: bytes = b"", source_vocab: int = 0, target_vocab: int = 0,
                 shared_vocab: bool = False, check: bool = False, shortlist=None, merge_batches: int = 0, merge_words: int = 0):
        """merge_batches / merge_words: merged launches (0 = the library's defaults: up to 8 consecutive batches of one padded
        length per launch pair within 8192 words; merge_batches = 1: never)."""
        self._keep = []
        cfg = _ServiceConfig(max_words, wrap_length, limit_factor, workers_per_device, pad_id, eos_id,
                             1 if alignments else 0, None, 0, source_vocab, target_vocab,
                             1 if shared_vocab else 0, 1 if check else 0, None, 0, merge_batches, merge_words)
        if lexical_shortlist:
            buf = C.create_string_buffer(bytes(lexical_shortlist), len(lexical_shortlist))
            self._keep.append(buf)
            cfg.lexical_shortlist = C.cast(buf, C.c_void_p)
            cfg.lexical_shortlist_bytes = len(lexical_shortlist)
        elif shortlist is not None:
            sl = np.ascontiguousarray(shortlist, dtype=np.uint32)
            self._keep.append(sl)
            cfg.shortlist = sl.ctypes.data_as(C.c_void_p)
            cfg.n_shortlist = sl.size
        arr = (C.c_void_p * len(models))(*[m.h for m in models])
        self.h = C.c_void_p()
        if host_lib().slimt_hip_service_create(C.byref(cfg), arr, len(models), C.byref(self.h)):
            raise SlimtHipError(host_lib().slimt_hip_service_last_error().decode())

    def translate_flat(self, tokens: np.ndarray, offsets: np.ndarray) -> ServiceResult:
        """tokens uint32 (flat), offsets uint64 [n + 1]. Blocking; thread-safe."""
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = C.c_void_p()
        if host_lib().slimt_hip_service_translate(self.h, _p(tokens), _p(offsets), offsets.size - 1, C.byref(out)):
            raise SlimtHipError(host_lib().slimt_hip_service_last_error().decode())
        return ServiceResult(out, np.diff(offsets).astype(np.int64))

    def translate(self, sentences) -> ServiceResult:
        lens = np.fromiter((len(s) for s in sentences), dtype=np.uint64, count=len(sentences))
        offsets = np.zeros(len(sentences) + 1, np.uint64)
        np.cumsum(lens, out=offsets[1:])
        tokens = np.fromiter(itertools.chain.from_iterable(sentences), dtype=np.uint32, count=int(offsets[-1]))
        return self.translate_flat(tokens, offsets)

    def close(self):
        if getattr(self, "h", None):
            host_lib().slimt_hip_service_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
