"""ctypes binding of the CPU oracle (oracle/libslimt_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package (slimt_amd).
PARITY UNPINNED: see oracle/slimt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libslimt_oracle.so")

FAITHFUL, PORTABLE = 0, 1


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("slimt_oracle.c", "slimt_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libslimt_oracle.so"])
    return _LIB_PATH


class _Param(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("type", C.c_int32),
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("data", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, f32, sz, i32 = C.c_void_p, C.c_float, C.c_size_t, C.c_int
        L.so_exp.restype = f32
        L.so_exp.argtypes = [f32]
        L.so_sigmoid.restype = f32
        L.so_sigmoid.argtypes = [f32]
        L.so_row_sum.restype = f32
        L.so_row_sum.argtypes = [vp, sz]
        L.so_set_mode.argtypes = [i32]
        L.so_get_mode.restype = i32
        L.so_quantize.argtypes = [vp, f32, sz, vp]
        L.so_gemm_i8_signed.argtypes = [vp, vp, sz, sz, sz, vp]
        L.so_gemm_i8_shifted.argtypes = [vp, vp, sz, sz, sz, vp]
        L.so_affine.argtypes = [vp, sz, sz, vp, sz, vp, f32, f32, vp]
        L.so_affine_acc.argtypes = [vp, sz, sz, vp, sz, f32, vp]
        L.so_affine_select.argtypes = [vp, sz, sz, vp, sz, vp, f32, f32, vp, sz, vp]
        L.so_affine_ruy.argtypes = [vp, sz, sz, vp, sz, vp, f32, f32, vp]
        L.so_prepare_weight_transposed.argtypes = [vp, vp, f32, sz, sz]
        L.so_prepare_weight_quantized_transposed.argtypes = [vp, vp, sz, sz]
        L.so_unquantize_embedding.argtypes = [vp, f32, sz, vp]
        L.so_layer_norm.argtypes = [vp, vp, vp, f32, sz, sz, vp]
        L.so_softmax.argtypes = [vp, sz, sz, vp]
        L.so_highway.argtypes = [vp, vp, vp, sz, vp]
        L.so_relu.argtypes = [vp, sz, vp]
        L.so_add.argtypes = [vp, vp, sz, vp]
        L.so_sinusoidal_signal.argtypes = [i32, sz, sz, vp]
        L.so_transform_embedding.argtypes = [vp, sz, sz, sz, sz]
        L.so_index_select.argtypes = [vp, vp, sz, sz, vp]
        L.so_transpose_3120.argtypes = [vp, sz, sz, sz, sz, vp]
        L.so_bmm.argtypes = [vp, vp, sz, sz, sz, sz, sz, i32, f32, vp]
        L.so_sdpa.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz, sz, vp, vp]
        L.so_greedy_sample.argtypes = [vp, sz, sz, vp, vp]
        L.so_make_mask.argtypes = [vp, sz, sz, vp]
        L.so_model_create.restype = vp
        L.so_model_create.argtypes = [vp, sz, i32, i32, i32]
        L.so_model_destroy.argtypes = [vp]
        L.so_model_set_reference_cost.argtypes = [vp, i32]
        L.so_model_set_threads.argtypes = [vp, i32]
        L.so_embed.argtypes = [vp, vp, sz, sz, vp]
        L.so_encode.argtypes = [vp, vp, vp, sz, sz, vp]
        L.so_encoder_layer.argtypes = [vp, i32, vp, vp, sz, sz, vp, vp]
        L.so_decode_step.argtypes = [vp, vp, vp, sz, sz, vp, vp, vp, sz, vp, vp]
        L.so_cross_attention.argtypes = [vp, i32, vp, vp, vp, sz, sz, vp, vp]
        L.so_shortlist_checksum.restype = C.c_uint64
        L.so_shortlist_checksum.argtypes = [vp, sz]
        L.so_shortlist_parse.restype = i32
        L.so_shortlist_parse.argtypes = [vp, sz, i32, sz, vp]
        L.so_shortlist_generate.restype = sz
        L.so_shortlist_generate.argtypes = [vp, i32, sz, sz, vp, sz, vp]
        L.so_translate.restype = sz
        L.so_translate.argtypes = [vp, vp, vp, sz, sz, vp, sz, f32, C.c_uint32, vp, vp, vp]
        _lib = L
    return _lib


def set_mode(mode: int) -> None:
    lib().so_set_mode(mode)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- op level -------------------------------------------------------------

def quantize(x, a_quant):
    x = _f32(x)
    q = np.empty(x.shape, dtype=np.int8)
    lib().so_quantize(_p(x), a_quant, x.size, _p(q))
    return q


def gemm_i8(q, W, shifted=True):
    q = np.ascontiguousarray(q, dtype=np.int8)
    W = np.ascontiguousarray(W, dtype=np.int8)
    M, K = q.shape
    N = W.shape[0]
    acc = np.empty((M, N), dtype=np.int32)
    f = lib().so_gemm_i8_shifted if shifted else lib().so_gemm_i8_signed
    f(_p(q), _p(W), M, K, N, _p(acc))
    return acc


def affine(x, W, bias, a_quant, b_quant, provider="intgemm"):
    x = _f32(x)
    W = np.ascontiguousarray(W, dtype=np.int8)
    M, K = x.reshape(-1, x.shape[-1]).shape
    N = W.shape[0]
    b = None if bias is None else _f32(bias).reshape(-1)
    y = np.empty(x.shape[:-1] + (N,), dtype=np.float32)
    f = lib().so_affine if provider == "intgemm" else lib().so_affine_ruy
    f(_p(x), M, K, _p(W), N, _p(b), a_quant, b_quant, _p(y))
    return y


def affine_acc(x, W, a_quant):
    x = _f32(x)
    W = np.ascontiguousarray(W, dtype=np.int8)
    M, K = x.reshape(-1, x.shape[-1]).shape
    N = W.shape[0]
    acc = np.empty((M, N), dtype=np.int32)
    lib().so_affine_acc(_p(x), M, K, _p(W), N, a_quant, _p(acc))
    return acc


def affine_select(x, W, bias, a_quant, b_quant, idx):
    x = _f32(x)
    W = np.ascontiguousarray(W, dtype=np.int8)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    M, K = x.reshape(-1, x.shape[-1]).shape
    N = W.shape[0]
    b = _f32(bias).reshape(-1)
    y = np.empty(x.shape[:-1] + (idx.size,), dtype=np.float32)
    lib().so_affine_select(_p(x), M, K, _p(W), N, _p(b), a_quant, b_quant, _p(idx), idx.size, _p(y))
    return y


def layer_norm(x, scale, bias, eps=1e-6):
    x = _f32(x)
    s, b = _f32(scale).reshape(-1), _f32(bias).reshape(-1)
    cols = x.shape[-1]
    y = np.empty_like(x)
    lib().so_layer_norm(_p(x), _p(s), _p(b), eps, x.size // cols, cols, _p(y))
    return y


def softmax(x):
    x = _f32(x)
    cols = x.shape[-1]
    y = np.empty_like(x)
    lib().so_softmax(_p(x), x.size // cols, cols, _p(y))
    return y


def highway(x, y, g):
    x, y, g = _f32(x), _f32(y), _f32(g)
    out = np.empty_like(x)
    lib().so_highway(_p(x), _p(y), _p(g), x.size, _p(out))
    return out


def sdpa(q, k, v, mask):
    q, k, v, mask = _f32(q), _f32(k), _f32(v), _f32(mask)
    B, H, Tq, dh = q.shape
    S = k.shape[2]
    out = np.empty_like(q)
    attn = np.empty((B, H, Tq, S), dtype=np.float32)
    lib().so_sdpa(_p(q), _p(k), _p(v), _p(mask), B, H, Tq, S, dh, _p(out), _p(attn))
    return out, attn


def sinusoidal_signal(start, seq, dim):
    out = np.empty((seq, dim), dtype=np.float32)
    lib().so_sinusoidal_signal(start, seq, dim, _p(out))
    return out


def greedy_sample(logits, words=None):
    logits = _f32(logits)
    B, N = logits.shape
    w = None if words is None else np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(B, dtype=np.uint32)
    lib().so_greedy_sample(_p(logits), B, N, _p(w), _p(out))
    return out


def make_mask(lengths, S):
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    m = np.empty((lengths.size, S), dtype=np.float32)
    lib().so_make_mask(_p(lengths), lengths.size, S, _p(m))
    return m


# ---- model level ------------------------------------------------------------

class OracleModel:
    """Restatement of slimt::Transformer + Model::forward on a synth.Model."""

    def __init__(self, model, reference_cost: bool = False, threads: int = 1):
        self.model = model
        self._keep = []
        arr = (_Param * len(model.params))()
        for i, p in enumerate(model.params.values()):
            buf = np.frombuffer(p.payload(), dtype=np.uint8).copy()
            self._keep.append(buf)
            name = p.name.encode()
            self._keep.append(name)
            arr[i] = _Param(name, 0 if p.kind == "f32" else 1, p.rows, p.cols,
                            buf.ctypes.data_as(C.c_void_p))
        self._arr = arr
        self.h = lib().so_model_create(C.cast(arr, C.c_void_p), len(model.params),
                                       model.enc_layers, model.dec_layers, model.H)
        if not self.h:
            raise RuntimeError("so_model_create failed (missing parameter?)")
        lib().so_model_set_reference_cost(self.h, int(reference_cost))
        lib().so_model_set_threads(self.h, threads)
        self.D, self.H, self.V, self.Ld = model.D, model.H, model.V, model.dec_layers

    def __del__(self):
        if getattr(self, "h", None):
            lib().so_model_destroy(self.h)
            self.h = None

    def embed(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        B, S = ids.shape
        out = np.empty((B, S, self.D), dtype=np.float32)
        lib().so_embed(self.h, _p(ids), B, S, _p(out))
        return out

    def encoder_layer(self, layer, x, mask, want_attn=False):
        x, mask = _f32(x), _f32(mask)
        B, S, _ = x.shape
        out = np.empty_like(x)
        attn = np.empty((B, self.H, S, S), dtype=np.float32) if want_attn else None
        lib().so_encoder_layer(self.h, layer, _p(x), _p(mask), B, S, _p(out), _p(attn))
        return (out, attn) if want_attn else out

    def encode(self, x, mask):
        x, mask = _f32(x), _f32(mask)
        B, S, _ = x.shape
        out = np.empty_like(x)
        lib().so_encode(self.h, _p(x), _p(mask), B, S, _p(out))
        return out

    def cross_attention(self, layer, yq, encoder_out, mask):
        """The attention proper of decoder layer `layer` (0-based) on a projected query yq [B,D]: joined heads
        [B,D] (before the output projection) and probabilities [B,H,S], in the current float order."""
        yq, encoder_out, mask = _f32(yq), _f32(encoder_out), _f32(mask)
        B, S, _ = encoder_out.shape
        out = np.empty((B, self.D), dtype=np.float32)
        attn = np.empty((B, self.H, S), dtype=np.float32)
        lib().so_cross_attention(self.h, layer, _p(yq), _p(encoder_out), _p(mask), B, S, _p(out), _p(attn))
        return out, attn

    def decode_step(self, encoder_out, mask, states, prev, shortlist):
        """states: f32 [Ld,B,D], updated in place. Returns logits, attn."""
        encoder_out, mask = _f32(encoder_out), _f32(mask)
        assert states.dtype == np.float32 and states.flags.c_contiguous
        B, S, _ = encoder_out.shape
        pv = None if prev is None else np.ascontiguousarray(prev, dtype=np.uint32)
        sl = None if shortlist is None else np.ascontiguousarray(shortlist, dtype=np.uint32)
        N = self.V if sl is None else sl.size
        logits = np.empty((B, N), dtype=np.float32)
        attn = np.empty((B, self.H, 1, S), dtype=np.float32)
        lib().so_decode_step(self.h, _p(encoder_out), _p(mask), B, S, _p(states), _p(pv),
                             _p(sl), 0 if sl is None else sl.size, _p(logits), _p(attn))
        return logits, attn

    def translate(self, ids, lengths, shortlist=None, limit_factor=1.5, eos_id=0,
                  want_align=False):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        B, S = ids.shape
        Tmax = int(np.float32(limit_factor) * np.float32(S))
        sl = None if shortlist is None else np.ascontiguousarray(shortlist, dtype=np.uint32)
        out_ids = np.zeros((B, max(Tmax, 1)), dtype=np.uint32)
        out_len = np.zeros(B, dtype=np.uint32)
        align = np.zeros((B, max(Tmax, 1), S), dtype=np.float32) if want_align else None
        steps = lib().so_translate(self.h, _p(ids), _p(lengths), B, S, _p(sl),
                                   0 if sl is None else sl.size, limit_factor, eos_id,
                                   _p(out_ids), _p(out_len), _p(align))
        return out_ids, out_len, align, int(steps)


class _Shortlist(C.Structure):
    _fields_ = [("frequent", C.c_uint64), ("best", C.c_uint64),
                ("word_to_offset_size", C.c_uint64), ("shortlist_size", C.c_uint64),
                ("word_to_offset", C.c_void_p), ("shortlist", C.c_void_p)]


def shortlist_checksum(blob: bytes) -> int:
    return int(lib().so_shortlist_checksum(blob, len(blob)))


class OracleShortlist:
    """ShortlistGenerator (Shortlist.cc:41-175) over a binary shortlist blob."""

    def __init__(self, blob: bytes, source_vocab: int, target_vocab: int, shared: bool = False,
                 check: bool = False):
        self._blob = bytes(blob)  # the parsed struct points into it
        self._buf = C.create_string_buffer(self._blob, len(self._blob))
        self._sl = _Shortlist()
        rc = lib().so_shortlist_parse(self._buf, len(self._blob), int(check), target_vocab,
                                      C.byref(self._sl))
        if rc != 0:
            raise ValueError(f"binary shortlist rejected (code {rc})")
        self.source_vocab, self.target_vocab, self.shared = source_vocab, target_vocab, shared
        self.frequent, self.best = int(self._sl.frequent), int(self._sl.best)

    def generate(self, ids: np.ndarray, lengths: np.ndarray) -> np.ndarray:
        """ids [B,S] padded, lengths [B]: the words of Input::words() are the
        first lengths[b] tokens of every row (Input.cc:24)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        words = np.concatenate([ids[b, : int(lengths[b])] for b in range(ids.shape[0])]) \
            if ids.shape[0] else np.zeros((0,), np.uint32)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        out = np.zeros((max(1, self.target_vocab),), dtype=np.uint32)
        n = lib().so_shortlist_generate(C.byref(self._sl), int(self.shared), self.source_vocab,
                                        self.target_vocab, _p(words), words.size, _p(out))
        return out[: int(n)].copy()


def batcher_generate(requests, max_words: int, wrap_length: int, tgt_length_limit_factor: float = 3.0):
    """Batcher::enqueue of every request, then Batcher::generate until empty
    (slimt/Batcher.cc:77-147). requests: list of lists of segment lengths (request
    id = list index). Returns the batches as lists of (request id, segment index)."""
    pivot_slack = int(np.float32(wrap_length) * np.float32(tgt_length_limit_factor) - np.float32(wrap_length))
    n_buckets = wrap_length + pivot_slack + 1
    if n_buckets - 1 > max_words:
        raise ValueError("wrap_length > max_words")
    buckets = {}
    running_max = 0
    for rid, lengths in enumerate(requests):
        for idx, n in enumerate(lengths):
            buckets.setdefault(int(n), []).append((rid, idx))  # std::set order: (request id, index)
            running_max = max(running_max, int(n))
    for b in buckets.values():
        b.sort()
    batches = []
    while True:
        batch = []
        full = False
        for length in range(0, running_max + 1):
            bucket = buckets.get(length, [])
            while bucket:
                if (len(batch) + 1) * length <= max_words:
                    batch.append(bucket.pop(0))
                else:
                    full = True
                    break
            if full:
                break
        if not batch:
            return batches
        batches.append(batch)

