"""Synthetic Bergamot/Marian student models and inputs (no network, no real
model files in this environment), plus a reader/writer for the Marian ``.bin``
v1 container that slimt's ``io::load_items`` consumes.

Reference format: slimt/Io.hh:19-29, slimt/Io.cc:114-161 (header / names /
shapes / 256-B aligned payloads), type ids Io.cc:37-84
(float32 = 0x0404, int8 = 0x0101, intgemm8 = 0x4101). An ``intgemm8`` payload is
the int8 matrix stored as B^T -- ``[N][K]``, K contiguous -- followed by one
f32 quantisation multiplier (Io.cc:225-239). Parameter names/shapes follow
Modules.cc:336-406 and Transformer.cc:104-118,227-232.
"""
from __future__ import annotations

import dataclasses
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TYPE_F32 = 0x0404
TYPE_I8 = 0x0101
TYPE_IG8 = 0x4101

PRESETS = {
    # name: (D, F, H, enc_layers, dec_layers, V)   (Model.cc:206-245, README.md:29-37)
    "tiny11": (256, 1536, 8, 6, 2, 32000),
    "base": (512, 2048, 8, 6, 2, 32000),
    # small shapes for fast CPU tests (same structure)
    "micro": (64, 128, 4, 2, 2, 512),
    "mini": (128, 256, 8, 2, 2, 2048),
}


@dataclasses.dataclass
class Param:
    name: str
    kind: str  # "f32" | "ig8"
    rows: int  # logical shape[-2]  (K for weights, V for Wemb)
    cols: int  # logical shape[-1]  (N for weights, D for Wemb)
    data: np.ndarray  # f32 [rows, cols]; ig8: int8 payload in FILE order
    mult: float = 0.0  # ig8 only: trailing quantisation multiplier (b_quant)

    def payload(self) -> bytes:
        if self.kind == "f32":
            return np.ascontiguousarray(self.data, dtype=np.float32).tobytes()
        return np.ascontiguousarray(self.data, dtype=np.int8).tobytes() + struct.pack(
            "<f", self.mult
        )


@dataclasses.dataclass
class Model:
    name: str
    D: int
    F: int
    H: int
    enc_layers: int
    dec_layers: int
    V: int
    params: Dict[str, Param]

    def __getitem__(self, k: str) -> Param:
        return self.params[k]


# Synthetic model FAMILIES: the default is what every fixture and benchmark of rounds 1-5 used; the others
# vary what the K/V cache forms' hit rates depend on -- the spread of the int8 weights (sigma, or a heavy-tailed Student-t
# draw of the same variance), the range of the activation multipliers (QuantMultA = 127 / absmax_a) and the spread of the
# LayerNorm scales. Any values are legal models; parity must hold for all of them and the forms must stay exact.
FAMILIES = {
    "default": {},
    "w48": {"weight_sigma": 48.0},
    "w64": {"weight_sigma": 64.0},
    "heavy": {"weight_dist": "student3"},
    "a2_6": {"absmax_a": (2.0, 6.0)},
    "a8_24": {"absmax_a": (8.0, 24.0)},
    "ln0.3": {"ln_sigma": 0.3},
    "w64_heavy_a8_24": {"weight_sigma": 64.0, "weight_dist": "student3", "absmax_a": (8.0, 24.0)},
}
_FAMILY = {"weight_sigma": 32.0, "weight_dist": "normal", "absmax_a": (4.0, 12.0), "ln_sigma": 0.05}


def _weights(rng, shape, fam):
    if fam["weight_dist"] == "student3":  # heavy tails, scaled to the same standard deviation (Var t_3 = 3)
        x = rng.standard_t(3.0, size=shape) * (fam["weight_sigma"] / np.sqrt(3.0))
    else:
        x = rng.normal(0.0, fam["weight_sigma"], size=shape)
    return np.clip(np.rint(x), -127, 127).astype(np.int8)


def _ig8(rng, name, K, N, absmax_range=(0.3, 1.0), fam=_FAMILY) -> Param:
    """int8 weight, logical [K, N], payload [N][K]."""
    q = _weights(rng, (N, K), fam)
    absmax = rng.uniform(*absmax_range)
    return Param(name, "ig8", K, N, q, float(np.float32(127.0 / absmax)))


def _f32(name, arr) -> Param:
    arr = np.asarray(arr, dtype=np.float32)
    if arr.ndim == 1:
        arr = arr[None, :]
    return Param(name, "f32", arr.shape[0], arr.shape[1], arr)


def _quant_a(rng, name, fam=_FAMILY) -> Param:
    absmax = rng.uniform(*fam["absmax_a"])
    return _f32(name, np.array([[127.0 / absmax]], dtype=np.float32))


def make_model(
    preset: str = "tiny11",
    seed: int = 1234,
    eos_id: int = 0,
    eos_bias: float = -100.0,
    dims: Optional[Tuple[int, int, int, int, int, int]] = None,
    family: str = "default",
) -> Model:
    """Seeded random model with realistic quantisation ranges.

    eos_bias is added to ``decoder_ff_logit_out_b[eos_id]``: very negative =>
    no sentence ever finishes (fixed-length benchmark runs); moderately
    positive => staggered finishing (correctness fixtures).
    """
    D, F, H, Le, Ld, V = dims if dims is not None else PRESETS[preset]
    fam = dict(_FAMILY, **FAMILIES[family])
    rng = np.random.Generator(np.random.PCG64(seed))
    P: Dict[str, Param] = {}

    def add(p: Param):
        P[p.name] = p

    def affine(prefix, w, b, K, N):
        # W2 is kept small: relu() has a positive mean, and a large random W2
        # turns that mean into one input-independent direction that dominates
        # the residual stream (every sentence then decodes the same token).
        add(_ig8(rng, prefix + w, K, N, (0.05, 0.15) if w == "W2" else (0.3, 1.0), fam))
        add(_f32(prefix + b, rng.normal(0.0, 0.05, size=(1, N))))
        add(_quant_a(rng, prefix + w + "_QuantMultA", fam))

    def ln(prefix):
        add(_f32(prefix + "_ln_scale", 1.0 + rng.normal(0.0, fam["ln_sigma"], size=(1, D))))
        add(_f32(prefix + "_ln_bias", rng.normal(0.0, 0.05, size=(1, D))))

    # Wemb: payload [V][D] (B^T of the output layer's [D, V]); Io.cc:182-224
    q = _weights(rng, (V, D), fam)
    # small embedding range: with tied embeddings a large E makes the random
    # decoder collapse onto "repeat the previous token"; keeping |E|*sqrt(D)
    # below the other residual-stream terms gives varied greedy outputs.
    add(Param("Wemb", "ig8", V, D, q, float(np.float32(127.0 / rng.uniform(0.3, 0.6)))))
    add(_quant_a(rng, "none_QuantMultA", fam))  # none_QuantMultA, Transformer.cc:106-112
    out_b = rng.normal(0.0, 0.05, size=(1, V)).astype(np.float32)
    out_b[0, eos_id] += np.float32(eos_bias)
    add(_f32("decoder_ff_logit_out_b", out_b))

    for i in range(1, Le + 1):
        L = f"encoder_l{i}"
        for s in "qkvo":
            affine(L + "_self_", "W" + s, "b" + s, D, D)
        ln(L + "_self_Wo")
        affine(L + "_ffn_", "W1", "b1", D, F)
        affine(L + "_ffn_", "W2", "b2", F, D)
        ln(L + "_ffn_ffn")
    for i in range(1, Ld + 1):
        L = f"decoder_l{i}"
        for s in "qkvo":
            affine(L + "_context_", "W" + s, "b" + s, D, D)
        ln(L + "_context_Wo")
        affine(L + "_ffn_", "W1", "b1", D, F)
        affine(L + "_ffn_", "W2", "b2", F, D)
        ln(L + "_ffn_ffn")
        add(_ig8(rng, L + "_rnn_W", D, D, fam=fam))
        add(_quant_a(rng, L + "_rnn_W_QuantMultA", fam))
        add(_ig8(rng, L + "_rnn_Wf", D, D, fam=fam))
        add(_f32(L + "_rnn_bf", rng.normal(0.0, 0.05, size=(1, D))))
        add(_quant_a(rng, L + "_rnn_Wf_QuantMultA", fam))
        ln(L + "_rnn_ffn")
    # Tied embeddings + residual stream make a random decoder repeat its own
    # previous token forever. Flipping the sign of the last decoder LayerNorm
    # scale turns that self-reinforcement into self-avoidance, which gives
    # varied greedy trajectories (better test coverage; any values are legal).
    last = P[f"decoder_l{Ld}_ffn_ffn_ln_scale"]
    last.data = (-last.data).astype(np.float32)
    return Model(preset, D, F, H, Le, Ld, V, P)


def make_shortlist(V: int, n: int, seed: int = 99, frequent: int = 100) -> np.ndarray:
    """Sorted unique target ids containing 0..frequent-1, size multiple of 8
    (Shortlist.cc:125-127,158-172)."""
    n = min(n, V)
    n -= n % 8
    rng = np.random.Generator(np.random.PCG64(seed))
    frequent = min(frequent, n)
    rest = rng.choice(np.arange(frequent, V), size=n - frequent, replace=False)
    ids = np.concatenate([np.arange(frequent), rest]).astype(np.uint32)
    ids.sort()
    return ids


SHORTLIST_MAGIC = 0xF11A48D5013417F5  # Shortlist.hh:40


def shortlist_checksum(body: bytes) -> int:
    """hash_combine over the uint64 words after {magic, checksum}
    (Shortlist.cc:66-76, Utils.hh:47-67; libstdc++'s std::hash<uint64_t> is the identity)."""
    seed = 0
    mask = (1 << 64) - 1
    for (v,) in struct.iter_unpack("<Q", body[: len(body) // 8 * 8]):
        seed ^= (v + 0x9E3779B9 + ((seed << 6) & mask) + (seed >> 2)) & mask
    return seed


def make_lexical_shortlist(V_src: int, V_tgt: int, frequent: int = 100, best: int = 100,
                           seed: int = 7, empty_fraction: float = 0.1, min_count: int = 0) -> bytes:
    """A synthetic binary lexical shortlist in the layout ShortlistGenerator::load
    reads (Shortlist.hh:77-84, Shortlist.cc:41-104): header of six uint64, the
    word_to_offset table (V_src + 1 entries), then per source word a sorted list
    of <= `best` unique target ids; some words have an empty list."""
    rng = np.random.Generator(np.random.PCG64(seed))
    counts = rng.integers(min(min_count, best), best + 1, size=V_src)
    counts[rng.random(V_src) < empty_fraction] = 0
    counts[-1] = max(1, min(best, V_tgt))  # content_check wants every offset but the last < size (Shortlist.cc:18-21)
    offsets = np.zeros(V_src + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(counts)
    lists = np.zeros(int(offsets[-1]), dtype=np.uint32)
    if best <= 1:  # one candidate per word: no sorting, no per-word draw (benchmark-sized vocabularies)
        lists[:] = rng.integers(0, V_tgt, size=lists.size)
    else:
        for w in range(V_src):
            k = int(counts[w])
            if k:
                lists[int(offsets[w]): int(offsets[w + 1])] = np.sort(
                    rng.choice(V_tgt, size=k, replace=False)).astype(np.uint32)
    body = struct.pack("<4Q", frequent, best, offsets.size, lists.size) + offsets.tobytes() + lists.tobytes()
    return struct.pack("<2Q", SHORTLIST_MAGIC, shortlist_checksum(body)) + body


def make_batch(
    V: int, B: int, S: int, seed: int = 4321, eos_id: int = 0, ragged: bool = False,
    pad_id: int = 0,
) -> Tuple[np.ndarray, np.ndarray]:
    """Token ids [B,S] (uint32, padded with pad_id) and lengths [B]. Every
    sentence ends with EOS (TextProcessor.cc:142-143)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = np.full((B, S), pad_id, dtype=np.uint32)
    if ragged:
        lens = rng.integers(max(1, S // 4), S + 1, size=B)
        lens[rng.integers(0, B)] = S  # at least one full-length row
    else:
        lens = np.full(B, S)
    for b in range(B):
        L = int(lens[b])
        ids[b, : L - 1] = rng.integers(2, V, size=L - 1)
        ids[b, L - 1] = eos_id
    return ids, lens.astype(np.uint32)


# --------------------------------------------------------------------------
# Marian .bin v1 container
# --------------------------------------------------------------------------

def write_bin(model: Model, extra_yaml: bool = True) -> bytes:
    items: List[Tuple[str, int, Tuple[int, int], bytes]] = []
    for p in model.params.values():
        t = TYPE_F32 if p.kind == "f32" else TYPE_IG8
        items.append((p.name, t, (p.rows, p.cols), p.payload()))
    if extra_yaml:  # scripts/marian-file-inspect.py:160-165
        y = b"synthetic: true\n\x00"
        items.append(("special:model.yml", TYPE_I8, (1, len(y)), y))
    out = bytearray()
    out += struct.pack("<QQ", 1, len(items))
    for name, t, shape, payload in items:
        out += struct.pack("<QQQQ", len(name) + 1, t, len(shape), len(payload))
    for name, *_ in items:
        out += name.encode() + b"\x00"
    for _, _, shape, _ in items:
        out += struct.pack("<%di" % len(shape), *shape)
    pos = len(out) + 8
    pad = (-pos) % 256
    out += struct.pack("<Q", pad) + b"\x00" * pad
    for *_, payload in items:
        out += payload
    return bytes(out)


def read_bin(buf: bytes) -> List[Param]:
    """Parse like io::load_items (Io.cc:114-161) without its conversions."""
    off = 0
    version, n = struct.unpack_from("<QQ", buf, off)
    off += 16
    if version != 1:
        raise ValueError("binary file version %d != 1" % version)
    headers = [struct.unpack_from("<QQQQ", buf, off + 32 * i) for i in range(n)]
    off += 32 * n
    names = []
    for h in headers:
        names.append(buf[off : off + h[0] - 1].decode())
        off += h[0]
    shapes = []
    for h in headers:
        shapes.append(struct.unpack_from("<%di" % h[2], buf, off))
        off += 4 * h[2]
    (pad,) = struct.unpack_from("<Q", buf, off)
    off += 8 + pad
    out: List[Param] = []
    for h, name, shape in zip(headers, names, shapes):
        raw = buf[off : off + h[3]]
        off += h[3]
        rows, cols = (shape[-2], shape[-1]) if len(shape) >= 2 else (1, shape[-1])
        if h[1] == TYPE_F32:
            out.append(Param(name, "f32", rows, cols,
                             np.frombuffer(raw, dtype=np.float32).reshape(rows, cols).copy()))
        elif h[1] == TYPE_IG8:
            n_el = rows * cols
            q = np.frombuffer(raw[:n_el], dtype=np.int8)
            (mult,) = struct.unpack_from("<f", raw, n_el)
            q = q.reshape(rows, cols) if name == "Wemb" else q.reshape(cols, rows)
            out.append(Param(name, "ig8", rows, cols, q.copy(), mult))
        else:
            out.append(Param(name, "i8", rows, cols, np.frombuffer(raw, dtype=np.int8).copy()))
    return out


def model_from_bin(buf: bytes, preset_name: str = "from_bin", heads: int = 8,
                   enc_layers: int = 6, dec_layers: int = 2) -> Model:
    params = {p.name: p for p in read_bin(buf) if p.kind in ("f32", "ig8")}
    V, D = params["Wemb"].rows, params["Wemb"].cols
    F = params["encoder_l1_ffn_W1"].cols
    return Model(preset_name, D, F, heads, enc_layers, dec_layers, V, params)
