"""The reference's Python API over the HIP engine: what
`bindings/python/slimt.cpp:144-221` exposes -- Package, Config, preset, Model,
Service.translate / .pivot, Response, AnnotatedText, Range, Encoding -- with one
addition, the `device` the model's weights live on.

    from slimt_amd import frontend as slimt
    model = slimt.Model(slimt.preset.tiny(), slimt.Package(model="model.bin",
                        vocabulary="vocab.spm", shortlist="lex.s2t.bin"), device=0)
    service = slimt.Service(workers=4)
    responses = service.translate(model, ["Hello world. How are you?"], html=False)
    responses[0].target.text, responses[0].alignments

Text in, text out: TextProcessor (text.py) splits and tokenises, the C++ batching
service (host/Service.{hh,cc} through include/slimt_hip_service.h: token-budget batches
of at most `max_words` padded tokens formed by the reference's rule, Batcher.cc:95-120,
double-buffered pinned workers, the batch's lexical shortlist generated on the device)
translates, and the returned ids / alignment rows become the Response
(Request.cc:136-170). The compute is the C-ABI library's; nothing here falls back
to a CPU model.
"""
from __future__ import annotations

import functools
import threading
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from . import capi, synth
from .text import AnnotatedText, Encoding, Range, TextProcessor, Vocabulary  # noqa: F401 (API surface)
from .text import _Lazy

Alignment = np.ndarray  # [target token][source token] f32, p(source | target); indexes like a list of rows


@dataclass
class Package:  # slimt.cpp:182-191: paths (or bytes) of the three model files
    model: str | bytes = ""
    vocabulary: str | bytes = ""
    shortlist: str | bytes = ""
    ssplit: str | bytes = ""  # optional non-breaking-prefix list (Package.hh ssplit)


@dataclass
class Config:  # Model::Config, Model.hh:33-51
    encoder_layers: int = 6
    decoder_layers: int = 2
    feed_forward_depth: int = 2
    num_heads: int = 8
    split_mode: str = "sentence"


class preset:  # Model.cc:206-245
    @staticmethod
    def tiny() -> Config:
        return Config(6, 2, 2, 8, "sentence")

    @staticmethod
    def base() -> Config:
        return Config(6, 2, 2, 8, "sentence")

    @staticmethod
    def nano() -> Config:
        return Config(4, 2, 2, 8, "sentence")


@dataclass
class Response:  # Response.hh: source / target with sentence + token ranges, soft alignments
    source: AnnotatedText = field(default_factory=AnnotatedText)
    target: AnnotatedText = field(default_factory=AnnotatedText)
    alignments: List[Alignment] = field(default_factory=list)

    def to(self, encoding: Encoding) -> None:
        self.source.to(encoding)
        self.target.to(encoding)


def _target_boundaries(vocabulary, begin: int, words) -> List[int]:
    """Token boundaries (byte offsets in the target text) of a decoded sentence that starts at `begin`."""
    return [begin + o for o in vocabulary.decode_boundaries(words)]


def _blob(x) -> bytes:
    if isinstance(x, (bytes, bytearray, memoryview)):
        return bytes(x)
    with open(x, "rb") as f:
        return f.read()


class Model:
    """slimt::Model (Model.hh:31-83): vocabulary + text processor + the transformer's weights on
    `device` + the optional lexical shortlist generator."""

    _next_id = 0

    def __init__(self, config: Config, package: Package, device: int = 0):
        self.config = config
        self.id = Model._next_id
        Model._next_id += 1
        self.vocabulary = Vocabulary(package.vocabulary)
        self.processor = TextProcessor(config.split_mode, self.vocabulary,
                                       _blob(package.ssplit) if package.ssplit else b"")
        host = synth.model_from_bin(_blob(package.model), heads=config.num_heads,
                                    enc_layers=config.encoder_layers, dec_layers=config.decoder_layers)
        if self.vocabulary.size() > host.V:
            raise ValueError("vocabulary has %d pieces, the model's embedding %d rows"
                             % (self.vocabulary.size(), host.V))
        self.device = device
        capi.request_hw_queues(32)  # a Service runs `workers` x 2 streams on this device (no-op once HIP is up)
        self.engine = capi.Model(host, device)
        self.dims = (host.D, host.F, host.V)
        self.shortlist_generator = None
        self.shortlist_blob = b""
        if package.shortlist:  # Model::make_shortlist_generator, Model.cc:60-72
            self.shortlist_blob = _blob(package.shortlist)
            # ShortlistGenerator(view, source, target): shared = false, check = false (Model.cc:73-80, Shortlist.hh:47-52)
            self.shortlist_generator = capi.ShortlistGenerator(self.shortlist_blob, host.V, host.V,
                                                               shared=False, check=False, device=device)

    def close(self) -> None:
        if self.shortlist_generator is not None:
            self.shortlist_generator.close()
            self.shortlist_generator = None
        self.engine.close()


class _LazyAlignment:
    """One sentence's [target tokens, source tokens] matrix inside a call's block of alignment rows;
    materialised (a reshaped view, no copy) on first use. Indexes, iterates and converts like the
    array it stands for."""
    __slots__ = ("_block", "_a", "_b", "_rows", "_cols", "_arr")

    def __init__(self, block, a, b, rows, cols):
        self._block, self._a, self._b, self._rows, self._cols, self._arr = block, a, b, rows, cols, None

    def array(self) -> np.ndarray:
        if self._arr is None:
            self._arr = self._block[self._a:self._b].reshape(self._rows, self._cols) if self._b > self._a else \
                np.zeros((self._rows, self._cols), np.float32)
        return self._arr

    def __array__(self, dtype=None, copy=None):
        a = self.array()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self._rows

    def __getitem__(self, i):
        return self.array()[i]

    def __iter__(self):
        return iter(self.array())

    @property
    def shape(self):
        return (self._rows, self._cols)


@dataclass
class _Unit:  # one segment of one request
    request: int
    index: int
    words: List[int]


class Service:
    """slimt::Async (Frontend.hh:20-78) behind the binding's blocking calls (slimt.cpp:44-135)."""

    def __init__(self, workers: int = 1, cache_size: int = 0, max_words: int = 1024, wrap_length: int = 128,
                 tgt_length_limit_factor: float = 1.5):
        if workers < 1:
            raise ValueError("workers must be >= 1")
        self.workers = workers
        self.cache_size = cache_size  # accepted for signature parity; no translation cache here
        self.max_words = max_words
        self.wrap_length = wrap_length
        self.limit_factor = tgt_length_limit_factor
        self.pipeline_documents = 256  # translate(): documents per pipelined chunk (one chunk: no pipeline)
        self._engines = {}  # model id -> capi.BatchService (host/Service over its C ABI)
        self._lock = threading.Lock()

    # -- batching ---------------------------------------------------------------------------------
    def _batches(self, units: Sequence[_Unit]) -> List[List[_Unit]]:
        """The reference's batch-forming rule (Batcher::generate, slimt/Batcher.cc:95-120), as the C++
        Service's LengthQueue applies it: walk the lengths UPWARDS, arrival order inside a length, and
        keep adding sentences while (count + 1) * length fits the word budget -- a batch is as wide as
        the last sentence added, and a sentence's padding (hence its length limit floor(1.5 S)) is the
        same here, in host/Service.cc and in the reference."""
        order = sorted(units, key=lambda u: (len(u.words), u.request, u.index))
        out, cur = [], []
        for u in order:
            if cur and (len(cur) + 1) * len(u.words) > self.max_words:
                out.append(cur)
                cur = []
            cur.append(u)
        if cur:
            out.append(cur)
        return out

    # the engine's longest source (the reference wraps at 128, Frontend.hh:27, and its batcher leaves
    # slack above that for the rare longer segment, Batcher.cc:83-88)
    ENGINE_LIMIT = 128

    def _split_long(self, words: List[int], eos: int) -> List[List[int]]:
        """A segment longer than the engine takes (only a pivot's second hop can produce one: its
        input is the first model's output re-tokenised, not wrapped text) goes through in pieces of
        at most ENGINE_LIMIT tokens, every piece ending in EOS."""
        if len(words) <= self.ENGINE_LIMIT:
            return [words]
        body = list(words[:-1]) if words and words[-1] == eos else list(words)
        step = self.ENGINE_LIMIT - 1
        return [body[i:i + step] + [eos] for i in range(0, len(body), step)]

    def _engine(self, model: Model) -> "capi.BatchService":
        """The C++ batching service for `model` (host/Service.{hh,cc} behind include/slimt_hip_service.h):
        token-budget batches under the rule of _batches, `workers` double-buffered workers with pinned
        staging, the batch's lexical shortlist generated on the device (Model.cc:117-120). Units,
        batches, padding and result routing used to be Python objects; they are C++ now, and one call
        carries a whole translate()."""
        with self._lock:
            eng = self._engines.get(model.id)
            if eng is None:
                V = model.dims[2]
                eng = capi.BatchService([model.engine], max_words=max(self.max_words, 1),
                                        wrap_length=min(self.ENGINE_LIMIT, max(self.max_words, 1)),
                                        limit_factor=self.limit_factor, workers_per_device=self.workers, pad_id=0,
                                        eos_id=model.vocabulary.eos_id(), alignments=True,
                                        lexical_shortlist=model.shortlist_blob, source_vocab=V, target_vocab=V,
                                        shared_vocab=False, check=False)  # as Model.cc:73-80 constructs it
                self._engines[model.id] = eng
        return eng

    def _translate_segments(self, model: Model, per_request: List[List[List[int]]]):
        eos = model.vocabulary.eos_id()
        flat, owner = [], []  # the sentences of the one request the C++ service gets; (request, index, piece)
        pieces_of = {}
        limit = self.ENGINE_LIMIT
        if all(len(seg) <= limit for segs in per_request for seg in segs):  # the usual call: nothing to split
            flat = [seg for segs in per_request for seg in segs]
        else:
            for r, segs in enumerate(per_request):
                for i, seg in enumerate(segs):
                    if len(seg) <= limit:
                        flat.append(seg)
                        owner.append((r, i, 0))
                        continue
                    pieces = self._split_long(list(seg), eos)
                    pieces_of[(r, i)] = len(pieces)
                    for k, piece in enumerate(pieces):
                        flat.append(piece)
                        owner.append((r, i, k))
        histories = [[None] * len(segs) for segs in per_request]
        if not flat:
            return histories
        res = self._engine(model).translate(flat)
        targets, t_off = res.targets.copy(), res.target_offsets.astype(np.int64)
        align, a_off = res.alignments.copy(), res.align_offsets.astype(np.int64)
        src_len = res.source_lengths
        res.close()
        # per sentence: (target ids, alignment); the alignment is cut out of the call's one block when
        # somebody asks for it (_LazyAlignment): most callers read a few, and 24,000 reshaped views
        # per call were a tenth of a second of interpreter time
        parts = {}
        words_all = targets.tolist()
        to, ao, sl = t_off.tolist(), a_off.tolist(), src_len.tolist()
        if not pieces_of:
            n = 0
            for hist in histories:
                for i in range(len(hist)):
                    words = words_all[to[n]:to[n + 1]]
                    hist[i] = (words, _LazyAlignment(align, ao[n], ao[n + 1], len(words), sl[n]))
                    n += 1
            return histories
        for n, (r, i, k) in enumerate(owner):
            words = words_all[to[n]:to[n + 1]]
            alignment = _LazyAlignment(align, ao[n], ao[n + 1], len(words), sl[n])
            if (r, i) in pieces_of:
                parts[(r, i, k)] = (words, alignment.array(), sl[n])
            else:
                histories[r][i] = (words, alignment)
        for (r, i), n in pieces_of.items():
            # a split segment: targets concatenated (inner EOS dropped), alignment rows block-diagonal
            # over the pieces' source tokens (the pieces' inner EOS columns dropped, the last one kept)
            got = [parts[(r, i, k)] for k in range(n)]
            widths = [L - 1 for _, _, L in got[:-1]] + [got[-1][2]]
            total = sum(widths)
            out_words, rows, col = [], [], 0
            for k, (words, alignment, L) in enumerate(got):
                last = k == n - 1
                keep = len(words) if last or not len(words) or words[-1] != eos else len(words) - 1
                for t in range(keep):
                    row = np.zeros(total, np.float32)
                    row[col:col + widths[k]] = np.asarray(alignment[t], np.float32)[:widths[k]]
                    rows.append(row)
                out_words.extend(int(w) for w in words[:keep])
                col += widths[k]
            histories[r][i] = (out_words, np.stack(rows) if rows else np.zeros((0, total), np.float32))
        return histories

    # -- the binding's calls -----------------------------------------------------------------------
    def _respond_many(self, model: Model, sources: Sequence[AnnotatedText], histories) -> List[Response]:
        """Request::complete (Request.cc:136-170) for a whole call: every sentence's ids decoded in
        one SentencePiece batch, the source's gaps kept, target token ranges resolved on demand."""
        flat = [words for hist in histories for words, _ in hist]  # lists of ids
        decoded = model.vocabulary.decode_text_batch(flat, self.workers) if flat else []
        out, k, v = [], 0, model.vocabulary
        for source, hist in zip(sources, histories):
            # AnnotatedText.append_lazy_sentence per sentence, written out: the target text is joined once per
            # response, sentences recorded directly (this loop holds the interpreter lock while the
            # engine's threads wait for the next chunk)
            resp = Response(source=source)
            parts, sent, alignments = [], resp.target._sent, resp.alignments
            src_sent, data, prev_end, pos = source._sent, source.data, 0, 0
            for s, (words, alignment) in enumerate(hist):
                w = src_sent[s]
                lazy = type(w) is _Lazy
                gap = data[prev_end:(w.begin if lazy else w[0])]  # source.gap_bytes(s)
                prev_end = w.end if lazy else w[-1]
                text = decoded[k].encode("utf-8")
                begin = pos + len(gap)
                pos = begin + len(text)
                parts.append(gap)
                parts.append(text)
                sent.append(_Lazy(begin, pos, len(words), functools.partial(_target_boundaries, v, begin, words)))
                alignments.append(alignment)
                k += 1
            parts.append(data[prev_end:] if hist else data)  # the gap behind the last sentence
            resp.target.data = b"".join(parts)
            out.append(resp)
        return out

    def translate(self, model: Model, texts: Sequence[str], html: bool = False,
                  encoding: Encoding = Encoding.UTF8) -> List[Response]:
        if html:
            raise NotImplementedError("HTML markup transfer is outside the ported path (SURVEY.md §2)")
        # Large calls go through in chunks of documents, pipelined: while the engine translates chunk k
        # (a C call: no interpreter lock held), this thread splits and tokenises chunk k + 1 and
        # assembles the responses of chunk k - 1. Batches are formed per chunk (as they are per
        # arrival window in the reference's Async service).
        chunk = self.pipeline_documents
        if len(texts) <= chunk:
            processed = model.processor.process_many(texts, self.wrap_length, self.workers)
            histories = self._translate_segments(model, [segs for _, segs in processed])
            out = self._respond_many(model, [src for src, _ in processed], histories)
        else:
            from concurrent.futures import ThreadPoolExecutor
            out, pending = [], []
            with ThreadPoolExecutor(max_workers=2) as pool:
                def finish(item):
                    processed, fut = item
                    out.extend(self._respond_many(model, [src for src, _ in processed], fut.result()))
                for k in range(0, len(texts), chunk):
                    processed = model.processor.process_many(texts[k:k + chunk], self.wrap_length, self.workers)
                    pending.append((processed, pool.submit(self._translate_segments, model, [segs for _, segs in processed])))
                    while len(pending) > 2:
                        finish(pending.pop(0))
                while pending:
                    finish(pending.pop(0))
        for r in out:
            r.to(encoding)
        return out

    def pivot(self, first: Model, second: Model, texts: Sequence[str], html: bool = False) -> List[Response]:
        """source -> pivot with `first`, pivot -> target with `second`, sentence for sentence;
        alignments are marginalised over the pivot tokens (Response.cc:13-195)."""
        if html:
            raise NotImplementedError("HTML markup transfer is outside the ported path (SURVEY.md §2)")
        firsts = self.translate(first, texts, encoding=Encoding.Byte)
        second_in = [second.processor.process_annotated(r.target) for r in firsts]
        histories = self._translate_segments(second, [segs for _, segs in second_in])
        seconds = self._respond_many(second, [src for src, _ in second_in], histories)
        return [combine(r1, r2) for r1, r2 in zip(firsts, seconds)]

    def close(self) -> None:
        for eng in self._engines.values():
            eng.close()
        self._engines.clear()


# -------------------------------------------------------------------------------------------------
def transfer_through_characters(source_side_pivots: Sequence[Range], target_side_pivots: Sequence[Range],
                                pivot_given_targets: Alignment) -> Alignment:
    """p(q' | t) over the second model's pivot tokens q' -> p(q | t) over the first model's
    pivot tokens q: each q' spreads its probability evenly over its bytes, each q collects
    what falls into its range; a trailing zero-width q' (EOS) is shared out evenly
    (Response.cc:13-116)."""
    T, nq = len(pivot_given_targets), len(source_side_pivots)
    out = [[0.0] * nq for _ in range(T)]
    sq = qt = 0
    while sq < nq and qt < len(target_side_pivots):
        a, b = source_side_pivots[sq], target_side_pivots[qt]
        if a.begin == b.begin and a.end == b.end:
            for t in range(T):
                out[t][sq] += pivot_given_targets[t][qt]
            sq, qt = sq + 1, qt + 1
            continue
        if b.size() == 0:  # a piece without surface before the end (SentencePiece's bare "▁"): its mass
            for t in range(T):  # goes to the pivot token at its position (the reference asserts overlap
                out[t][sq] += pivot_given_targets[t][qt]  # here, Response.cc:49, and divides by 0)
            qt += 1
            continue
        if a.size() == 0:  # likewise on the first model's side: nothing to collect
            sq += 1
            continue
        left, right = max(a.begin, b.begin), min(a.end, b.end)
        if right > left:
            share = (right - left) / float(b.size())
            for t in range(T):
                out[t][sq] += share * pivot_given_targets[t][qt]
        if a.end == b.end:
            sq, qt = sq + 1, qt + 1
        elif a.end > b.end:
            qt += 1
        else:
            sq += 1
    while qt < len(target_side_pivots):  # what is left has no surface: EOS
        for t in range(T):
            gift = pivot_given_targets[t][qt] / max(1, nq)
            for s in range(nq):
                out[t][s] += gift
        qt += 1
    return out


def remap_alignments(first: Response, second: Response) -> List[Alignment]:
    """p(s | t) = sum_q p(s | q) p(q | t) per sentence (Response.cc:118-177)."""
    out = []
    for sid in range(first.source.sentence_count()):
        s_given_q, q_given_t = first.alignments[sid], second.alignments[sid]
        q1 = [first.target.word_as_range(sid, i) for i in range(first.target.word_count(sid))]
        q2 = [second.source.word_as_range(sid, i) for i in range(second.source.word_count(sid))]
        remapped = np.asarray(transfer_through_characters(q1, q2, q_given_t), np.float32).reshape(len(q_given_t), len(q1))
        sq = np.asarray(s_given_q, np.float32).reshape(len(q1), -1)
        out.append(remapped @ sq)
    return out


def combine(first: Response, second: Response) -> Response:  # Response.cc:179-191
    r = Response()
    if len(first.alignments):
        r.alignments = remap_alignments(first, second)
    r.source, r.target = first.source, second.target
    return r
