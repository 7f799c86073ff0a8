"""Text side of the translation path: sentence splitting,
SentencePiece vocabulary, segment wrapping and the annotated source / target text
a Response carries. Host code only -- it feeds `slimt_hip_translate` token ids and
turns the returned ids back into text.

Mirrors the reference's interface (same names, argument meaning and results):
  Range, AnnotatedText        slimt/Annotation.hh:16-230, Annotation.cc:15-85
  Vocabulary                  slimt/Vocabulary.cc:24-104 (encode -> ids + byte views, decode)
  Splitter, sentence_stream   slimt/Splitter.cc:23-46,125-245 (rules), :247-360 (modes)
  TextProcessor               slimt/TextProcessor.cc:66-199 (tokenize, process, wrap)

All ranges are BYTE offsets into the UTF-8 text (the reference's Encoding::Byte);
`AnnotatedText.to(Encoding.UTF8)` converts them to code-point offsets as
Annotation.cc:87-160 does. SentencePiece itself comes from the `sentencepiece`
wheel (the reference links the same library, Vocabulary.hh:9).
"""
from __future__ import annotations

import enum
import functools
from dataclasses import dataclass
from typing import Iterator, List, Optional, Sequence, Tuple

import regex


class Encoding(enum.IntEnum):  # Types.hh (Encoding::Byte / Encoding::UTF8)
    Byte = 0
    UTF8 = 1


@dataclass(frozen=True)
class Range:
    begin: int = 0
    end: int = 0

    def size(self) -> int:
        return self.end - self.begin

    def __repr__(self) -> str:  # bindings/python/slimt.cpp:151-154
        return "{%d, %d}" % (self.begin, self.end)


class _Lazy:
    """A sentence whose token ranges are computed when first asked for."""
    __slots__ = ("begin", "end", "count", "resolve")

    def __init__(self, begin: int, end: int, count: int, resolve):
        self.begin, self.end, self.count, self.resolve = begin, end, count, resolve


class AnnotatedText:
    """A text blob with sentence and (sub)word boundaries. Between two sentences lies
    a gap (whitespace the translation keeps); gap i precedes sentence i and gap
    sentence_count() follows the last one (Annotation.hh:16-60)."""

    def __init__(self, text: str | bytes = b""):
        self.data: bytes = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        # per sentence: token begin offsets + sentence end, or a _Lazy that produces them on
        # first use (ranges cost a protobuf walk per piece; translating needs only the ids)
        self._sent: List[object] = []
        self._encoding = Encoding.Byte
        self._cp: Optional[List[int]] = None

    # -- the reference's accessors -----------------------------------------------------------
    @property
    def text(self) -> str:
        return self.data.decode("utf-8", errors="replace")

    def sentence_count(self) -> int:
        return len(self._sent)

    @property
    def _words(self) -> "_WordsView":
        return _WordsView(self)

    def _offsets(self, sentence_id: int) -> List[int]:
        w = self._sent[sentence_id]
        if isinstance(w, _Lazy):
            offs = w.resolve()
            if len(offs) != w.count + 1:
                raise RuntimeError("lazy annotation resolved to a different token count")
            offs[0], offs[-1] = w.begin, w.end  # the bounds it was recorded with (gaps depend on them)
            self._sent[sentence_id] = w = offs
        return w

    def _bounds(self, sentence_id: int) -> Tuple[int, int]:
        w = self._sent[sentence_id]
        return (w.begin, w.end) if isinstance(w, _Lazy) else (w[0], w[-1])

    def word_count(self, sentence_id: int) -> int:
        w = self._sent[sentence_id]
        return w.count if isinstance(w, _Lazy) else len(w) - 1

    def _range(self, b: int, e: int) -> Range:
        if self._encoding == Encoding.UTF8:
            cp = self._codepoints()
            return Range(cp[b], cp[e])
        return Range(b, e)

    def word_as_range(self, sentence_id: int, word_id: int) -> Range:
        w = self._offsets(sentence_id)
        return self._range(w[word_id], w[word_id + 1])

    def sentence_as_range(self, sentence_id: int) -> Range:
        return self._range(*self._bounds(sentence_id))

    def _gap_bytes(self, gap_id: int) -> Tuple[int, int]:
        b = self._bounds(gap_id - 1)[1] if gap_id > 0 else 0
        e = self._bounds(gap_id)[0] if gap_id < len(self._sent) else len(self.data)
        return b, e

    def gap_as_range(self, gap_id: int) -> Range:
        return self._range(*self._gap_bytes(gap_id))

    def word(self, sentence_id: int, word_id: int) -> str:
        w = self._offsets(sentence_id)
        return self.data[w[word_id]:w[word_id + 1]].decode("utf-8", errors="replace")

    def sentence(self, sentence_id: int) -> str:
        b, e = self._bounds(sentence_id)
        return self.data[b:e].decode("utf-8", errors="replace")

    def gap(self, gap_id: int) -> str:
        b, e = self._gap_bytes(gap_id)
        return self.data[b:e].decode("utf-8", errors="replace")

    def gap_bytes(self, gap_id: int) -> bytes:
        b, e = self._gap_bytes(gap_id)
        return self.data[b:e]

    def to(self, encoding: Encoding) -> None:
        """Ranges in bytes or in code points (Annotation.cc:87-160)."""
        self._encoding = Encoding(encoding)

    def _codepoints(self) -> List[int]:
        """cp[i] = code points that start before byte i (token boundaries are lead bytes)."""
        if self._cp is None or len(self._cp) != len(self.data) + 1:
            cp, n = [0] * (len(self.data) + 1), 0
            for i, c in enumerate(self.data):
                cp[i] = n
                if (c & 0xC0) != 0x80:
                    n += 1
            cp[len(self.data)] = n
            self._cp = cp
        return self._cp

    # -- construction --------------------------------------------------------------------------
    def record_existing_sentence(self, token_ranges: Sequence[Tuple[int, int]], sentence_begin: int) -> None:
        """A sentence that already lies in the text: contiguous (begin, end) byte ranges of
        its tokens (Annotation.cc:54-85). An empty list records an empty sentence."""
        if token_ranges:
            offs = [b for b, _ in token_ranges] + [token_ranges[-1][1]]
            for (b0, e0), (b1, _) in zip(token_ranges, token_ranges[1:]):
                if e0 != b1:
                    raise ValueError("tokens of a sentence must be contiguous")
        else:
            offs = [sentence_begin]
        self._check_order(offs[0], offs[-1])
        self._sent.append(offs)

    def record_lazy_sentence(self, begin: int, end: int, count: int, resolve) -> None:
        """record_existing_sentence for a sentence of `count` tokens in [begin, end) whose token
        boundaries `resolve()` returns (count + 1 ascending offsets) when somebody asks."""
        self._check_order(begin, end)
        self._sent.append(_Lazy(begin, end, count, resolve))

    def _check_order(self, begin: int, end: int) -> None:
        last = self._bounds(len(self._sent) - 1)[1] if self._sent else 0
        if begin < last or end > len(self.data) or end < begin:
            raise ValueError("sentence outside the text or before the previous one")

    def append_sentence(self, prefix: bytes | str, tokens: Sequence[bytes]) -> None:
        """Append gap text, then a sentence given as its token strings (Annotation.cc:20-43)."""
        self.append_ending_whitespace(prefix)
        offs, o = [], len(self.data)
        for t in tokens:
            offs.append(o)
            o += len(t)
        offs.append(o)
        self.data += b"".join(tokens)
        self._sent.append(offs)
        self._cp = None

    def append_lazy_sentence(self, prefix: bytes | str, sentence: bytes, count: int, resolve) -> None:
        """append_sentence with the sentence's text given whole; `resolve()` returns the token
        boundaries relative to the sentence (count + 1 offsets, 0 .. len(sentence))."""
        self.append_ending_whitespace(prefix)
        begin = len(self.data)
        self.data += sentence
        self._sent.append(_Lazy(begin, begin + len(sentence), count, lambda: [begin + o for o in resolve()]))
        self._cp = None

    def append_ending_whitespace(self, whitespace: bytes | str) -> None:
        self.data += whitespace.encode("utf-8") if isinstance(whitespace, str) else whitespace
        self._cp = None


class _WordsView:
    """Read access to the resolved token boundaries of every sentence (a list of lists)."""

    def __init__(self, owner: AnnotatedText):
        self.owner = owner

    def __len__(self) -> int:
        return len(self.owner._sent)

    def __getitem__(self, sentence_id: int) -> List[int]:
        if sentence_id < 0:
            sentence_id += len(self.owner._sent)
        return self.owner._offsets(sentence_id)


# ----------------------------------------------------------------------------------------------
class Vocabulary:
    """SentencePiece model: text <-> ids with the byte range of every piece."""

    def __init__(self, model: str | bytes):
        import sentencepiece
        if isinstance(model, (bytes, bytearray, memoryview)):  # Vocabulary(View), Vocabulary.cc:24-27
            self.sp = sentencepiece.SentencePieceProcessor(model_proto=bytes(model))
        else:  # Vocabulary(path), Vocabulary.cc:29-32
            self.sp = sentencepiece.SentencePieceProcessor(model_file=model)

    def size(self) -> int:
        return self.sp.get_piece_size()

    def eos_id(self) -> int:
        return self.sp.eos_id()

    def pad_id(self) -> int:
        return self.sp.pad_id()

    def encode(self, line: bytes | str, add_eos: bool = False) -> Tuple[List[int], List[Tuple[int, int]]]:
        """ids and, per id, the (begin, end) byte range of its surface in `line`
        (Vocabulary.cc:34-75; the appended EOS has no range, as there)."""
        raw = line.encode("utf-8") if isinstance(line, str) else bytes(line)
        proto = self.sp.encode(raw.decode("utf-8", errors="replace"), out_type="proto")
        words = [p.id for p in proto.pieces]
        views = [(p.begin, p.end) for p in proto.pieces]
        if add_eos:
            words.append(self.eos_id())
        return words, views

    def encode_batch(self, lines: Sequence[bytes], num_threads: int = 0):
        """encode() of many lines on SentencePiece's own thread pool (the reference
        tokenises sentence by sentence, TextProcessor.cc:111-124; this is where the text
        side gets its throughput from)."""
        texts = [l.decode("utf-8", errors="replace") for l in lines]
        protos = self.sp.encode(texts, out_type="proto", num_threads=num_threads or None)
        return [([p.id for p in pr.pieces], [(p.begin, p.end) for p in pr.pieces]) for pr in protos]

    def encode_ids_batch(self, lines: Sequence[str], num_threads: int = 0) -> List[List[int]]:
        """ids only (no ranges): SentencePiece's batch call, one C++ pass on its thread pool."""
        return self.sp.encode(list(lines), out_type=int, num_threads=num_threads or None)

    def decode_text_batch(self, sentences: Sequence[Sequence[int]], num_threads: int = 0) -> List[str]:
        return self.sp.decode([w if type(w) is list else (w.tolist() if hasattr(w, "tolist") else list(map(int, w)))
                               for w in sentences], num_threads=num_threads or None)

    def encode_boundaries(self, line: str) -> List[int]:
        """Token boundaries (byte offsets into `line`, one more than tokens) of encode(line)."""
        proto = self.sp.encode(line, out_type="proto")
        return [p.begin for p in proto.pieces] + [proto.pieces[-1].end if proto.pieces else 0]

    def decode_boundaries(self, words: Sequence[int]) -> List[int]:
        """Token boundaries of decode(words) in the decoded text (the EOS piece is empty)."""
        proto = self.sp.decode([int(w) for w in words], out_type="proto")
        return [p.begin for p in proto.pieces] + [proto.pieces[-1].end if proto.pieces else 0]

    def decode(self, words: Sequence[int], ignore_eos: bool = False) -> Tuple[bytes, List[Tuple[int, int]]]:
        """Decoded text and the byte range of every piece in it; the EOS piece is an empty
        range at the end and is dropped with ignore_eos (Vocabulary.cc:77-104)."""
        proto = self.sp.decode([int(w) for w in words], out_type="proto")
        views = [(p.begin, p.end) for p in proto.pieces]
        if ignore_eos and views:
            views.pop()
        return proto.text.encode("utf-8"), views


# ----------------------------------------------------------------------------------------------
_EOS_MARKS = ".?!։。？！"  # . ? ! Armenian full stop, CJK 。？！
_CANDIDATE = regex.compile(
    r"(?P<prefix>[\p{L}\p{N}]*)(?P<punct>[" + regex.escape(_EOS_MARKS) + r"]+)"
    r"(?P<tail>['\")\]’”\p{Pf}]*(?:\[\p{Nd}+[\p{Nd},\s]*\p{Nd}\])?['\")\]’”\p{Pf}]*)"
    r"(?P<ws>\s*)", regex.UNICODE)
# the same candidate without its prefix group: `search` then jumps from mark to mark (a leading
# [\p{L}\p{N}]* makes it try a letter run at every position of every word); the prefix -- needed only
# for a single '.' before an upper-case letter or a digit -- is read backwards from the mark
_MARK = regex.compile(
    r"(?P<punct>[" + regex.escape(_EOS_MARKS) + r"]+)"
    r"(?P<tail>['\")\]’”\p{Pf}]*(?:\[\p{Nd}+[\p{Nd},\s]*\p{Nd}\])?['\")\]’”\p{Pf}]*)"
    r"(?P<ws>\s*)", regex.UNICODE)
_PREFIX_CHAR = regex.compile(r"[\p{L}\p{N}]", regex.UNICODE)
_NEXT_WORD = regex.compile(r"[^\s\p{L}\p{N}\p{M}\p{S}]*\s*(?P<lead>[\p{L}\p{M}\p{N}]*)", regex.UNICODE)
_LOWER = regex.compile(r"\p{M}*\p{Ll}", regex.UNICODE)
_UPPER = regex.compile(r"\p{M}*[\p{Lu}\p{Lt}]", regex.UNICODE)
_DIGIT = regex.compile(r"[\p{Nd}\p{Nl}]", regex.UNICODE)
_OTHER = regex.compile(r"\p{M}*\p{Lo}", regex.UNICODE)


class Splitter:
    """Rule-based sentence boundary detection with a list of non-breaking prefixes.

    A boundary is a run of sentence-final punctuation (+ closing quotes / brackets / a
    footnote mark) followed by whitespace (none needed after 。！？), unless what follows
    starts in lower case, or the mark is a single '.' after a known prefix ("Mr.", or a
    NUMERIC_ONLY prefix before a digit: "No. 5"), or it is a bracketed ellipsis "[...]".
    Marks inside a token ("a.b.c", "3.14") have no whitespace after them and never break.
    (Behaviour of Splitter.cc:125-245.)"""

    def __init__(self, prefix_file: str = ""):
        self.prefix_type = {}
        if prefix_file:
            self.load(prefix_file)

    def load(self, fname: str) -> None:
        with open(fname, "rb") as f:
            self.load_from_serialized(f.read())

    def load_from_serialized(self, buffer: bytes | str) -> None:
        text = buffer.decode("utf-8", errors="replace") if isinstance(buffer, (bytes, bytearray)) else buffer
        for line in text.splitlines():  # "<prefix> [#NUMERIC_ONLY#]", '#' starts a comment (Splitter.cc:30-46)
            m = regex.match(r"([^#\s]*)\s*(#\s*NUMERIC_ONLY\s*#)?", line)
            if m and m.group(1):
                self.prefix_type[m.group(1)] = 2 if m.group(2) else 1

    def prefix_class(self, prefix: str) -> int:
        return self.prefix_type.get(prefix, 0)

    def split(self, paragraph: str) -> Iterator[Tuple[int, int]]:
        """(begin, end) character ranges of the sentences of one paragraph."""
        n = len(paragraph)
        pos = 0

        def prefix(m, floor):     # the letters / digits right before the mark (not before `floor`: where this
            b = m.start("punct")  # candidate search began, as the one-regex form's match does)
            while b > floor:
                ch = paragraph[b - 1]
                if not (ch.isalnum() if ch < "\x80" else _PREFIX_CHAR.match(ch, concurrent=False)):
                    break
                b -= 1
            return paragraph[b:m.start("punct")]

        while True:
            while pos < n and paragraph[pos].isspace():
                pos += 1
            if pos >= n:
                return
            start, scan, end = pos, pos, None
            while end is None:
                # concurrent=False: by default the regex module releases the interpreter lock around every
                # match on a str; with the service's other Python threads waiting for it, getting it back cost
                # more than the match (5.6 us per search against 1.7 us alone)
                floor = scan
                m = _MARK.search(paragraph, scan, concurrent=False)
                if not m:
                    break
                scan = m.end()
                punct, ws = m.group("punct"), m.group("ws")
                if not ws and punct not in ("。", "！", "？"):
                    scan = m.end("punct")
                    continue

                # what the next word starts with decides (1 caseless letter, 2 lower, 3 upper / title, 4 digit,
                # 0 anything else). ASCII letter or digit right after the whitespace: no regex needed
                nxt = paragraph[scan] if scan < n else ""
                if "a" <= nxt <= "z":
                    continue
                if "A" <= nxt <= "Z":
                    kind = 3
                elif "0" <= nxt <= "9":
                    kind = 4
                else:
                    lead = _NEXT_WORD.match(paragraph, scan, concurrent=False).group("lead")
                    kind = 1 if _OTHER.match(lead) else 2 if _LOWER.match(lead) else 3 if _UPPER.match(lead) else \
                        4 if _DIGIT.match(lead) else 0
                if kind == 2:
                    continue
                if kind == 3:
                    if punct == "." and self.prefix_class(prefix(m, floor)) != 0:
                        continue
                elif kind == 4:
                    if punct == "." and self.prefix_class(prefix(m, floor)) == 2:
                        continue
                elif kind == 0:
                    if punct == "..." and m.group("tail") == "]" and m.start("punct") > start + 1 and \
                            paragraph[m.start("punct") - 1] == "[":
                        continue
                end = m.start("ws")
            if end is None:  # no boundary left: the rest, right-trimmed, is the last sentence
                end = n
                while end > start and paragraph[end - 1].isspace():
                    end -= 1
                yield start, end
                return
            yield start, end
            pos = end


def sentence_stream(text: bytes, splitter: Splitter, mode: str) -> Iterator[Tuple[int, int]]:
    """(begin, end) byte ranges of the sentences of `text` (SentenceStream, Splitter.cc:301-360).
    mode: "sentence" = one sentence per line; "paragraph" = one paragraph per line, split by the
    rules; "wrapped_text" = paragraphs separated by blank lines, line breaks inside a paragraph
    are ordinary whitespace. Empty lines yield empty sentences, as the reference's stream does
    (TextProcessor.cc:118 drops them after tokenisation)."""
    if mode not in ("sentence", "paragraph", "wrapped_text"):
        raise ValueError("Unknown ssplitmode %r, choose one of sentence, paragraph, wrapped_text" % mode)
    if mode == "wrapped_text":
        paragraphs = []
        for m in regex.finditer(rb"(?:[^\n]|\n(?![\r\n]))+", text):
            paragraphs.append((m.start(), m.end()))
    else:
        paragraphs, pos = [], 0
        while pos < len(text):
            nl = text.find(b"\n", pos)
            end = len(text) if nl < 0 else nl
            e = end
            while e > pos and text[e - 1:e] == b"\r":
                e -= 1
            paragraphs.append((pos, e))
            pos = end + 1
    for b, e in paragraphs:
        if mode == "sentence":
            yield b, e
            continue
        para = text[b:e].decode("utf-8", errors="replace")
        if len(para) == e - b:  # ASCII: character offsets are byte offsets
            for cb, ce in splitter.split(para):
                yield b + cb, b + ce
            continue
        for cb, ce in splitter.split(para):  # character -> byte offsets of the two cuts
            sb = b + len(para[:cb].encode("utf-8"))
            yield sb, sb + len(para[cb:ce].encode("utf-8"))


# ----------------------------------------------------------------------------------------------
def _source_boundaries(vocabulary, begin: int, line: str) -> List[int]:
    """Token boundaries (byte offsets in the text) of a sentence that starts at `begin`, EOS included."""
    offs = [begin + o for o in vocabulary.encode_boundaries(line)]
    return offs + [offs[-1]]  # the EOS: empty, at the end


class TextProcessor:
    """text -> (annotated source, segments of token ids ending in EOS) (TextProcessor.cc:78-199)."""

    def __init__(self, mode: str, vocabulary: Vocabulary, prefixes: bytes | str = b""):
        if mode not in ("sentence", "paragraph", "wrapped_text"):
            raise ValueError("Unknown ssplitmode %r, choose one of sentence, paragraph, wrapped_text" % mode)
        self.mode = mode
        self.vocabulary = vocabulary
        self.splitter = Splitter()
        if prefixes:
            self.splitter.load_from_serialized(prefixes)

    def process(self, text: str | bytes, wrap_length: int, num_threads: int = 0) -> Tuple[AnnotatedText, List[List[int]]]:
        """Split, tokenise, wrap every sentence into segments of at most wrap_length ids
        including the EOS appended to each (TextProcessor.cc:101-163)."""
        return self.process_many([text], wrap_length, num_threads)[0]

    def process_many(self, texts: Sequence[str | bytes], wrap_length: int, num_threads: int = 0):
        """process() of many texts with ONE SentencePiece batch call for all their sentences.
        Token ranges of the source annotation are resolved lazily (only a wrapped sentence
        needs them here, to find where its parts start)."""
        if wrap_length < 2:
            raise ValueError("wrap_length must leave room for one token and EOS")
        sources = [AnnotatedText(t) for t in texts]
        spans, lines = [], []
        for src in sources:
            sp = list(sentence_stream(src.data, self.splitter, self.mode))
            spans.append(sp)
            lines += [src.data[b:e].decode("utf-8", errors="replace") for b, e in sp]
        encoded = self.vocabulary.encode_ids_batch(lines, num_threads) if lines else []
        eos, step, v = self.vocabulary.eos_id(), wrap_length - 1, self.vocabulary
        out, k = [], 0

        for src, sp in zip(sources, spans):
            segments: List[List[int]] = []
            sent, last, size = src._sent, 0, len(src.data)
            for b, e in sp:
                words, line = encoded[k], lines[k]
                k += 1
                if not words:  # nothing after normalisation (TextProcessor.cc:118)
                    continue
                # the sentence is what its tokens cover: SentencePiece drops surrounding whitespace
                if line[0].isspace() or line[-1].isspace():
                    lead = len(line) - len(line.lstrip())
                    b += len(line[:lead].encode("utf-8"))
                    line = line.strip()
                    e = b + len(line.encode("utf-8"))
                if len(words) <= step:
                    # record_lazy_sentence, inlined (a fifth of this loop went into its calls)
                    if b < last or e > size or e < b:
                        raise ValueError("sentence outside the text or before the previous one")
                    words.append(eos)  # SentencePiece's own list: nobody else holds it
                    sent.append(_Lazy(b, e, len(words), functools.partial(_source_boundaries, v, b, line)))
                    last = e
                    segments.append(words)
                    continue
                bounds = [b + o for o in v.encode_boundaries(line)]
                for off in range(0, len(words), step):
                    part = bounds[off:off + step + 1]
                    src.record_existing_sentence(list(zip(part[:-1], part[1:])) + [(part[-1], part[-1])], part[0])
                    segments.append(list(words[off:off + step]) + [eos])
                last = e
            out.append((src, segments))
        return out

    def process_annotated(self, source: AnnotatedText) -> Tuple[AnnotatedText, List[List[int]]]:
        """Re-tokenise text whose sentences are already marked (the pivot's second hop): one
        segment per sentence, no wrapping (TextProcessor.cc:165-199)."""
        out = AnnotatedText(source.data)
        eos, v = self.vocabulary.eos_id(), self.vocabulary
        spans = [source._bounds(s) for s in range(source.sentence_count())]
        lines = [out.data[b:e].decode("utf-8", errors="replace") for b, e in spans]
        encoded = v.encode_ids_batch(lines) if lines else []
        segments = []
        for (b, e), line, words in zip(spans, lines, encoded):
            def resolve(b=b, e=e, line=line, n=len(words)):
                offs = [b + o for o in v.encode_boundaries(line)] if n else [e]
                return offs + [offs[-1]]
            if words:
                out.record_lazy_sentence(b, e, len(words) + 1, resolve)
            else:
                out.record_existing_sentence([(e, e)], e)
            segments.append(list(words) + [eos])
        return out, segments
