"""the text side (slimt_amd/text.py) and the reference's Python
API over the HIP engine (slimt_amd/frontend.py).

CPU tests: vocabulary views, sentence splitting modes and rules, segment wrapping,
annotation bookkeeping, alignment remapping. GPU test: text in -> text out through
Service.translate / .pivot, tokens equal the oracle's translation of the same
segments, alignment rows equal its alignment rows."""
import io
import random

import numpy as np
import pytest

from slimt_amd import text


@pytest.fixture(scope="module")
def corpus():
    rnd = random.Random(7)
    words = ["".join(rnd.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(2, 8))) for _ in range(1500)]
    words += ["über", "naïve", "日本", "語"]
    sents = []
    for _ in range(6000):
        s = " ".join(rnd.choice(words) for _ in range(rnd.randint(3, 18)))
        sents.append(s[0].upper() + s[1:] + rnd.choice(".?!"))
    return sents


@pytest.fixture(scope="module")
def spm_model(corpus):
    """A 512-piece unigram model trained here on synthetic text (eos = 0 like the reference's
    vocabularies, no pad / bos)."""
    import sentencepiece
    out = io.BytesIO()
    sentencepiece.SentencePieceTrainer.train(sentence_iterator=iter(corpus), model_writer=out, vocab_size=512,
                                             model_type="unigram", pad_id=-1, unk_id=1, bos_id=-1, eos_id=0,
                                             minloglevel=2)
    return out.getvalue()


def test_vocabulary_views_cover_the_line(spm_model, corpus):
    v = text.Vocabulary(spm_model)
    assert v.size() == 512 and v.eos_id() == 0
    for line in corpus[:50] + ["Zwei über drei. 日本語 ok"]:
        raw = line.encode()
        words, views = v.encode(raw, add_eos=True)
        assert words[-1] == v.eos_id() and len(views) == len(words) - 1
        assert views[0][0] == 0 and views[-1][1] == len(raw)
        assert all(e0 == b1 for (_, e0), (b1, _) in zip(views, views[1:]))  # contiguous
        decoded, dviews = v.decode(words)
        assert len(dviews) == len(words) and dviews[-1][0] == dviews[-1][1] == len(decoded)  # EOS: empty, at the end
        assert len(v.decode(words, ignore_eos=True)[1]) == len(words) - 1
    a = v.encode_batch([c.encode() for c in corpus[:64]], num_threads=4)
    assert all(a[i] == v.encode(corpus[i]) for i in range(64))


def test_splitter_rules():
    sp = text.Splitter()
    sp.load_from_serialized("Mr\nDr\nNo #NUMERIC_ONLY#\n# a comment line\n")
    assert sp.prefix_class("Mr") == 1 and sp.prefix_class("No") == 2 and sp.prefix_class("Xy") == 0

    def split(t):
        return [t[b:e] for b, e in sp.split(t)]

    assert split("Hello world. This is Mr. Smith speaking! Is it? yes it is.  Trailing  ") == \
        ["Hello world.", "This is Mr. Smith speaking!", "Is it? yes it is.", "Trailing"]
    assert split("See No. 5 now. No. Five. Dr. 7 left.") == ["See No. 5 now.", "No. Five.", "Dr.", "7 left."]
    assert split("He said \"Stop.\" Then left [1]. Next one... and on.") == \
        ["He said \"Stop.\"", "Then left [1].", "Next one... and on."]
    assert split("a.b.c is 3.14 today. Fine [...] then. Done") == ["a.b.c is 3.14 today.", "Fine [...] then.", "Done"]
    assert split("你好。世界！Ok") == ["你好。", "世界！", "Ok"]  # no whitespace needed after CJK marks
    assert split("   ") == [] and split("") == []


def test_splitter_mark_search_equals_the_one_regex_form():
    """Splitter.split finds its candidates with the prefix-free _MARK pattern (linear) and reads the
    prefix backwards; the restatement's original form, one regex with a leading [\\p{L}\\p{N}]* prefix
    group (text._CANDIDATE), must give the same cuts on every text."""
    import random
    from slimt_amd import text as T
    sp = T.Splitter()
    sp.load_from_serialized("Mr\nDr\nNo #NUMERIC_ONLY#\nSt\nvs\ne.g\nU.S\n")

    def one_regex_split(paragraph):
        n, pos = len(paragraph), 0
        while True:
            while pos < n and paragraph[pos].isspace():
                pos += 1
            if pos >= n:
                return
            start, scan, end = pos, pos, None
            while end is None:
                m = T._CANDIDATE.search(paragraph, scan)
                if not m:
                    break
                scan = m.end()
                punct, ws = m.group("punct"), m.group("ws")
                if not ws and punct not in ("。", "！", "？"):
                    scan = m.end("punct")
                    continue
                lead = T._NEXT_WORD.match(paragraph, m.end()).group("lead")
                if T._OTHER.match(lead):
                    pass
                elif T._LOWER.match(lead):
                    continue
                elif T._UPPER.match(lead):
                    if punct == "." and sp.prefix_class(m.group("prefix")) != 0:
                        continue
                elif T._DIGIT.match(lead):
                    if punct == "." and sp.prefix_class(m.group("prefix")) == 2:
                        continue
                else:
                    if punct == "..." and m.group("tail") == "]" and m.start("punct") > start + 1 and \
                            paragraph[m.start("punct") - 1] == "[":
                        continue
                end = m.start("ws")
            if end is None:
                end = n
                while end > start and paragraph[end - 1].isspace():
                    end -= 1
                yield start, end
                return
            yield start, end
            pos = end

    rnd = random.Random(3)
    atoms = ["Mr", "Dr", "No", "St", "vs", "e.g", "U.S", "Hello", "world", "x", "3", "14", "a.b", "ab", "Ünï", "日本", "γ", "Z"]
    puncts = [".", "?", "!", "...", ".)", ".\"", "。", "！", "? ", ". ", ".  ", " ", " ", " ", "[...]", " [...] ", ".[1]",
              ".[12, 3] ", "'", "”", ",", "\t"]
    for _ in range(4000):
        s = "".join(rnd.choice(atoms) + rnd.choice(puncts) for _ in range(rnd.randint(1, 14)))
        assert list(sp.split(s)) == list(one_regex_split(s)), s


def test_sentence_stream_modes():
    sp = text.Splitter()
    data = "Line one. Line two.\r\n\nPara two\nwrapped here. End.\n".encode()

    def run(mode):
        return [data[b:e] for b, e in text.sentence_stream(data, sp, mode)]

    assert run("sentence") == [b"Line one. Line two.", b"", b"Para two", b"wrapped here. End."]
    assert run("paragraph") == [b"Line one.", b"Line two.", b"Para two", b"wrapped here.", b"End."]
    assert run("wrapped_text") == [b"Line one.", b"Line two.", b"Para two\nwrapped here.", b"End."]
    with pytest.raises(ValueError):
        list(text.sentence_stream(data, sp, "lines"))


def test_text_processor_wraps_and_annotates(spm_model):
    v = text.Vocabulary(spm_model)
    tp = text.TextProcessor("paragraph", v)
    src = "  First sentence here. Second one is a bit longer than the first!\n\nÜber naïve 日本語?  "
    ann, segments = tp.process(src, wrap_length=8)
    assert ann.text == src
    assert ann.sentence_count() == len(segments) >= 4  # the long sentence is wrapped
    raw = src.encode()
    rebuilt = b""
    for s in range(ann.sentence_count()):
        seg = segments[s]
        assert 2 <= len(seg) <= 8 and seg[-1] == v.eos_id() and v.eos_id() not in seg[:-1]
        assert ann.word_count(s) == len(seg)  # one range per id, the EOS one is empty
        assert ann.word_as_range(s, len(seg) - 1).size() == 0
        r = ann.sentence_as_range(s)
        assert raw[r.begin:r.end].decode() == ann.sentence(s)
        words = b"".join(raw[ann.word_as_range(s, w).begin:ann.word_as_range(s, w).end] for w in range(len(seg)))
        assert words == raw[r.begin:r.end]
        rebuilt += ann.gap(s).encode() + raw[r.begin:r.end]
    assert rebuilt + ann.gap(ann.sentence_count()).encode() == raw  # gaps + sentences tile the text
    # unwrapped: ids of a sentence == the vocabulary's ids of its text
    ann2, seg2 = tp.process("One short line.", wrap_length=128)
    assert seg2 == [v.encode("One short line.", add_eos=True)[0]]
    # code-point ranges
    ann.to(text.Encoding.UTF8)
    last = ann.sentence_as_range(ann.sentence_count() - 1)
    assert src[last.begin:last.end] == ann.sentence(ann.sentence_count() - 1)
    with pytest.raises(ValueError):
        tp.process("x", wrap_length=1)
    with pytest.raises(ValueError):
        text.TextProcessor("lines", v)


def test_annotated_text_append_and_ranges():
    t = text.AnnotatedText()
    t.append_sentence("  ", [b"Hal", b"lo", b""])
    t.append_sentence(" ", ["wör".encode(), b"ld", b""])
    t.append_ending_whitespace("\n")
    assert t.text == "  Hallo wörld\n" and t.sentence_count() == 2
    assert t.word_count(0) == 3 and t.word(1, 0) == "wör" and t.sentence(1) == "wörld"
    assert t.gap(0) == "  " and t.gap(1) == " " and t.gap(2) == "\n"
    assert t.word_as_range(1, 1) == text.Range(12, 14)  # bytes: ö is two
    t.to(text.Encoding.UTF8)
    assert t.word_as_range(1, 1) == text.Range(11, 13)
    assert repr(text.Range(3, 9)) == "{3, 9}"
    with pytest.raises(ValueError):
        text.AnnotatedText("abcdef").record_existing_sentence([(0, 2), (3, 4)], 0)  # not contiguous


def test_alignment_transfer_keeps_probability_mass():
    from slimt_amd import frontend
    R = text.Range
    # first model's pivot tokens: "ab" "cd" "ef" + EOS; second model's tokenisation: "a" "bcde" "f" + EOS
    q1 = [R(0, 2), R(2, 4), R(4, 6), R(6, 6)]
    q2 = [R(0, 1), R(1, 5), R(5, 6), R(6, 6)]
    rng = np.random.default_rng(3)
    p = rng.random((5, 4)).astype(np.float32)
    p /= p.sum(axis=1, keepdims=True)
    out = np.asarray(frontend.transfer_through_characters(q1, q2, p.tolist()))
    assert out.shape == (5, 4)
    assert np.allclose(out.sum(axis=1), 1.0, atol=1e-6)
    want0 = p[:, 0] + p[:, 1] * 0.25  # "a" + the 'b' quarter of "bcde"
    assert np.allclose(out[:, 0], want0, atol=1e-6)
    assert np.allclose(out[:, 3], p[:, 3], atol=1e-6)  # EOS meets EOS
    same = np.asarray(frontend.transfer_through_characters(q1, q1, p.tolist()))
    assert np.allclose(same, p, atol=1e-7)
    # a surface-less piece in front (SentencePiece's bare "▁") keeps its mass
    q3 = [R(0, 0)] + q2
    p3 = rng.random((5, 5)).astype(np.float32)
    p3 /= p3.sum(axis=1, keepdims=True)
    out3 = np.asarray(frontend.transfer_through_characters(q1, q3, p3.tolist()))
    assert np.allclose(out3.sum(axis=1), 1.0, atol=1e-6)
    assert np.allclose(out3[:, 0], p3[:, 0] + p3[:, 1] + p3[:, 2] * 0.25, atol=1e-6)
    # no EOS on the first side: the second side's EOS mass is shared out
    out2 = np.asarray(frontend.transfer_through_characters(q1[:3], q2, p.tolist()))
    assert np.allclose(out2.sum(axis=1), 1.0, atol=1e-6)


def test_service_batches_by_token_budget():
    from slimt_amd import frontend
    svc = frontend.Service(workers=1, max_words=64)
    try:
        units = [frontend._Unit(0, i, [1] * n) for i, n in enumerate([3, 30, 7, 16, 16, 2, 9, 31, 5])]
        batches = svc._batches(units)
        assert sorted(u.index for b in batches for u in b) == list(range(9))
        for b in batches:
            width = max(len(u.words) for u in b)
            assert len(b) * width <= 64 and len(b[-1].words) == width
            assert [len(u.words) for u in b] == sorted(len(u.words) for u in b)
        # the reference's walk (Batcher.cc:95-120): lengths ascending, (count + 1) * length <= max_words
        assert [[len(u.words) for u in b] for b in batches] == [[2, 3, 5, 7, 9], [16, 16], [30, 31]]
        # ... the same batches the C++ LengthQueue forms (oracle.batcher_generate restates the reference)
        from oracle import oracle as O
        want = O.batcher_generate([[len(u.words) for u in units]], 64, 32, 1.5)
        assert [[(u.request, u.index) for u in b] for b in batches] == want
        # a segment longer than the engine takes (a pivot's second hop) travels in pieces that end in EOS
        long = list(range(5, 5 + 300)) + [0]
        pieces = svc._split_long(long, 0)
        assert [len(p) for p in pieces] == [128, 128, 47] and all(p[-1] == 0 for p in pieces)
        assert [w for p in pieces for w in p[:-1]] == long[:-1]
        assert svc._split_long([4, 5, 0], 0) == [[4, 5, 0]]
    finally:
        svc.close()


@pytest.mark.gpu
def test_service_translate_and_pivot_text_to_text(hip, oracle, spm_model, corpus):
    """Text -> Response through the reference's Python API on a synthetic 512-piece model whose
    vocabulary is the trained SentencePiece model: per sentence, the target ids are the oracle's
    greedy translation of the same segment and the alignment rows are its alignment rows."""
    from slimt_amd import frontend, synth
    m = synth.make_model("micro", eos_bias=3.0)  # V = 512 = the vocabulary's size
    blob = synth.make_lexical_shortlist(m.V, m.V, frequent=32, best=8, seed=5)
    package = frontend.Package(model=synth.write_bin(m), vocabulary=spm_model, shortlist=blob)
    cfg = frontend.Config(encoder_layers=m.enc_layers, decoder_layers=m.dec_layers, num_heads=m.H,
                          split_mode="paragraph")
    model = frontend.Model(cfg, package, device=0)
    svc = frontend.Service(workers=2, max_words=256, wrap_length=24)
    try:
        texts = [" ".join(corpus[i:i + 3]) + "\n" + corpus[i + 3] for i in range(0, 40, 4)]
        texts.append("")
        responses = svc.translate(model, texts, html=False, encoding=frontend.Encoding.Byte)
        assert len(responses) == len(texts)
        om = oracle.OracleModel(m)
        osl = oracle.OracleShortlist(blob, m.V, m.V)
        oracle.set_mode(oracle.PORTABLE)
        n_sent = 0
        for t, r in zip(texts, responses):
            assert r.source.text == t
            src, segments = model.processor.process(t, 24)
            assert r.source.sentence_count() == r.target.sentence_count() == len(segments) == len(r.alignments)
            for s, seg in enumerate(segments):
                # structure here; exact tokens / alignments per batch below (the shortlist is a
                # function of the batch a sentence travelled in)
                assert r.target.word_count(s) == len(r.alignments[s]) >= 1
                assert all(len(row) == len(seg) for row in r.alignments[s])
                assert r.target.gap(s) == r.source.gap(s)
                n_sent += 1
        assert n_sent >= 40
        # exact parity: rebuild the service's batches and compare every sentence with the oracle
        per_request = [model.processor.process(t, 24)[1] for t in texts]
        units = [frontend._Unit(ri, i, seg) for ri, segs in enumerate(per_request) for i, seg in enumerate(segs)]
        for batch in svc._batches(units):
            B, S = len(batch), max(len(u.words) for u in batch)
            ids = np.zeros((B, S), np.uint32)
            lens = np.zeros(B, np.uint32)
            for i, u in enumerate(batch):
                ids[i, :len(u.words)] = u.words
                lens[i] = len(u.words)
            sl = osl.generate(ids, lens)
            w_out, w_len, w_al, _ = om.translate(ids, lens, sl, 1.5, 0, want_align=True)
            for i, u in enumerate(batch):
                r = responses[u.request]
                n, L = int(w_len[i]), int(lens[i])
                decoded, views = model.vocabulary.decode(w_out[i, :n].tolist())
                assert r.target.sentence(u.index) == decoded.decode()
                assert r.target.word_count(u.index) == n
                assert np.array_equal(np.asarray(r.alignments[u.index], np.float32), w_al[i, :n, :L])
        oracle.set_mode(oracle.FAITHFUL)
        # pivot: source -> pivot -> target with the same model twice; alignments stay distributions
        piv = svc.pivot(model, model, texts[:4])
        for t, r in zip(texts[:4], piv):
            assert r.source.text == t and r.source.sentence_count() == r.target.sentence_count()
            for s in range(r.source.sentence_count()):
                a = np.asarray(r.alignments[s])
                assert a.shape == (r.target.word_count(s), r.source.word_count(s))
                assert np.allclose(a.sum(axis=1), 1.0, atol=1e-4)
        with pytest.raises(NotImplementedError):
            svc.translate(model, ["<p>x</p>"], html=True)
    finally:
        svc.close()
        model.close()
