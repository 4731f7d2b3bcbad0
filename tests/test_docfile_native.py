"""
Native docfile reader (tmvb_docfile_read) against the Python mirror of readcorp
(src/Corpus.jl:277-299): same packed CSR for every switch combination, the reference's error message for a
block that does not parse, defaults of 1 for missing counts / ratings.  Host-only: runs without a GPU.
"""
import numpy as np
import pytest


def _packed_via_python(tmvb, path, **kw):
    corp = tmvb.readcorp(docfile=path, **kw)
    return tmvb.PackedCorpus.from_corpus(corp)


@pytest.mark.parametrize("counts,readers,ratings", [(False, False, False), (True, False, False), (True, True, False), (True, True, True)])
def test_matches_python_readcorp(tmvb, tmp_path, counts, readers, ratings):
    pc0 = tmvb.syn_citeu(M=120, V=300, U=40, seed=2)
    rng = np.random.default_rng(1)
    pc0.ratings = rng.integers(1, 6, size=pc0.nR).astype(np.int32)
    corp = pc0.to_corpus()
    path = str(tmp_path / "docs.txt")
    tmvb.writecorp(corp, docfile=path, counts=counts, readers=readers, ratings=ratings)
    want = _packed_via_python(tmvb, path, counts=counts, readers=readers, ratings=ratings)
    got = tmvb.readcorp_packed(path, counts=counts, readers=readers, ratings=ratings, condense=False, V=want.V, U=want.U)
    for name in ("doc_ptr", "terms", "counts", "rdr_ptr", "readers", "ratings"):
        assert np.array_equal(getattr(got, name), getattr(want, name)), name
    if not counts:
        assert np.all(got.counts == 1)
    if readers and not ratings:
        assert np.all(got.ratings == 1)


def test_condense_merges_duplicate_terms(tmvb, tmp_path):
    path = tmp_path / "dup.txt"
    path.write_text("5,2,5,9,2,5\n1,1,2,1,3,1\n7\n4\n")          # two documents, counts on
    pc = tmvb.readcorp_packed(str(path), counts=True)
    assert pc.M == 2 and pc.V == 9
    assert pc.doc_ptr.tolist() == [0, 3, 4]
    assert pc.terms.tolist() == [1, 4, 8, 6] and pc.counts.tolist() == [4, 4, 1, 4]      # 0-based ids, counts added
    raw = tmvb.readcorp_packed(str(path), counts=True, condense=False)
    assert raw.terms.tolist() == [4, 1, 4, 8, 1, 4, 6]


def test_other_delimiter_and_crlf(tmvb, tmp_path):
    path = tmp_path / "tab.txt"
    path.write_bytes(b"3\t1\t2\r\n10\r\n")
    pc = tmvb.readcorp_packed(str(path), delim="\t")
    assert pc.M == 2 and pc.terms.tolist() == [0, 1, 2, 9]


@pytest.mark.parametrize("text,doc,line", [("1,2\n1,x\n3\n1\n", 1, 1), ("1,2\n1,1\n3,0\n1,1\n", 2, 3), ("1,2\n1\n", 1, 1)])
def test_bad_block_raises_reference_message(tmvb, tmp_path, text, doc, line):
    path = tmp_path / "bad.txt"
    path.write_text(text)
    with pytest.raises(tmvb.CorpusError, match=f"document {doc} beginning on line {line} failed to load."):
        tmvb.readcorp_packed(str(path), counts=True)


def test_missing_file(tmvb, tmp_path):
    with pytest.raises(tmvb.CorpusError):
        tmvb.readcorp_packed(str(tmp_path / "nope.txt"))
