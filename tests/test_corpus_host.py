"""Host-side corpus substrate: docfile format (src/Corpus.jl:277-325), check_corp rules, CSR packing, sharding."""
import numpy as np
import pytest


def test_readcorp_roundtrip_matches_reference_format(tmvb, tmp_path):
    # README :54-70 / src/Corpus.jl:289: blocks of terms / counts / readers / ratings lines, 1-based ids
    p = tmp_path / "docs.txt"
    p.write_text("1,3,5\n2,1,4\n2,7\n1,1\n4\n9\n\n\n")
    corp = tmvb.readcorp(docfile=str(p), counts=True, readers=True, ratings=True)
    assert len(corp) == 2
    assert corp[0].terms.tolist() == [1, 3, 5] and corp[0].counts.tolist() == [2, 1, 4]
    assert corp[0].readers.tolist() == [2, 7] and corp[1].readers.tolist() == []
    corp.vocab = {i: str(i) for i in range(1, 6)}
    corp.users = {i: str(i) for i in range(1, 8)}
    tmvb.check_corp(corp)
    pc = tmvb.PackedCorpus.from_corpus(corp)
    assert pc.terms.tolist() == [0, 2, 4, 3] and pc.doc_ptr.tolist() == [0, 3, 4]
    assert pc.readers.tolist() == [1, 6] and pc.rdr_ptr.tolist() == [0, 2, 2]
    out = tmp_path / "out.txt"
    tmvb.writecorp(corp, docfile=str(out), counts=True, readers=True, ratings=True)
    corp2 = tmvb.readcorp(docfile=str(out), counts=True, readers=True, ratings=True)
    assert all(np.array_equal(a.terms, b.terms) and np.array_equal(a.ratings, b.ratings) for a, b in zip(corp, corp2))


def test_check_doc_and_check_corp_rules(tmvb):
    with pytest.raises(tmvb.DocumentError):
        tmvb.Document(terms=[0, 1])                       # src/Corpus.jl:42
    with pytest.raises(tmvb.DocumentError):
        tmvb.Document(terms=[1, 2], counts=[1])           # :44
    corp = tmvb.Corpus([tmvb.Document(terms=[1, 9])], vocab=["a", "b"])
    with pytest.raises(tmvb.CorpusError):
        tmvb.check_corp(corp)                             # :117 term key not in vocab


def test_shard_bounds_are_nnz_balanced_and_cover(tmvb):
    pc = tmvb.syn_nsf(M=3000, V=2000, seed=3)
    for ws in (1, 2, 3, 8):
        b = pc.shard_bounds(ws)
        assert b[0][0] == 0 and b[-1][1] == pc.M and all(b[r][1] == b[r + 1][0] for r in range(ws - 1))
        nn = [pc.doc_ptr[e] - pc.doc_ptr[s] for s, e in b]
        assert max(nn) - min(nn) <= 2 * pc.N.max()
    sh = pc.shard(*pc.shard_bounds(2)[1])
    assert sh.doc_ptr[0] == 0 and sh.nnz == pc.doc_ptr[pc.M] - pc.doc_ptr[pc.shard_bounds(2)[1][0]]


def test_synthetic_corpora_are_deterministic_and_condensed(tmvb):
    a = tmvb.syn_nsf(M=500, V=3000, seed=11); b = tmvb.syn_nsf(M=500, V=3000, seed=11)
    assert np.array_equal(a.terms, b.terms) and np.array_equal(a.counts, b.counts)
    for d in range(a.M):
        t = a.terms[a.doc_ptr[d]:a.doc_ptr[d + 1]]
        assert np.all(np.diff(t) > 0)                     # unique + sorted (condense_corp!, src/Corpus.jl:523)
    c = tmvb.syn_citeu(M=400, V=1000, U=300, seed=5)
    for d in range(c.M):
        r = c.readers[c.rdr_ptr[d]:c.rdr_ptr[d + 1]]
        assert len(r) >= 1 and np.all(np.diff(r) > 0) and r.max() < 300
