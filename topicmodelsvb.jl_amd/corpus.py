"""
Corpus substrate: the reference's Document / Corpus types (src/Corpus.jl:14-78), its docfile
format (readcorp, src/Corpus.jl:277-325; README :54-70) and the packed CSR layout the engine
uploads (the corpus half of update_buffer!, src/modelutils.jl:370-388).

Ids in `Document` and in docfiles are 1-based like the reference; `PackedCorpus` is 0-based.
Also holds the seeded synthetic corpora SYN-NSF / SYN-CITEU used because
the real nsfdocs.txt / citeudocs.txt are not part of the reference snapshot.
"""
from __future__ import annotations

import numpy as np

from ._lib import CorpusError, DocumentError


class Document:
    """src/Corpus.jl:14-26 (keyword constructor) + check_doc :41-50"""

    def __init__(self, terms=(), counts=None, readers=(), ratings=None, title=""):
        self.terms = np.asarray(terms, dtype=np.int64).reshape(-1)
        self.counts = np.ones(len(self.terms), dtype=np.int64) if counts is None else np.asarray(counts, dtype=np.int64).reshape(-1)
        self.readers = np.asarray(readers, dtype=np.int64).reshape(-1)
        self.ratings = np.ones(len(self.readers), dtype=np.int64) if ratings is None else np.asarray(ratings, dtype=np.int64).reshape(-1)
        self.title = title
        check_doc(self)

    def __len__(self):          # Base.length(doc) = length(doc.terms)
        return len(self.terms)

    def size(self):             # Base.size(doc) = sum(doc.counts)
        return int(self.counts.sum())


def check_doc(doc: Document):
    if not np.all(doc.terms > 0):
        raise DocumentError("all terms must be positive integers.")
    if not np.all(doc.counts > 0):
        raise DocumentError("all counts must be positive integers.")
    if len(doc.terms) != len(doc.counts):
        raise DocumentError("terms and counts vectors must have the same length.")
    if not np.all(doc.readers > 0):
        raise DocumentError("all readers must be positive integers.")
    if not np.all(doc.ratings > 0):
        raise DocumentError("all ratings must be positive integers.")
    if len(doc.readers) != len(doc.ratings):
        raise DocumentError("readers and ratings vectors must have the same length.")


class Corpus:
    """src/Corpus.jl:62-78.  vocab / users: dict key -> name, or a list of names (keys 1..n)."""

    def __init__(self, docs=None, vocab=None, users=None):
        self.docs = list(docs) if docs is not None else []
        self.vocab = self._as_dict(vocab)
        self.users = self._as_dict(users)
        for d, doc in enumerate(self.docs):
            try:
                check_doc(doc)
            except DocumentError:
                raise CorpusError(f"document {d + 1} failed check.")
        if any(k <= 0 for k in self.vocab):
            raise CorpusError("all vocab keys must be positive integers.")
        if any(k <= 0 for k in self.users):
            raise CorpusError("all user keys must be positive integers.")

    @staticmethod
    def _as_dict(x):
        if x is None:
            return {}
        if isinstance(x, dict):
            return dict(x)
        return {i + 1: str(v) for i, v in enumerate(x)}

    def __len__(self):
        return len(self.docs)

    def __iter__(self):
        return iter(self.docs)

    def __getitem__(self, d):
        return self.docs[d]

    def size(self):             # Base.size(corp) = (M, V, U)
        return len(self.docs), len(self.vocab), len(self.users)


def check_corp(corp: Corpus):
    """src/Corpus.jl:111-122"""
    for d, doc in enumerate(corp.docs):
        try:
            check_doc(doc)
        except DocumentError:
            raise CorpusError(f"document {d + 1} failed check.")
    V, U = len(corp.vocab), len(corp.users)
    for doc in corp.docs:
        if len(doc.terms) and (doc.terms.max() > V or any(int(t) not in corp.vocab for t in np.unique(doc.terms))):
            raise CorpusError("documents contain term keys not found in corpus vocabulary (see fixcorp! function).")
        if len(doc.readers) and (doc.readers.max() > U or any(int(r) not in corp.users for r in np.unique(doc.readers))):
            raise CorpusError("documents contain user keys not found in corpus users (see fixcorp! function).")
    if V != max(list(corp.vocab.keys()) + [0]):
        raise CorpusError("corpus vocab keys must form unit range starting at 1 (see fixcorp! function).")
    if U != max(list(corp.users.keys()) + [0]):
        raise CorpusError("corpus user keys must form unit range starting at 1 (see fixcorp! function).")


def readcorp(docfile="", vocabfile="", userfile="", titlefile="", delim=",", counts=False, readers=False, ratings=False):
    """readcorp (src/Corpus.jl:277-325): blocks of `counts+readers+ratings+1` delimited lines."""
    if ratings and not readers:
        ratings = False
    corp = Corpus()
    if docfile:
        block = 1 + int(counts) + int(readers) + int(ratings)
        with open(docfile) as f:
            lines = f.read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()
        names = [n for n, on in zip(("terms", "counts", "readers", "ratings"), (True, counts, readers, ratings)) if on]
        for d in range(0, len(lines), block):
            try:
                vals = [[int(p) for p in ln.split(delim)] if ln.strip() else [] for ln in lines[d:d + block]]
                corp.docs.append(Document(**dict(zip(names, vals))))
            except Exception:
                k = d // block + 1
                raise CorpusError(f"document {k} beginning on line {d + 1} failed to load.")
    if vocabfile:
        corp.vocab = _read_keyed(vocabfile)
    if userfile:
        corp.users = _read_keyed(userfile)
    if titlefile:
        with open(titlefile) as f:
            for doc, t in zip(corp.docs, f.read().split("\n")):
                doc.title = t
    return corp


def readcorp_packed(docfile, delim=",", counts=False, readers=False, ratings=False, condense=True, V=None, U=None):
    """The document part of readcorp (src/Corpus.jl:277-299) through the library's streaming parser
    (`tmvb_docfile_read`, include/tmvb.h): the reference's docfile format straight into the packed 0-based CSR, without
    per-document Python objects (NSF-sized files parse in well under a second).  condense=True merges equal term ids of
    a document (condense_corp!, src/Corpus.jl:523-531), the form the engine needs.  V / U default to the largest id seen."""
    import ctypes as C
    from ._lib import check, lib

    class _DocFile(C.Structure):
        _fields_ = [("M", C.c_int64), ("nnz", C.c_int64), ("nR", C.c_int64), ("V_seen", C.c_int64), ("U_seen", C.c_int64),
                    ("doc_ptr", C.POINTER(C.c_int64)), ("terms", C.POINTER(C.c_int32)), ("counts", C.POINTER(C.c_int32)),
                    ("rdr_ptr", C.POINTER(C.c_int64)), ("readers", C.POINTER(C.c_int32)), ("ratings", C.POINTER(C.c_int32))]

    f = _DocFile()
    L = lib()
    L.tmvb_docfile_free.restype = None
    check(L.tmvb_docfile_read(str(docfile).encode(), C.c_char(delim.encode()), C.c_int32(bool(counts)), C.c_int32(bool(readers)),
                              C.c_int32(bool(ratings)), C.c_int32(bool(condense)), C.byref(f)))
    try:
        arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[:int(n)].astype(dt, copy=True)
        pc = PackedCorpus(arr(f.doc_ptr, f.M + 1, np.int64), arr(f.terms, f.nnz, np.int32), arr(f.counts, f.nnz, np.int32),
                          int(V) if V is not None else int(f.V_seen),
                          arr(f.rdr_ptr, f.M + 1, np.int64), arr(f.readers, f.nR, np.int32), arr(f.ratings, f.nR, np.int32),
                          int(U) if U is not None else int(f.U_seen))
    finally:
        L.tmvb_docfile_free(C.byref(f))
    return pc


def _read_keyed(path):
    out = {}
    with open(path) as f:
        for ln in f:
            ln = ln.rstrip("\n")
            if not ln:
                continue
            k, _, v = ln.partition("\t")
            out[int(k)] = v
    if any(k <= 0 for k in out):
        raise CorpusError("all vocab keys must be positive integers.")
    return out


def writecorp(corp: Corpus, docfile="", delim=",", counts=False, readers=False, ratings=False):
    """Document part of writecorp (src/Corpus.jl:366-...)."""
    if ratings and not readers:
        ratings = False
    with open(docfile, "w") as f:
        for doc in corp.docs:
            f.write(delim.join(str(int(t)) for t in doc.terms) + "\n")
            if counts:
                f.write(delim.join(str(int(c)) for c in doc.counts) + "\n")
            if readers:
                f.write(delim.join(str(int(r)) for r in doc.readers) + "\n")
            if ratings:
                f.write(delim.join(str(int(r)) for r in doc.ratings) + "\n")


class PackedCorpus:
    """0-based CSR the engine uploads: doc_ptr i64[M+1], terms/counts i32[nnz] (+ readers)."""

    def __init__(self, doc_ptr, terms, counts, V, rdr_ptr=None, readers=None, ratings=None, U=0):
        self.doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
        self.terms = np.ascontiguousarray(terms, dtype=np.int32)
        self.counts = np.ascontiguousarray(counts, dtype=np.int32)
        self.M = len(self.doc_ptr) - 1
        self.V, self.U = int(V), int(U)
        if rdr_ptr is None:
            rdr_ptr = np.zeros(self.M + 1, dtype=np.int64)
            readers = np.zeros(0, dtype=np.int32)
            ratings = np.zeros(0, dtype=np.int32)
        self.rdr_ptr = np.ascontiguousarray(rdr_ptr, dtype=np.int64)
        self.readers = np.ascontiguousarray(readers, dtype=np.int32)
        self.ratings = np.ascontiguousarray(ratings, dtype=np.int32)

    @property
    def nnz(self):
        return int(self.doc_ptr[-1])

    @property
    def nR(self):
        return int(self.rdr_ptr[-1])

    @property
    def N(self):
        return np.diff(self.doc_ptr)

    @property
    def C(self):
        cs = np.concatenate([[0], np.cumsum(self.counts, dtype=np.int64)])
        return cs[self.doc_ptr[1:]] - cs[self.doc_ptr[:-1]]

    @classmethod
    def from_corpus(cls, corp: Corpus):
        M, V, U = corp.size()
        doc_ptr = np.zeros(M + 1, dtype=np.int64)
        rdr_ptr = np.zeros(M + 1, dtype=np.int64)
        for d, doc in enumerate(corp.docs):
            doc_ptr[d + 1] = doc_ptr[d] + len(doc.terms)
            rdr_ptr[d + 1] = rdr_ptr[d] + len(doc.readers)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)
        terms = cat([doc.terms for doc in corp.docs]) - 1        # src/modelutils.jl:371 `.- 1`
        counts = cat([doc.counts for doc in corp.docs])
        readers = cat([doc.readers for doc in corp.docs]) - 1
        ratings = cat([doc.ratings for doc in corp.docs])
        return cls(doc_ptr, terms, counts, V, rdr_ptr, readers, ratings, U)

    def to_corpus(self) -> Corpus:
        docs = []
        for d in range(self.M):
            a, b = self.doc_ptr[d], self.doc_ptr[d + 1]
            ra, rb = self.rdr_ptr[d], self.rdr_ptr[d + 1]
            docs.append(Document(terms=self.terms[a:b].astype(np.int64) + 1, counts=self.counts[a:b],
                                 readers=self.readers[ra:rb].astype(np.int64) + 1, ratings=self.ratings[ra:rb]))
        return Corpus(docs, vocab=[str(i + 1) for i in range(self.V)], users=[str(i + 1) for i in range(self.U)])

    def shard(self, d0: int, d1: int) -> "PackedCorpus":
        """Documents [d0, d1) as their own CSR (same vocabulary / users)."""
        a, b = self.doc_ptr[d0], self.doc_ptr[d1]
        ra, rb = self.rdr_ptr[d0], self.rdr_ptr[d1]
        return PackedCorpus(self.doc_ptr[d0:d1 + 1] - a, self.terms[a:b], self.counts[a:b], self.V,
                            self.rdr_ptr[d0:d1 + 1] - ra, self.readers[ra:rb], self.ratings[ra:rb], self.U)

    def shard_bounds(self, world_size: int):
        """Contiguous document ranges balanced by nnz + nR."""
        w = (self.doc_ptr + self.rdr_ptr).astype(np.float64) + np.arange(self.M + 1) * 1e-3
        tot = w[-1]
        cuts = [0]
        for r in range(1, world_size):
            cuts.append(int(np.searchsorted(w, tot * r / world_size)))
        cuts.append(self.M)
        return [(min(cuts[r], self.M), min(max(cuts[r + 1], cuts[r]), self.M)) for r in range(world_size)]


# ------------------------------------------------------------------------- synthetic corpora
def _condense(doc_ids, words, M, V):
    """(doc, word) token pairs -> CSR with unique sorted terms per document and their counts
    (the effect of condense_corp!, src/Corpus.jl:523)."""
    key = doc_ids.astype(np.int64) * V + words.astype(np.int64)
    uniq, cnt = np.unique(key, return_counts=True)
    d = uniq // V
    terms = (uniq - d * V).astype(np.int32)
    doc_ptr = np.zeros(M + 1, dtype=np.int64)
    np.add.at(doc_ptr, d + 1, 1)
    return np.cumsum(doc_ptr), terms, cnt.astype(np.int32)


def synthetic_lda_corpus(M, V, seed, Kstar=50, topic_conc=0.05, zipf_s=1.05, theta_conc=0.1,
                         len_mu=4.75, len_sigma=0.45, len_min=10, len_max=1000):
    """LDA generative process (cf. gendoc, src/modelutils.jl:594-612) with Zipf-tilted topics."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = 1.0 / np.arange(1, V + 1, dtype=np.float64) ** zipf_s
    rng.shuffle(base)
    topics = rng.gamma(topic_conc, size=(Kstar, V)) * base[None, :] + 1e-300
    topics /= topics.sum(axis=1, keepdims=True)
    theta = rng.gamma(theta_conc, size=(M, Kstar)) + 1e-300
    theta /= theta.sum(axis=1, keepdims=True)
    C = np.clip(np.rint(rng.lognormal(len_mu, len_sigma, size=M)), len_min, len_max).astype(np.int64)
    zc = rng.multinomial(C, theta)                      # M x Kstar topic counts per document
    docs_all, words_all = [], []
    doc_idx = np.arange(M, dtype=np.int64)
    for k in range(Kstar):
        nk = int(zc[:, k].sum())
        if nk == 0:
            continue
        cdf = np.cumsum(topics[k]); cdf[-1] = 1.0
        words_all.append(np.searchsorted(cdf, rng.random(nk), side="right").astype(np.int32))
        docs_all.append(np.repeat(doc_idx, zc[:, k]))
    doc_ptr, terms, counts = _condense(np.concatenate(docs_all), np.minimum(np.concatenate(words_all), V - 1), M, V)
    return doc_ptr, terms, counts


def syn_nsf(M=128804, V=25319, seed=20260928) -> PackedCorpus:
    """SYN-NSF: NSF-shaped corpus (M=128804, V=25319; README :34-36), SURVEY.md section 8d."""
    doc_ptr, terms, counts = synthetic_lda_corpus(M, V, seed)
    return PackedCorpus(doc_ptr, terms, counts, V)


def syn_citeu(M=16980, V=8000, U=5551, seed=20260929) -> PackedCorpus:
    """SYN-CITEU: CiteULike-shaped corpus with readers (README :38-41), ratings == 1."""
    doc_ptr, terms, counts = synthetic_lda_corpus(M, V, seed, Kstar=20, len_mu=4.5, len_sigma=0.5, len_max=600)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    # readers per document: 1 + heavy-tailed, mean ~ 12; at least 150 documents with one reader
    R = 1 + np.minimum(np.floor(rng.pareto(1.3, size=M) * 3.6), min(400, U - 1)).astype(np.int64)
    R[rng.choice(M, size=min(160, max(M // 10, 1)), replace=False)] = 1
    pop = 1.0 / np.arange(1, U + 1, dtype=np.float64)
    rng.shuffle(pop)
    pop /= pop.sum()
    rdr_ptr = np.concatenate([[0], np.cumsum(R)])
    # Zipf(1.0)-popular readers without replacement per document: Gumbel top-k in blocks
    readers = np.empty(rdr_ptr[-1], dtype=np.int32)
    logp = np.log(pop)
    order = np.argsort(R, kind="stable")
    pos = 0
    while pos < M:
        blk = order[pos:pos + 2048]
        kmax = int(R[blk].max())
        gk = logp[None, :] + rng.gumbel(size=(len(blk), U))
        top = np.argpartition(-gk, kth=min(kmax, U - 1), axis=1)[:, :kmax] if kmax < U else np.argsort(-gk, axis=1)
        # order the candidates by score so that the first R_d are the top R_d
        sc = np.take_along_axis(gk, top, axis=1)
        top = np.take_along_axis(top, np.argsort(-sc, axis=1), axis=1)
        for j, d in enumerate(blk):
            readers[rdr_ptr[d]:rdr_ptr[d + 1]] = np.sort(top[j, :R[d]])
        pos += len(blk)
    ratings = np.ones(rdr_ptr[-1], dtype=np.int32)
    return PackedCorpus(doc_ptr, terms, counts, V, rdr_ptr, readers, ratings, U)


def dirichlet_rows(K, V, seed=7):
    """K rows ~ Dirichlet(V, 1) (the reference's beta init, src/LDA.jl:35) as normalised Exp(1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    b = rng.standard_exponential(size=(K, V))
    return np.asfortranarray(b / b.sum(axis=1, keepdims=True))
