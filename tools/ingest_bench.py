#!/usr/bin/env python
"""Corpus ingest AT SIZE (round-4 review, missing #7): the synthetic corpora of the benched configurations written in the
reference's docfile format (src/Corpus.jl:277-325: per document one line of 1-based term ids, one of counts, one of reader ids), read back by the native
parser (tmvb_docfile_read, csrc/tmvb_core.hip -- the call a Julia `readcorp` replacement binds) and compared BYTE FOR BYTE with the generator's packed
CSR; the parser's throughput in MB/s of docfile text.  Host-only: no GPU involved.

    python tools/ingest_bench.py [nsf] [citeu] [--docs N]        (default: both, full size)  ->  one JSON line per corpus
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmvb_amd

tm = tmvb_amd.pkg


def write_docfile(pc, path, readers=False):
    """the reference's format, counts on (and readers on for CiteULike): ids 1-based, comma separated, one line per field per document"""
    t1 = (pc.terms.astype(np.int64) + 1).astype(str)
    c = pc.counts.astype(str)
    r1 = (pc.readers.astype(np.int64) + 1).astype(str) if readers else None
    with open(path, "w") as f:
        out = []
        for d in range(pc.M):
            a, b = int(pc.doc_ptr[d]), int(pc.doc_ptr[d + 1])
            out.append(",".join(t1[a:b])); out.append(",".join(c[a:b]))
            if readers:
                ra, rb = int(pc.rdr_ptr[d]), int(pc.rdr_ptr[d + 1])
                out.append(",".join(r1[ra:rb]))
            if len(out) >= 30000:
                f.write("\n".join(out) + "\n"); out = []
        if out:
            f.write("\n".join(out) + "\n")


def round_trip(name, pc, readers, tmpdir, repeats=3):
    path = os.path.join(tmpdir, f"{name}docs.txt")
    t0 = time.perf_counter()
    write_docfile(pc, path, readers=readers)
    t_write = time.perf_counter() - t0
    size = os.path.getsize(path)
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        got = tm.readcorp_packed(path, counts=True, readers=readers, ratings=False, condense=True, V=pc.V, U=pc.U if readers else None)
        best = min(best, time.perf_counter() - t0)
    same = {n: bool(np.array_equal(getattr(got, n), getattr(pc, n))) for n in ("doc_ptr", "terms", "counts")}
    if readers:
        same.update({n: bool(np.array_equal(getattr(got, n), getattr(pc, n))) for n in ("rdr_ptr", "readers")})
        same["ratings_all_one"] = bool(np.all(got.ratings == 1))               # ratings default to 1 (src/Corpus.jl:21)
    os.remove(path)
    return {"corpus": name, "documents": int(pc.M), "V": int(pc.V), "U": int(pc.U) if readers else 0, "nnz": int(pc.nnz), "nR": int(pc.nR) if readers else 0,
            "docfile_MB": round(size / 1e6, 2), "python_write_s": round(t_write, 2), "native_read_s": round(best, 4),
            "native_read_MBps": round(size / 1e6 / best, 1), "tokens_per_s": round((pc.nnz + (pc.nR if readers else 0)) / best), "csr_identical": all(same.values()), "fields": same}


def main():
    argv = sys.argv[1:]
    docs = int(argv[argv.index("--docs") + 1]) if "--docs" in argv else None
    which = [a for a in argv if a in ("nsf", "citeu")] or ["nsf", "citeu"]
    with tempfile.TemporaryDirectory() as td:
        for w in which:
            if w == "nsf":
                pc = tm.syn_nsf(M=docs) if docs else tm.syn_nsf()
                r = round_trip("nsf", pc, False, td)
            else:
                pc = tm.syn_citeu(M=docs) if docs else tm.syn_citeu()
                r = round_trip("citeu", pc, True, td)
            print(json.dumps(r), flush=True)
            if not r["csr_identical"]:
                sys.exit(1)


if __name__ == "__main__":
    main()
