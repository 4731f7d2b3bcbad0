"""
Ingest at size: tools/ingest_bench.py writes the benched synthetic corpora in the reference's docfile format
(src/Corpus.jl:277-325) and reads them back through the native parser; here on 2 000-document slices of both corpora (CPU, seconds): the packed CSR must
be byte-identical to the generator's, readers and default ratings included.  The full-size figures are in profiles/r5_ingest.jsonl.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_docfile_round_trip_of_both_benched_corpora(tmvb, tmp_path):
    import ingest_bench as ib
    r = ib.round_trip("nsf", tmvb.syn_nsf(M=2000), False, str(tmp_path), repeats=1)
    assert r["csr_identical"] and r["documents"] == 2000 and r["nnz"] > 100000 and r["native_read_MBps"] > 1.0, r
    r = ib.round_trip("citeu", tmvb.syn_citeu(M=2000), True, str(tmp_path), repeats=1)
    assert r["csr_identical"] and r["fields"]["ratings_all_one"] and r["nR"] > 0, r
