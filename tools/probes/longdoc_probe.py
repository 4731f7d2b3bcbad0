#!/usr/bin/env python
"""How does the LDA engine do on a corpus of LONG documents (every document on the LDS-tile kernel)?
Same token volume as SYN-NSF (~11 M), documents of ~400-800 unique terms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
from tmvb_amd_pkg.corpus import synthetic_lda_corpus
M = int(os.environ.get("M", 20000)); K = int(os.environ.get("K", 50)); V = 25319
doc_ptr, terms, counts = synthetic_lda_corpus(M, V, 5, len_mu=float(os.environ.get("MU", 6.9)), len_sigma=0.3, len_max=4000)
pc = tm.PackedCorpus(doc_ptr, terms, counts, V)
lens = np.diff(pc.doc_ptr)
print(f"M={M} nnz={pc.nnz} unique terms/doc mean {lens.mean():.0f} min {lens.min()} max {lens.max()}", flush=True)
gm = tm.gpuLDA(pc, K)
gm.beta = np.asfortranarray(tm.dirichlet_rows(K, V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
def it():
    gm.estep(10, 1.0 / K ** 2); gm.reduce_docs(); gm.update_beta(); gm.update_alpha(1000, 1.0 / K ** 2)
for _ in range(3): it()
gm.synchronize(); t0 = time.perf_counter()
for _ in range(10): it()
gm.synchronize(); sec = (time.perf_counter() - t0) / 10
print(f"{1e3 * sec:.2f} ms per iteration = {pc.nnz / sec / 1e9:.2f} G tokens/s (SYN-NSF: 10.9 M tokens in 1.0 ms = 10.9 G tokens/s); sweeps {gm.sweep_hist(11).tolist()}")
