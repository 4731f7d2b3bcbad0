"""
The inverted index built by several host threads (tmvb_build_inv_index: per-range histograms + one prefix + per-range scatter; the pieces of an LDA model on a
thread each) holds the postings in exactly the positions the one-thread counting sort gives them: the statistics pass sums in index order, so a model trained on
either build must come out bit-identical.  TMVB_CREATE_THREADS is read once per process: two subprocesses.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
pc = tm.syn_nsf(M=20000, V=6000)                     # ~1.7 M postings: above the threading threshold of 2^20
assert pc.nnz >= (1 << 20), pc.nnz
out = []
g = tm.gpuLDA(pc, 20)
traj = g.train(iter=4, checkelbo=1, tol=0.0, printelbo=False)
out.append(("lda", [float(x).hex() for x in traj], hashlib.sha256(np.ascontiguousarray(g.beta).tobytes()).hexdigest()[:16]))
g.close()
pc2 = tm.syn_citeu(M=16980, V=8000, U=5551)          # term index 1.07 M postings, reader index below the threshold
g = tm.gpuCTPF(pc2, 10)
traj = g.train(iter=3, checkelbo=1, tol=0.0, printelbo=False, recs=False)
out.append(("ctpf", [float(x).hex() for x in traj], hashlib.sha256(np.ascontiguousarray(g.alef).tobytes()).hexdigest()[:16]))
g.close()
print("RESULT", out)
"""


def run(threads):
    env = dict(os.environ)
    if threads is None:
        env.pop("TMVB_CREATE_THREADS", None)
    else:
        env["TMVB_CREATE_THREADS"] = str(threads)
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert lines, r.stdout[-2000:]
    return lines[-1]


def test_threaded_index_build_is_bit_identical_to_the_one_thread_build():
    one = run(0)
    assert run(None) == one
    assert run(3) == one
