"""
The library's ordering events carry no system-scope fence by default (csrc/tmvb_core.hip: tmvb_event_flags, TMVB_EVENT_FLAGS=1): they
order streams of one device.  A fence that mattered would show as a race, i.e. as results that differ from the fenced build's: the
three stream plans that cross streams most -- pipelined LDA (three pieces, shadow statistics passes, the side chain), CTPF (narrow /
wide / four-wave launches on three streams) and CTM (staged update_sigma!, regrouping on side streams) -- must be bit-identical under
TMVB_EVENT_FLAGS=0, 1 and 2 (the flag is read once per process: one subprocess each).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import hashlib, json, os, sys
sys.path.insert(0, %r)
import numpy as np
import tmvb_amd
tm = tmvb_amd.pkg
out = {}
os.environ["TMVB_LDA_PIECES"] = "3"
c = tm.syn_nsf(M=6000, V=3000, seed=41)
g = tm.gpuLDA(c, 50); g.beta = np.asfortranarray(tm.dirichlet_rows(50, c.V, seed=3)); g.beta_old = g.beta.copy(order="F"); g.update_buffer()
t = g.train(iter=12, tol=0.0, checkelbo=1, printelbo=False)
out["lda"] = hashlib.sha256(np.ascontiguousarray(g.beta).tobytes() + np.ascontiguousarray(g.gamma).tobytes() + np.asarray(t).tobytes()).hexdigest()
g.close()
c = tm.syn_nsf(M=3000, V=1500, seed=42)
g = tm.gpuCTM(c, 50); t = g.train(iter=6, tol=0.0, checkelbo=1, printelbo=False)
out["ctm"] = hashlib.sha256(np.ascontiguousarray(g.beta).tobytes() + np.ascontiguousarray(g.lam).tobytes() + np.ascontiguousarray(g.sigma).tobytes() + np.asarray(t).tobytes()).hexdigest()
g.close()
c = tm.syn_citeu(M=3000, V=2000, U=600, seed=43)
g = tm.gpuCTPF(c, 50); t = g.train(iter=12, tol=0.0, checkelbo=1, printelbo=False, recs=False)
out["ctpf"] = hashlib.sha256(np.ascontiguousarray(g.alef).tobytes() + np.ascontiguousarray(g.gimel).tobytes() + np.asarray(t).tobytes()).hexdigest()
g.close()
print(json.dumps(out))
""" % ROOT


def _run(flags):
    env = dict(os.environ, TMVB_EVENT_FLAGS=str(flags))
    r = subprocess.run([sys.executable, "-c", _SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_results_do_not_depend_on_the_event_flags():
    a, b, c = _run(0), _run(1), _run(2)
    assert a == b == c, (a, b, c)
