"""
Host-side timing of the library's asynchronous calls: tmvb_event_* (HIP timing events without the system-scope fence of a default
event; what bench.py brackets the E-step with) and the per-model E-step timing that the library records only on request
(TMVB_ESTEP_TIMING=1, read when the model is created -- LDA and, since round 4, CTPF: two default events per iteration on the stream the
whole iteration runs on cost a 0.14 ms CTPF iteration 6 %).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_timing_events_bracket_an_estep(tmvb):
    corpus = tmvb.syn_nsf(M=3000, V=2000, seed=5)
    gm = tmvb.gpuLDA(corpus, 50)
    a, b, c = gm.ctx.timing_event(), gm.ctx.timing_event(), gm.ctx.timing_event()
    a.record()
    gm.estep()
    b.record()
    gm.estep(); gm.estep()
    c.record()
    one, three = a.elapsed_ms(b), a.elapsed_ms(c)
    assert 0.0 < one < three < 1e4
    gm.synchronize()
    for e in (a, b, c):
        e.close()
    gm.close()


_SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
import tmvb_amd
tm = tmvb_amd.pkg
out = {}
for name, make in (("lda", lambda: tm.gpuLDA(tm.syn_nsf(M=500, V=400, seed=1), 50)), ("ctpf", lambda: tm.gpuCTPF(tm.syn_citeu(M=400, V=300, U=60, seed=2), 50))):
    g = make(); g.estep()
    try:
        out[name] = float(g.last_estep_ms())
    except ValueError as e:
        out[name] = str(e)
    g.close()
print(json.dumps(out))
""" % ROOT


@pytest.mark.parametrize("flag", ["0", "1"])
def test_estep_timing_is_recorded_on_request_only(flag):
    env = dict(os.environ, TMVB_ESTEP_TIMING=flag)
    r = subprocess.run([sys.executable, "-c", _SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for name in ("lda", "ctpf"):
        if flag == "1":
            assert isinstance(out[name], float) and 0.0 < out[name] < 1e3, out
        else:
            assert isinstance(out[name], str) and "TMVB_ESTEP_TIMING" in out[name], out
