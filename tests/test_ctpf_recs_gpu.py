"""
CTPF recommendation post-processing: tmvb_ctpf_recommend vs the oracle's fp64
restatement of src/CTPF.jl:379-399 (oracle.CTPF.recommend).
Tolerances: scores rel <= 5e-6 (fp32 state, f32 MFMA accumulation over K <= 64 products); rankings: the candidate
sets are identical and the ORACLE's scores read along the device ranking never increase by more than 1e-5 relative
(two candidates closer than the fp32 resolution may swap); exact ties must come in descending index order.
"""
import numpy as np
import pytest

from test_ctpf_gpu import force, make_pair, synth_case

pytestmark = pytest.mark.gpu


def check_ranking(got, want, oracle_scores):
    assert len(got) == len(want) and np.array_equal(np.sort(got), np.sort(want))
    s = oracle_scores[got]
    if len(s) > 1:
        assert np.all(s[1:] <= s[:-1] * (1 + 1e-5) + 1e-300)


@pytest.mark.parametrize("K,M,U", [(4, 40, 15), (12, 80, 60), (50, 150, 130)])
def test_scores_and_rankings_match_oracle(tmvb, oracle, K, M, U):
    g = synth_case(tmvb, K, M=M, V=200, U=U, seed=8)
    gm, om = make_pair(tmvb, oracle, g)
    om.train(iter=3, tol=0.0, checkelbo=float("inf"))
    force(gm, om)
    sc, drecs, urecs = om.recommend()
    gm.recommend()
    np.testing.assert_allclose(gm.scores, sc, rtol=5e-6)
    assert len(gm.drecs) == M and len(gm.urecs) == U
    for d in range(M):
        check_ranking(gm.drecs[d] - 1, drecs[d], sc[d, :])
    for u in range(U):
        check_ranking(gm.urecs[u] - 1, urecs[u], sc[:, u])
    # nothing a user has read is recommended to them, and vice versa
    for d in range(M):
        rd = g["readers"][g["rdr_ptr"][d]:g["rdr_ptr"][d + 1]]
        assert not np.intersect1d(gm.drecs[d] - 1, rd).size
        for u in rd:
            assert d not in (gm.urecs[u] - 1)


def test_exact_ties_in_descending_index_order(tmvb, oracle):
    g = synth_case(tmvb, 6, M=30, V=100, U=20, seed=5)
    gm, om = make_pair(tmvb, oracle, g)
    om.train(iter=2, tol=0.0, checkelbo=float("inf"))
    # documents 3, 7, 11 get identical expectations; users 2, 9 identical preferences: identical fp32 scores
    for d in (7, 11):
        om.gimel[:, d] = om.gimel[:, 3]; om.zayin[:, d] = om.zayin[:, 3]
    om.he[:, 9] = om.he[:, 2]
    force(gm, om)
    sc, drecs, urecs = om.recommend()
    gm.recommend()
    for u in range(gm.U):
        pos = {int(d): q for q, d in enumerate(gm.urecs[u] - 1)}
        tied = [d for d in (3, 7, 11) if d in pos]
        assert [d for d in sorted(tied, key=lambda d: pos[d])] == sorted(tied, reverse=True)      # 11, 7, 3
        opos = {int(d): q for q, d in enumerate(urecs[u])}
        assert [d for d in sorted(tied, key=lambda d: opos[d])] == sorted(tied, reverse=True)     # the oracle agrees
    for d in range(gm.M):
        pos = {int(u): q for q, u in enumerate(gm.drecs[d] - 1)}
        if 2 in pos and 9 in pos:
            assert pos[9] + 1 == pos[2]


def test_train_fills_recommendations(tmvb):
    g = synth_case(tmvb, 8, M=50, V=120, U=25, seed=3)
    pc = tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], g["V"], g["rdr_ptr"], g["readers"], g["ratings"], g["U"])
    m = tmvb.CTPF(pc, 8)
    assert m.scores is None and len(m.libs) == 25
    tmvb.gpu_train_ctpf(m, iter=3, tol=0.0, checkelbo=float("inf"), printelbo=False)
    assert m.scores.shape == (50, 25) and len(m.drecs) == 50 and len(m.urecs) == 25
    for u in range(25):
        assert not set(m.libs[u]) & set(int(x) for x in m.urecs[u])
        assert len(m.libs[u]) + len(m.urecs[u]) == 50
