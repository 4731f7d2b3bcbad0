"""
The decomposed update_elbo! of the CTPF path (round 5; ctpf_elbo_doc_parts_kernel in csrc/tmvb_ctpf.hip, src/CTPF.jl:111-247), as
tests/test_lda_elbo_parts_gpu.py / tests/test_ctm_elbo_parts_gpu.py: no entry (token / reader) is walked -- the checked iteration's document kernels
leave their softmax shifts, its statistics passes sum c log s per postings chunk, and the entries' remaining terms are sums the M-step already has.
Both forms against the fp64 oracle and against each other: every E-step kernel class (grid tile narrow / wide / four waves, LDS kernel, register tile),
ratings > 1, documents without readers, teacher-forced stepwise and through train!.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tol import within
from test_ctpf_gpu import force, load, make_pair, synth_case


@pytest.mark.parametrize("case", ["ctpf_m40_v60_u15_k4", "ctpf_m30_v40_u12_k6_r1", "syn_k12", "syn_k50", "syn_k64", "syn_k77", "syn_k100", "syn_k124"])
def test_both_forms_against_the_oracle_stepwise(tmvb, oracle, monkeypatch, case):
    g = load(case) if case.startswith("ctpf_") else synth_case(tmvb, int(case.split("_k")[1]), M=120, V=300, U=60)
    monkeypatch.setenv("TMVB_CTPF_ELBO_PARTS", "2")
    gp, om = make_pair(tmvb, oracle, g)
    monkeypatch.setenv("TMVB_CTPF_ELBO_PARTS", "0")
    gw, _ = make_pair(tmvb, oracle, g)
    for it in range(3):
        force(gp, om); force(gw, om)
        om.estep(viter=3, vtol=0.0); om.mstep(); e_o = om.update_elbo()
        vals = []
        for gm in (gp, gw):
            gm.estep(viter=3, vtol=0.0); gm.reduce_docs(); gm.mstep()
            vals.append(gm.update_elbo())
        assert gp.elbo_form() == 1 and gw.elbo_form() == 0
        within("ctpf.elbo_rel_step", abs(vals[0] - e_o) / abs(e_o), (case, it, "decomposed", vals[0], e_o))
        within("ctpf.elbo_rel_step", abs(vals[1] - e_o) / abs(e_o), (case, it, "table form", vals[1], e_o))
        within("ctpf.elbo_forms_rel", abs(vals[0] - vals[1]) / abs(vals[1]), (case, it, vals))


@pytest.mark.parametrize("grid", ["0", "1"])
def test_every_document_class(tmvb, oracle, monkeypatch, grid):
    """The corpus of test_long_documents_and_the_lane_per_token_path: all grid-tile classes, the LDS kernel, documents without readers; with
    TMVB_CTPF_GRID=0 the lane = token register tiles."""
    monkeypatch.setenv("TMVB_CTPF_GRID", grid)
    monkeypatch.setenv("TMVB_CTPF_ELBO_PARTS", "2")
    rng = np.random.default_rng(11)
    V, U, K = 1500, 700, 50
    shapes = [(30, 3), (64, 1), (90, 20), (128, 32), (129, 5), (192, 30), (100, 40), (128, 64), (60, 100), (250, 10), (200, 300), (20, 500),
              (300, 200), (380, 380), (600, 50), (10, 0), (5, 600)]
    tl, rl = [], []
    for n, r in shapes:
        tl.append((np.sort(rng.choice(V, size=n, replace=False)), rng.integers(1, 4, size=n)))
        rl.append((np.sort(rng.choice(U, size=r, replace=False)), rng.integers(1, 3, size=r)))
    g = dict(K=K, V=V, U=U, doc_ptr=np.concatenate([[0], np.cumsum([len(t) for t, _ in tl])]).astype(np.int64),
             terms=np.concatenate([t for t, _ in tl]).astype(np.int32), counts=np.concatenate([c for _, c in tl]).astype(np.int32),
             rdr_ptr=np.concatenate([[0], np.cumsum([len(t) for t, _ in rl])]).astype(np.int64),
             readers=np.concatenate([t for t, _ in rl]).astype(np.int32), ratings=np.concatenate([c for _, c in rl]).astype(np.int32),
             alef0=np.exp(tmvb.dirichlet_rows(K, V, seed=6) - 0.5))
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(2):
        force(gm, om)
        om.estep(viter=4, vtol=0.0); om.mstep(); e_o = om.update_elbo()
        gm.estep(viter=4, vtol=0.0); gm.reduce_docs(); gm.mstep(); e_g = gm.update_elbo()
        assert gm.elbo_form() == 1
        within("ctpf.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (grid, it, e_g, e_o))


def test_fallbacks(tmvb, oracle, monkeypatch):
    """The parts belong to ONE iteration with exactly one M-step behind its E-step; any other state takes the table form."""
    monkeypatch.setenv("TMVB_CTPF_ELBO_PARTS", "2")
    g = synth_case(tmvb, 20, M=60, V=200, U=40)
    gm, om = make_pair(tmvb, oracle, g)
    gm.estep(); gm.reduce_docs(); gm.mstep(); gm.update_elbo()
    assert gm.elbo_form() == 1
    gm.update_elbo()
    assert gm.elbo_form() == 1                               # evaluating twice changes nothing
    om.estep(viter=3, vtol=0.0); om.mstep()
    force(gm, om)                                            # the host sets the state
    e_g = gm.update_elbo(); e_o = om.update_elbo()
    assert gm.elbo_form() == 0
    gm.estep(); gm.reduce_docs()                             # no M-step behind the E-step
    gm.update_elbo()
    assert gm.elbo_form() == 0
    gm.mstep(); gm.mstep()                                   # two M-steps: alef_old is no longer the E-step's alef
    gm.update_elbo()
    assert gm.elbo_form() == 0
    gm.estep(viter=0, vtol=0.0); gm.reduce_docs(); gm.mstep(); gm.update_elbo()
    assert gm.elbo_form() == 0


def test_train_takes_the_decomposed_form_and_leaves_the_iteration_alone(tmvb, monkeypatch):
    pc = tmvb.syn_citeu(M=4000, V=3000, U=900, seed=9)
    K = 50
    out = []
    for env in ("1", "0"):
        monkeypatch.setenv("TMVB_CTPF_ELBO_PARTS", env)
        g = tmvb.gpuCTPF(pc, K)
        traj = np.asarray(g.train(iter=8, tol=0.0, checkelbo=1, printelbo=False), dtype=np.float64)
        out.append((g, traj))
    (gp, tp), (gw, tw) = out
    assert gp.elbo_form() == 1 and gw.elbo_form() == 0
    assert len(tp) == len(tw) == 8 and np.all(np.isfinite(tp))
    within("ctpf.elbo_forms_rel", np.abs(tp - tw) / np.abs(tw), (tp, tw))
    for n in ("alef", "he", "bet", "vav", "dalet", "het", "gimel", "zayin"):
        assert np.array_equal(getattr(gp, n), getattr(gw, n)), n
