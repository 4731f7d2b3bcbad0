"""
GPU parity tests for the LDA path: HIP engine (through the C ABI) vs the fp64 oracle and the
committed golden fixtures.  Stated fp64->fp32 tolerances:
  teacher-forced single step : gamma, Elogtheta rel <= 2e-4; beta rel <= 1e-4 on entries > 1e-6;
                               alpha rel <= 1e-4; ELBO rel <= 1e-6
  free running               : ELBO rel <= 1e-4 per iteration
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")

from tol import TOL, within          # every comparison by name: tests/tol.py holds the frozen tolerances

RTOL_ELBO_FREE = TOL["lda.elbo_rel_free"]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


def make_pair(tmvb, oracle, g):
    K, V = int(g["K"]), int(g["V"])
    pc = tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], V)
    gm = tmvb.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(g["beta0"]); gm.beta_old = gm.beta.copy(order="F")
    gm.update_buffer()
    om = oracle.LDA(oracle.CSR(g["doc_ptr"], g["terms"], g["counts"], V), K, g["beta0"])
    return gm, om


def rel(a, b, floor=0.0):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor if floor else 1e-300)


def force(gm, om):
    """teacher forcing: reset the device state to the oracle's state"""
    gm.alpha = om.alpha.copy(); gm.beta = om.beta.copy(order="F"); gm.beta_old = om.beta_old.copy(order="F")
    gm.gamma = om.gamma.copy(order="F"); gm.Elogtheta = om.Elogtheta.copy(order="F")
    gm.Elogtheta_old = om.Elogtheta_old.copy(order="F")
    gm.update_buffer()


@pytest.mark.parametrize("name", ["lda_m40_v60_k3", "lda_m40_v60_k7", "lda_m30_v50_k70_empty"])
def test_teacher_forced_fixed_sweeps(tmvb, oracle, name):
    """vtol=0 pins every document to exactly `viter` sweeps on both sides: pure arithmetic parity."""
    g = load(name)
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(4):
        force(gm, om)
        gm.estep(viter=3, vtol=0.0); gm.reduce_docs()
        om.estep(viter=3, vtol=0.0)
        gm.update_beta(); om.update_beta()
        gm.update_alpha(); om.update_alpha()
        e_g = gm.update_elbo(); e_o = om.update_elbo()
        gm.update_host()
        within("lda.gamma_rel", rel(gm.gamma, om.gamma), it)
        within("lda.Elogtheta_rel", rel(gm.Elogtheta, om.Elogtheta), it)
        within("lda.Elogtheta_rel", rel(gm.Elogtheta_old, om.Elogtheta_old), (it, "old"))
        big = om.beta > 1e-6
        within("lda.beta_rel", rel(gm.beta[big], om.beta[big]), it)
        within("lda.beta_abs", np.abs(gm.beta - om.beta), it)
        assert rel(gm.beta_old, om.beta_old, 1e-6).max() <= 1e-6, (it, "beta_old")      # the forced state itself (an fp32 round trip)
        within("lda.alpha_rel", rel(gm.alpha, om.alpha), it)
        within("lda.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (it, e_g, e_o))
        # check_model post-conditions (src/modelutils.jl:255-279)
        np.testing.assert_allclose(gm.beta.sum(axis=1), 1.0, rtol=1e-5)
        assert np.all(gm.gamma > 0) and np.all(gm.Elogtheta <= 0)


@pytest.mark.parametrize("name", ["lda_m40_v60_k3", "lda_m40_v60_k7"])
def test_teacher_forced_default_exit_rule(tmvb, oracle, name):
    """Default vtol: per-document early exit (src/LDA.jl:175).  Documents whose sweep count matches the
    oracle's must match tightly; mismatches (||delta|| straddling vtol in fp32) are counted."""
    g = load(name)
    gm, om = make_pair(tmvb, oracle, g)
    mism = tot = 0
    for it in range(4):
        force(gm, om)
        gm.estep(); gm.reduce_docs()
        sw_o = om.estep()
        gm.update_beta(); om.update_beta()
        gm.update_alpha(); om.update_alpha()
        gm.update_host()
        sw_g = gm.doc_sweeps()
        same = sw_g == np.asarray(sw_o)
        assert np.array_equal(gm.sweep_hist(), np.bincount(sw_g, minlength=11))
        mism += int((~same).sum()); tot += len(sw_o)
        # per-document state given the (forced) globals depends on the document's own sweeps only: every document whose
        # exit sweep agrees with the oracle's is compared, whatever the others did
        assert same.any()
        within("lda.gamma_rel", rel(gm.gamma[:, same], om.gamma[:, same]), it)
        within("lda.Elogtheta_rel", rel(gm.Elogtheta[:, same], om.Elogtheta[:, same]), it)
        if same.all():                                       # the globals see every document
            within("lda.alpha_rel", rel(gm.alpha, om.alpha), it)
            big = om.beta > 1e-6
            within("lda.beta_rel", rel(gm.beta[big], om.beta[big]), it)
    assert mism <= 0.05 * tot, f"{mism}/{tot} documents changed sweep count"


@pytest.mark.parametrize("name", ["lda_m40_v60_k3", "lda_m40_v60_k7", "lda_m30_v50_k70_empty"])
def test_free_running_train_vs_golden(tmvb, name):
    g = load(name)
    K, V = int(g["K"]), int(g["V"])
    gm = tmvb.gpuLDA(tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], V), K)
    gm.beta = np.asfortranarray(g["beta0"]); gm.beta_old = gm.beta.copy(order="F")
    traj = gm.train(iter=int(g["iters"]), tol=0.0, checkelbo=1, printelbo=False)
    gold = g["elbo_traj"]
    assert len(traj) == len(gold)            # ELBO increases every iteration on these fixtures
    within("lda.elbo_rel_free", np.abs(traj - gold) / np.abs(gold), (traj, gold))
    within("lda.alpha_rel_free", rel(gm.alpha, g["alpha"]))
    within("lda.beta_abs_free", np.abs(gm.beta - g["beta"]))
    # topics = descending sortperm of beta rows (src/gpuLDA.jl:374), 1-based
    assert sorted(gm.topics[0].tolist()) == list(range(1, V + 1))


def test_free_running_medium_corpus_tracks_the_oracle(tmvb, oracle):
    """25 free-running iterations on a 1 500-document corpus (K = 20): the device trajectory must not drift from the
    fp64 oracle's -- ELBO rel <= 1e-4 at every iteration, same stop decision under the signed rule (Q4)."""
    pc = tmvb.syn_nsf(M=1500, V=2000, seed=23)
    K = 20
    beta0 = tmvb.dirichlet_rows(K, pc.V, seed=4)
    gm = tmvb.gpuLDA(pc, K)
    gm.beta = np.asfortranarray(beta0); gm.beta_old = gm.beta.copy(order="F")
    om = oracle.LDA(oracle.CSR(pc.doc_ptr, pc.terms, pc.counts, pc.V), K, beta0)
    t_g = gm.train(iter=25, tol=1.0, checkelbo=1, printelbo=False)
    t_o = om.train(iter=25, tol=1.0, checkelbo=1)
    assert abs(len(t_g) - len(t_o)) <= 1                       # stop iteration +-1
    n = min(len(t_g), len(t_o))
    within("lda.elbo_rel_free", np.abs(t_g[:n] - t_o[:n]) / np.abs(t_o[:n]), (t_g, t_o))


@pytest.mark.parametrize("pieces", [None, 2])
@pytest.mark.parametrize("K", [7, 50])
def test_long_documents_stream_through_the_tile(tmvb, oracle, K, pieces, monkeypatch):
    """Long documents: 300 / 600 / 900 unique terms run on the 4-wave register kernel (2, 3, 4 tiles per wave at K = 50),
    1100+ on the LDS-tile kernel -- resident in a 156 KiB tile at K = 7, streamed in chunks (re-gathered per sweep) at
    K = 50, where the tile holds ~700 rows."""
    if pieces:                                         # pipelined statistics passes: the long documents belong to the last piece
        monkeypatch.setenv("TMVB_LDA_PIECES", str(pieces))
    rng = np.random.default_rng(5)
    V = 2600
    docs = []
    for n in (2000, 1500, 1100, 40, 3, 0, 900, 600, 300, 70, 130, 20, 250):
        t = np.sort(rng.choice(V, size=n, replace=False)); c = rng.integers(1, 4, size=n)
        docs.append((t, c))
    doc_ptr = np.concatenate([[0], np.cumsum([len(t) for t, _ in docs])])
    terms = np.concatenate([t for t, _ in docs]).astype(np.int32); counts = np.concatenate([c for _, c in docs]).astype(np.int32)
    beta0 = tmvb.dirichlet_rows(K, V, seed=3)
    g = dict(K=K, V=V, doc_ptr=doc_ptr, terms=terms, counts=counts, beta0=beta0)
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(3):
        force(gm, om)
        gm.estep(viter=4, vtol=0.0); gm.reduce_docs(); om.estep(viter=4, vtol=0.0)
        gm.update_beta(); om.update_beta()
        gm.update_alpha(); om.update_alpha()
        e_g = gm.update_elbo(); e_o = om.update_elbo()
        gm.update_host()
        within("lda.gamma_rel", rel(gm.gamma, om.gamma), it)
        big = om.beta > 1e-6
        within("lda.beta_rel", rel(gm.beta[big], om.beta[big]), it)
        within("lda.elbo_rel_step", abs(e_g - e_o) / abs(e_o), it)


def test_epsilon_keeps_phi_defined_where_a_beta_column_is_zero(tmvb, oracle):
    """update_phi! adds epsilon before normalising (src/LDA.jl:152): for a term whose beta column is all zero -- a vocabulary entry no training document
    used, met by predict on new text -- phi = eps / (K eps) = 1 / K exactly, where 0 / 0 would poison gamma.  Known answer: a document made of that
    term only gets gamma_i = alpha_i + C_d / K + eps; and the whole step against the oracle.  (The negative control of tests/test_mutants_gpu.py: the
    library built with epsilon dropped must FAIL here.)"""
    K, V = 5, 7
    rng = np.random.default_rng(3)
    beta0 = rng.random((K, V)); beta0[:, 3] = 0.0; beta0 /= beta0.sum(axis=1, keepdims=True)
    docs = [([3], [4]), ([0, 3, 5], [2, 1, 3]), ([1, 2, 6], [1, 1, 2])]
    doc_ptr = np.concatenate([[0], np.cumsum([len(t) for t, _ in docs])]).astype(np.int64)
    terms = np.concatenate([t for t, _ in docs]).astype(np.int32); counts = np.concatenate([c for _, c in docs]).astype(np.int32)
    g = dict(K=K, V=V, doc_ptr=doc_ptr, terms=terms, counts=counts, beta0=beta0)
    gm, om = make_pair(tmvb, oracle, g)
    gm.estep(viter=3, vtol=0.0); gm.reduce_docs(); om.estep(viter=3, vtol=0.0)
    gm.update_host()
    assert np.all(np.isfinite(gm.gamma)) and np.all(np.isfinite(gm.Elogtheta))
    np.testing.assert_allclose(gm.gamma[:, 0], 1.0 + 4.0 / K, rtol=1e-6)          # alpha = 1 (constructor), phi uniform
    within("lda.gamma_rel", rel(gm.gamma, om.gamma))
    gm.update_beta(); om.update_beta(); gm.update_host()
    assert np.all(np.isfinite(gm.beta))
    within("lda.beta_abs", np.abs(gm.beta - om.beta))


def test_k1_closed_form_on_device(tmvb):
    """K=1: phi == 1, gamma_d = alpha + C_d + eps, beta = empirical unigram distribution."""
    rng = np.random.default_rng(0)
    V, M = 9, 12
    docs = []
    for _ in range(M):
        t = np.sort(rng.choice(V, size=rng.integers(1, 6), replace=False)); c = rng.integers(1, 5, size=len(t))
        docs.append((t, c))
    doc_ptr = np.concatenate([[0], np.cumsum([len(t) for t, _ in docs])])
    pc = tmvb.PackedCorpus(doc_ptr, np.concatenate([t for t, _ in docs]), np.concatenate([c for _, c in docs]), V)
    gm = tmvb.gpuLDA(pc, 1)
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_host()
    np.testing.assert_allclose(gm.gamma[0], 1.0 + pc.C, rtol=1e-6)
    emp = np.bincount(pc.terms, weights=pc.counts, minlength=V)
    np.testing.assert_allclose(gm.beta[0], emp / emp.sum(), rtol=1e-6)


def test_argument_and_corpus_errors(tmvb):
    pc = tmvb.PackedCorpus([0, 2], [0, 1], [1, 1], 3)
    with pytest.raises(ValueError):
        tmvb.gpuLDA(pc, 0)                                   # src/gpuLDA.jl:47
    gm = tmvb.gpuLDA(pc, 2)
    with pytest.raises(ValueError):
        gm.train(iter=-1, printelbo=False)                   # src/gpuLDA.jl:350
    with pytest.raises(ValueError):
        gm.train(tol=-1.0, printelbo=False)                  # src/gpuLDA.jl:349
    with pytest.raises(ValueError):
        gm.train(checkelbo=0, printelbo=False)               # src/gpuLDA.jl:351
    bad = tmvb.PackedCorpus([0, 2], [0, 7], [1, 1], 3)       # term id outside the vocabulary
    with pytest.raises(tmvb.CorpusError):
        tmvb.gpuLDA(bad, 2)
    bad2 = tmvb.PackedCorpus([0, 2], [0, 1], [1, 0], 3)      # non-positive count (check_doc)
    with pytest.raises(tmvb.CorpusError):
        tmvb.gpuLDA(bad2, 2)


def test_gpu_macro_round_trip(tmvb, oracle):
    """`@gpu train!(model)` (src/macros.jl:113-150): host LDA -> device -> host, beta renormalised in fp64."""
    g = load("lda_m40_v60_k7")
    K, V = int(g["K"]), int(g["V"])
    m = tmvb.LDA(tmvb.PackedCorpus(g["doc_ptr"], g["terms"], g["counts"], V), K)
    m.beta = np.asfortranarray(g["beta0"]); m.beta_old = m.beta.copy(order="F")
    traj = tmvb.gpu_train(m, iter=int(g["iters"]), tol=0.0, printelbo=False)
    tmvb.check_model(m)                                     # Float64 stochasticity tolerance holds after :147
    within("lda.elbo_rel_free", np.abs(traj - g["elbo_traj"]) / np.abs(g["elbo_traj"]))
    assert np.array_equal(m.Elogtheta, m.Elogtheta_old) and np.array_equal(m.beta, m.beta_old)


def test_nsf_scale_invariants(tmvb):
    """Size-independent properties at a few thousand NSF-shaped documents (full-size run is bench.py):
    sum_i(gamma_id - alpha_i) = C_d, beta rows stochastic, total statistics mass = total token count."""
    pc = tmvb.syn_nsf(M=6000, V=25319, seed=1)
    K = 50
    gm = tmvb.gpuLDA(pc, K)
    alpha0 = gm.alpha.copy()
    gm.estep(); gm.reduce_docs()
    import ctypes as C
    ptr, n = gm.stats()
    gm.update_host()
    mass = (gm.gamma - alpha0[:, None]).sum(axis=0)
    np.testing.assert_allclose(mass, pc.C, rtol=2e-5)
    gm.update_beta(); gm.update_alpha(); gm.update_host()
    np.testing.assert_allclose(gm.beta.sum(axis=1), 1.0, rtol=1e-5)
    assert np.all(gm.alpha > 0) and np.all(np.isfinite(gm.alpha))
    e1 = gm.update_elbo()
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()
    e2 = gm.update_elbo()
    assert np.isfinite(e1) and np.isfinite(e2) and e2 > e1


@pytest.mark.parametrize("K", [50, 100])
def test_full_size_nsf_properties(tmvb, K):
    """BASELINE.json's full size (SYN-NSF M=128804, V=25319; K=50 = config 2, K=100 = config 3's model on one GPU) through
    size-independent properties: per-document mass conservation, stochastic beta, statistics mass = token count,
    increasing ELBO, run-to-run bitwise reproducibility of the (atomics-free) statistics."""
    pc = tmvb.syn_nsf()
    gm = tmvb.gpuLDA(pc, K)
    alpha0 = gm.alpha.copy()
    gm.estep(); gm.reduce_docs(); gm.update_host()
    np.testing.assert_allclose((gm.gamma - alpha0[:, None]).sum(axis=0), pc.C, rtol=3e-5)
    assert np.all(gm.gamma > 0) and np.all(gm.Elogtheta <= 0)
    gm.update_beta(); gm.update_alpha(); gm.update_host()
    np.testing.assert_allclose(gm.beta.sum(axis=1), 1.0, rtol=1e-5)
    beta1 = gm.beta.copy()
    e1 = gm.update_elbo()
    # same state again in a second model: identical bits (deterministic reductions, no atomics)
    gm2 = tmvb.gpuLDA(pc, K)
    gm2.estep(); gm2.reduce_docs(); gm2.update_beta(); gm2.update_alpha(); gm2.update_host()
    assert np.array_equal(gm2.beta, beta1) and np.array_equal(gm2.alpha, gm.alpha)
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_alpha()
    e2 = gm.update_elbo()
    assert np.isfinite(e1) and e2 > e1
    assert gm.sweep_hist().sum() == pc.M


@pytest.mark.parametrize("K", [3, 10, 18, 25, 33, 41, 50, 57, 64, 72, 81, 90, 100, 130, 300])
@pytest.mark.parametrize("merged", [True, False])
def test_every_kernel_path_by_k(tmvb, oracle, K, merged, monkeypatch):
    """K = 130 (LDS-tile kernel, float4 statistics kernel with stored weights) and K = 300 (scalar statistics kernel),
    and one K per register-tile instantiation (KP = 4, 12, 20, ..., 100; one result slot per lane up to KP = 60,
    two from KP = 68; statistics recomputed with 16- or 32-lane row slots) incl. multi-tile documents,
    teacher-forced against the oracle with pinned sweep counts."""
    if not merged:                                     # one launch per tile count (what large corpora use) instead of
        monkeypatch.setenv("TMVB_LDA_NO_MERGE", "1")   # the single mixed-tile launch of small corpora
    pc = tmvb.syn_nsf(M=120, V=900, seed=17)           # document lengths ~30..250 -> 1..4 tiles
    g = dict(K=K, V=pc.V, doc_ptr=pc.doc_ptr, terms=pc.terms, counts=pc.counts, beta0=tmvb.dirichlet_rows(K, pc.V, seed=3))
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(3):
        force(gm, om)
        gm.estep(viter=4, vtol=0.0); gm.reduce_docs(); om.estep(viter=4, vtol=0.0)
        gm.update_beta(); om.update_beta()
        gm.update_alpha(); om.update_alpha()
        e_g = gm.update_elbo(); e_o = om.update_elbo()
        gm.update_host()
        within("lda.gamma_rel", rel(gm.gamma, om.gamma), (it, "gamma"))
        big = om.beta > 1e-6
        within("lda.beta_rel", rel(gm.beta[big], om.beta[big]), (it, "beta"))
        within("lda.alpha_rel", rel(gm.alpha, om.alpha))
        within("lda.elbo_rel_step", abs(e_g - e_o) / abs(e_o))


@pytest.mark.parametrize("K", [10, 50, 100])
def test_lane_per_token_register_tile_path_still_matches_the_oracle(tmvb, oracle, K, monkeypatch):
    """Round 3 made the grid-tile kernel (tmvb_gridtile.h) the default for KP <= 100; the lane = token register-tile kernels of
    rounds 1-2 stay in the library behind TMVB_LDA_GRID=0 (and, multi-wave, for documents beyond the grid kernel's reach): they
    must keep matching the oracle."""
    monkeypatch.setenv("TMVB_LDA_GRID", "0")
    pc = tmvb.syn_nsf(M=120, V=900, seed=17)
    g = dict(K=K, V=pc.V, doc_ptr=pc.doc_ptr, terms=pc.terms, counts=pc.counts, beta0=tmvb.dirichlet_rows(K, pc.V, seed=3))
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(2):
        force(gm, om)
        gm.estep(viter=4, vtol=0.0); gm.reduce_docs(); om.estep(viter=4, vtol=0.0)
        gm.update_beta(); om.update_beta(); gm.update_alpha(); om.update_alpha()
        e_g = gm.update_elbo(); e_o = om.update_elbo()
        gm.update_host()
        within("lda.gamma_rel", rel(gm.gamma, om.gamma), (it, "gamma"))
        big = om.beta > 1e-6
        within("lda.beta_rel", rel(gm.beta[big], om.beta[big]), (it, "beta"))
        within("lda.alpha_rel", rel(gm.alpha, om.alpha))
        within("lda.elbo_rel_step", abs(e_g - e_o) / abs(e_o))


def test_grid_tile_long_documents_two_and_four_waves(tmvb, oracle):
    """Documents of 193 .. 768 unique terms run the grid-tile kernel with two / four waves per document (partial sums through
    LDS, identical tails in every wave); beyond that the four-wave lane = token kernel and the LDS kernel.  One corpus with all
    of them, K = 50 (six pairs per lane) and K = 100 (three), teacher-forced against the oracle with pinned sweeps."""
    rng = np.random.default_rng(3)
    V = 3000
    lens = [40, 150, 200, 300, 380, 390, 500, 760, 800, 1100, 1500, 90, 64, 65, 96, 97, 128, 129, 192, 193]
    docs = [(np.sort(rng.choice(V, size=n, replace=False)), rng.integers(1, 5, size=n)) for n in lens]
    doc_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    terms = np.concatenate([t for t, _ in docs]).astype(np.int32); counts = np.concatenate([c for _, c in docs]).astype(np.int32)
    for K in (50, 100):
        g = dict(K=K, V=V, doc_ptr=doc_ptr, terms=terms, counts=counts, beta0=tmvb.dirichlet_rows(K, V, seed=3))
        gm, om = make_pair(tmvb, oracle, g)
        for it in range(2):
            force(gm, om)
            gm.estep(viter=5, vtol=0.0); gm.reduce_docs(); om.estep(viter=5, vtol=0.0)
            gm.update_beta(); om.update_beta(); gm.update_alpha(); om.update_alpha()
            gm.update_host()
            assert np.all(gm.doc_sweeps() == 5)
            within("lda.gamma_rel", rel(gm.gamma, om.gamma), (K, it, "gamma"))
            within("lda.Elogtheta_rel", rel(gm.Elogtheta, om.Elogtheta), (K, it, "Elogtheta"))
            big = om.beta > 1e-6
            within("lda.beta_rel", rel(gm.beta[big], om.beta[big]), (K, it, "beta"))


def test_train_equals_stepwise_pipelined(tmvb):
    """train! enqueues its iterations without host synchronisation; the stepwise loop below synchronises after every
    operator.  Both must give the same state bit for bit: every stream that carries document kernels has to wait for the
    previous iteration's M-step (a missing wait on the chain stream of the pipelined plan once let the next E-step start
    under the running M-step -- invisible to tests that synchronise between steps)."""
    pc = tmvb.syn_nsf(M=30000, V=8000, seed=17)
    K = 50

    def fresh():
        g = tmvb.gpuLDA(pc, K)
        g.beta = np.asfortranarray(tmvb.dirichlet_rows(K, pc.V, seed=3)); g.beta_old = g.beta.copy(order="F"); g.update_buffer()
        return g
    a = fresh()
    a.train(iter=8, tol=0.0, checkelbo=np.inf, printelbo=False)
    b = fresh()
    for it in range(8):
        b.estep(10, 1.0 / K ** 2); b.synchronize()
        b.reduce_docs(); b.synchronize()
        b.update_beta(); b.synchronize()
        b.update_alpha(1000, 1.0 / K ** 2); b.synchronize()
    b.update_host()
    assert np.array_equal(a.beta, b.beta) and np.array_equal(a.alpha, b.alpha)
    assert np.array_equal(a.gamma, b.gamma) and np.array_equal(a.Elogtheta, b.Elogtheta)


def test_two_copy_kernel_for_short_documents(tmvb, oracle, monkeypatch):
    """lda_estep_tt_kernel (TMVB_LDA_TT=1; off by default: measured 2 % slower than the NP = 2 grid tile): documents of
    <= 64 unique terms with the tile held twice (lane = token for s = B e, lane = topic for g = B' w).  Same parity bar as every other path."""
    monkeypatch.setenv("TMVB_LDA_TT", "1")
    monkeypatch.setenv("TMVB_LDA_NO_MERGE", "1")          # one launch per class, as large corpora run
    pc = tmvb.syn_nsf(M=200, V=900, seed=21)
    K = 50
    assert (np.diff(pc.doc_ptr) <= 64).sum() >= 30 and (np.diff(pc.doc_ptr) > 64).sum() >= 30
    g = dict(K=K, V=pc.V, doc_ptr=pc.doc_ptr, terms=pc.terms, counts=pc.counts, beta0=tmvb.dirichlet_rows(K, pc.V, seed=3))
    gm, om = make_pair(tmvb, oracle, g)
    for it in range(3):
        force(gm, om)
        gm.estep(viter=5, vtol=0.0); gm.reduce_docs(); om.estep(viter=5, vtol=0.0)
        gm.update_beta(); om.update_beta()
        gm.update_alpha(); om.update_alpha()
        e_g = gm.update_elbo(); e_o = om.update_elbo()
        gm.update_host()
        within("lda.gamma_rel", rel(gm.gamma, om.gamma), (it, "gamma"))
        within("lda.Elogtheta_rel", rel(gm.Elogtheta, om.Elogtheta), (it, "Elogtheta"))
        big = om.beta > 1e-6
        within("lda.beta_rel", rel(gm.beta[big], om.beta[big]), (it, "beta"))
        within("lda.elbo_rel_step", abs(e_g - e_o) / abs(e_o), (it, e_g, e_o))

