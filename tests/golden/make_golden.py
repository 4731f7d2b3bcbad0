"""
Generates the committed golden fixtures tests/golden/*.npz.

The reference (Julia) cannot be run or imported in the build image and ships no golden vectors
(PARITY UNPINNED, see oracle/tmvb_oracle.h), so these fixtures are produced by the independent
NumPy/SciPy restatement oracle/oracle_np.py.  They pin the C oracle (and through it the HIP path)
against a second implementation and against accidental drift.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_np as onp  # noqa: E402


def mkcorp(rng, M, V, U=0, maxN=12, maxC=4, maxR=4, maxRat=1, empty_every=0):
    docs = []
    for d in range(M):
        n = int(rng.integers(1, maxN + 1))
        if empty_every and d % empty_every == empty_every - 1:
            n = 0
        t = np.sort(rng.choice(V, size=min(n, V), replace=False))
        c = rng.integers(1, maxC + 1, size=len(t))
        if U:
            r = int(rng.integers(0, maxR + 1))
            rd = np.sort(rng.choice(U, size=r, replace=False))
            rt = rng.integers(1, maxRat + 1, size=r)
            docs.append((t.tolist(), c.tolist(), rd.tolist(), rt.tolist()))
        else:
            docs.append((t.tolist(), c.tolist()))
    return docs


def pack(docs):
    doc_ptr = [0]; terms = []; counts = []; rdr_ptr = [0]; readers = []; ratings = []
    for doc in docs:
        terms += doc[0]; counts += doc[1]; doc_ptr.append(len(terms))
        if len(doc) > 2:
            readers += doc[2]; ratings += doc[3]
        rdr_ptr.append(len(readers))
    return dict(doc_ptr=np.array(doc_ptr, np.int64), terms=np.array(terms, np.int32), counts=np.array(counts, np.int32),
                rdr_ptr=np.array(rdr_ptr, np.int64), readers=np.array(readers, np.int32), ratings=np.array(ratings, np.int32))


def beta_init(rng, K, V):
    b = rng.exponential(size=(K, V))
    return b / b.sum(axis=1, keepdims=True)


def lda_case(name, seed, M, V, K, iters, empty_every=0):
    rng = np.random.default_rng(seed)
    docs = mkcorp(rng, M, V, empty_every=empty_every)
    beta0 = beta_init(rng, K, V)
    m = onp.LDA(docs, V, K, beta0)
    traj = m.train(iter=iters, tol=-1e300)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, V=V, K=K, iters=iters, beta0=beta0, **pack(docs),
                        alpha=m.alpha, beta=m.beta, beta_old=m.beta_old, gamma=np.stack(m.gamma, 1),
                        Elogtheta=np.stack(m.Elogtheta, 1), Elogtheta_old=np.stack(m.Elogtheta_old, 1),
                        elbo_traj=np.array(traj), sweeps=np.array(m.sweeps, np.int32))


def ctm_case(name, seed, M, V, K, iters):
    rng = np.random.default_rng(seed)
    docs = mkcorp(rng, M, V)
    beta0 = beta_init(rng, K, V)
    m = onp.CTM(docs, V, K, beta0)
    traj = m.train(iter=iters, tol=-1e300)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, V=V, K=K, iters=iters, beta0=beta0, **pack(docs),
                        mu=m.mu, sigma=m.sigma, invsigma=m.invsigma, beta=m.beta, lam=np.stack(m.lam, 1),
                        vsq=np.stack(m.vsq, 1), logzeta=m.logzeta, elbo_traj=np.array(traj))


def ctpf_case(name, seed, M, V, U, K, iters, maxRat):
    rng = np.random.default_rng(seed)
    docs = mkcorp(rng, M, V, U=U, maxRat=maxRat)
    alef0 = np.exp(beta_init(rng, K, V) - 0.5)
    m = onp.CTPF(docs, V, U, K, alef0)
    traj = m.train(iter=iters, tol=-1e300)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, V=V, U=U, K=K, iters=iters, alef0=alef0, **pack(docs),
                        alef=m.alef, he=m.he, bet=m.bet, vav=m.vav, dalet=m.dalet, het=m.het,
                        gimel=np.stack(m.gimel, 1), zayin=np.stack(m.zayin, 1), elbo_traj=np.array(traj))


def flda_case(name, seed, M, V, K, iters, empty_every=0):
    rng = np.random.default_rng(seed)
    docs = mkcorp(rng, M, V, empty_every=empty_every)
    beta0 = beta_init(rng, K, V)
    kappa0 = beta_init(rng, 1, V)[0]
    m = onp.fLDA(docs, V, K, beta0, kappa0)
    traj = m.train(iter=iters, tol=-1e300)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, V=V, K=K, iters=iters, beta0=beta0, kappa0=kappa0, **pack(docs),
                        eta=m.eta, alpha=m.alpha, kappa=m.kappa, beta=m.beta, beta_old=m.beta_old, gamma=np.stack(m.gamma, 1),
                        Elogtheta=np.stack(m.Elogtheta, 1), Elogtheta_old=np.stack(m.Elogtheta_old, 1),
                        tau=np.concatenate(m.tau) if m.tau else np.zeros(0), tau_old=np.concatenate(m.tau_old) if m.tau_old else np.zeros(0),
                        elbo_traj=np.array(traj), sweeps=np.array(m.sweeps, np.int32))


def fctm_case(name, seed, M, V, K, iters):
    rng = np.random.default_rng(seed)
    docs = mkcorp(rng, M, V)
    beta0 = beta_init(rng, K, V)
    kappa0 = beta_init(rng, 1, V)[0]
    m = onp.fCTM(docs, V, K, beta0, kappa0)
    traj = m.train(iter=iters, tol=-1e300)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, V=V, K=K, iters=iters, beta0=beta0, kappa0=kappa0, **pack(docs),
                        eta=m.eta, kappa=m.kappa, mu=m.mu, sigma=m.sigma, invsigma=m.invsigma, beta=m.beta, lam=np.stack(m.lam, 1),
                        vsq=np.stack(m.vsq, 1), logzeta=m.logzeta, tau=np.concatenate(m.tau), tau_old=np.concatenate(m.tau_old),
                        elbo_traj=np.array(traj), sweeps=np.array(m.sweeps, np.int32))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "filtered":      # only the filtered-model fixtures (added in round 2)
        flda_case("flda_m40_v60_k5", 41, 40, 60, 5, 4)
        flda_case("flda_m30_v50_k9_empty", 42, 30, 50, 9, 3, empty_every=7)
        fctm_case("fctm_m30_v50_k4", 51, 30, 50, 4, 3)
        print("filtered-model golden fixtures written to", HERE)
        sys.exit(0)
    lda_case("lda_m40_v60_k3", 11, 40, 60, 3, 5)
    lda_case("lda_m40_v60_k7", 12, 40, 60, 7, 5)
    lda_case("lda_m30_v50_k70_empty", 13, 30, 50, 70, 3, empty_every=7)
    ctm_case("ctm_m40_v60_k5", 21, 40, 60, 5, 4)
    ctpf_case("ctpf_m40_v60_u15_k4", 31, 40, 60, 15, 4, 4, maxRat=3)
    ctpf_case("ctpf_m30_v40_u12_k6_r1", 32, 30, 40, 12, 6, 3, maxRat=1)
    flda_case("flda_m40_v60_k5", 41, 40, 60, 5, 4)
    flda_case("flda_m30_v50_k9_empty", 42, 30, 50, 9, 3, empty_every=7)
    fctm_case("fctm_m30_v50_k4", 51, 30, 50, 4, 3)
    print("golden fixtures written to", HERE)
