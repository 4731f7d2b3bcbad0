"""
oracle_np.py -- independent NumPy/SciPy restatement of the reference's CPU train! paths.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (the reference ships no golden vectors and Julia is
not available; see oracle/tmvb_oracle.h).  This module exists to cross-check the C oracle
(oracle/*.c) with a second implementation that shares no code with it: it is written in the
reference's own matrix style, uses scipy.special for digamma/trigamma/lgamma and numpy.linalg for
the dense solves.  Pure-Python loops => small corpora only.

Documents are lists of (terms, counts[, readers, ratings]) with 0-based int ids.
All citations are file:line relative to the reference repository root.
"""
from __future__ import annotations

import math
import numpy as np
from scipy.special import digamma, polygamma, gammaln

EPSILON = float(np.spacing(1e-14))  # src/utils.jl:3  eps(1e-14)


def _trigamma(x):
    return polygamma(1, x)


def additive_logistic_cols(x):
    """src/utils.jl:114-122 with dims=1"""
    x = np.exp(x - x.max(axis=0, keepdims=True)) if x.size else x
    return x / x.sum(axis=0, keepdims=True) if x.size else x


def _xlogx_entropy(p):
    p = np.asarray(p)
    nz = p > 0
    return -float(np.sum(p[nz] * np.log(p[nz])))


class LDA:
    """src/LDA.jl:6-47 (state) -- beta0 must be supplied (the reference draws it from
    Dirichlet(V,1) with Julia's RNG, src/LDA.jl:35)."""

    def __init__(self, docs, V, K, beta0):
        self.docs = [(np.asarray(t, dtype=np.int64), np.asarray(c, dtype=np.float64)) for t, c in docs]
        self.M, self.V, self.K = len(docs), V, K
        self.alpha = np.ones(K)
        self.beta = np.array(beta0, dtype=np.float64).reshape(K, V).copy()
        self.beta_old = self.beta.copy()
        self.beta_temp = np.zeros((K, V))
        e0 = -np.euler_gamma * np.ones(K) - digamma(K)               # :38
        self.Elogtheta = [e0.copy() for _ in range(self.M)]
        self.Elogtheta_old = [e0.copy() for _ in range(self.M)]
        self.gamma = [np.ones(K) for _ in range(self.M)]
        self.phi = None
        self.elbo = 0.0
        self.sweeps = []

    # :150-154
    def update_phi(self, d):
        terms, _ = self.docs[d]
        phi = EPSILON + self.beta[:, terms] * np.exp(self.Elogtheta[d])[:, None]
        self.phi = phi / phi.sum(axis=0, keepdims=True)

    # :143-146
    def update_gamma(self, d):
        _, counts = self.docs[d]
        self.gamma[d] = EPSILON + (self.alpha + self.phi @ counts)

    # :136-139
    def update_Elogtheta(self, d):
        self.Elogtheta_old[d] = self.Elogtheta[d]
        self.Elogtheta[d] = digamma(self.gamma[d]) - digamma(self.gamma[d].sum())

    # :129-132 (duplicate ids: last write wins, as in Julia's A[:,idx] += B)
    def update_beta_doc(self, d):
        terms, counts = self.docs[d]
        self.beta_temp[:, terms] = self.beta_temp[:, terms] + self.phi * counts[None, :]

    # :121-125
    def update_beta(self):
        self.beta_old = self.beta
        self.beta = self.beta_temp / self.beta_temp.sum(axis=1, keepdims=True)
        self.beta_temp = np.zeros((self.K, self.V))

    # :97-118
    def update_alpha(self, niter, ntol):
        Elogtheta_sum = np.sum(np.stack(self.Elogtheta, axis=0), axis=0)
        nu = float(self.K)
        M = self.M
        for _ in range(niter):
            rho = 1.0
            a = self.alpha
            grad = nu / a + M * (digamma(a.sum()) - digamma(a)) + Elogtheta_sum
            h_inv = -1.0 / (M * _trigamma(a) + nu / a ** 2)
            p = (grad - np.dot(grad, h_inv) / (1.0 / (M * _trigamma(a.sum())) + h_inv.sum())) * h_inv
            while np.min(a - rho * p) < 0:
                rho *= 0.5
            new = np.abs(a - rho * p)
            self.alpha = np.sign(a) * np.minimum(new, np.finfo(np.float64).max)   # @finite, src/macros.jl:53-54
            if (rho * np.linalg.norm(grad) < ntol) and (nu / self.K < ntol):
                break
            nu *= 0.5
        self.alpha = self.alpha + EPSILON

    # :83-93 and :50-80
    def update_elbo(self):
        elbo = 0.0
        a = self.alpha
        for d in range(self.M):
            terms, counts = self.docs[d]
            phi = EPSILON + self.beta_old[:, terms] * np.exp(self.Elogtheta_old[d])[:, None]
            phi = phi / phi.sum(axis=0, keepdims=True)
            El, g = self.Elogtheta[d], self.gamma[d]
            e_ptheta = gammaln(a.sum()) - gammaln(a).sum() + np.dot(a - 1.0, El)
            e_pz = np.dot(phi @ counts, El)
            e_pw = float(np.sum((phi * np.log(self.beta[:, terms] + EPSILON)) @ counts))
            if self.K == 1:
                ent = 0.0
            else:
                g0 = g.sum()
                ent = gammaln(g).sum() - gammaln(g0) + (g0 - self.K) * digamma(g0) - np.dot(g - 1.0, digamma(g))
            e_qtheta = -ent
            e_qz = -sum(c * _xlogx_entropy(phi[:, n]) for n, c in enumerate(counts))
            elbo += e_ptheta + e_pz + e_pw - e_qtheta - e_qz
        self.elbo = float(elbo)
        return self.elbo

    # :161-191 + modelutils.jl:574-585
    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        traj = []
        if all(len(t) == 0 for t, _ in self.docs):
            iter = 0
        if checkelbo <= iter:
            self.update_elbo()
        for k in range(1, iter + 1):
            sw = []
            for d in range(self.M):
                s = 0
                for _ in range(viter):
                    s += 1
                    self.update_phi(d)
                    self.update_gamma(d)
                    self.update_Elogtheta(d)
                    if np.linalg.norm(self.Elogtheta[d] - self.Elogtheta_old[d]) < vtol:
                        break
                sw.append(s)
                self.update_beta_doc(d)
            self.sweeps.append(sw)
            self.update_beta()
            self.update_alpha(niter, ntol)
            if checkelbo != math.inf and k % checkelbo == 0:
                old = self.elbo
                new = self.update_elbo()
                traj.append(new)
                if (new - old) < tol:
                    break
            else:
                traj.append(float("nan"))
        return traj


class fLDA(LDA):
    """src/fLDA.jl:6-60 (state): filtered LDA.  beta0 and kappa0 must be supplied (the reference draws both from
    Dirichlet(V, 1) with Julia's RNG, :40-42)."""

    def __init__(self, docs, V, K, beta0, kappa0):
        super().__init__(docs, V, K, beta0)
        self.eta = 0.5                                                  # :38
        self.kappa = np.array(kappa0, dtype=np.float64).copy()
        self.kappa_old = self.kappa.copy()
        self.kappa_temp = np.zeros(V)
        self.tau = [np.full(len(t), self.eta) for t, _ in self.docs]    # :49
        self.tau_old = [x.copy() for x in self.tau]

    # :188-191
    def update_phi(self, d):
        terms, _ = self.docs[d]
        self.phi = additive_logistic_cols(self.tau[d][None, :] * np.log(self.beta[:, terms] + EPSILON) + self.Elogtheta[d][:, None])

    # :180-185
    def update_tau(self, d):
        self.tau_old[d] = self.tau[d]
        terms, _ = self.docs[d]
        with np.errstate(divide="ignore", over="ignore"):
            prod = np.prod(self.beta[:, terms] ** (-self.phi), axis=0) if len(terms) else np.zeros(0)
        self.tau[d] = self.eta / (EPSILON + (self.eta + (1.0 - self.eta) * (self.kappa[terms] * prod)))

    # :159-162, :145-148 (duplicate ids: last write wins)
    def update_beta_doc(self, d):
        terms, counts = self.docs[d]
        self.beta_temp[:, terms] = self.beta_temp[:, terms] + self.phi * (self.tau[d] * counts)[None, :]

    def update_kappa_doc(self, d):
        terms, counts = self.docs[d]
        self.kappa_temp[terms] = self.kappa_temp[terms] + (1.0 - self.tau[d]) * counts

    # :138-142
    def update_kappa(self):
        self.kappa_old = self.kappa
        self.kappa = self.kappa_temp / self.kappa_temp.sum()
        self.kappa_temp = np.zeros(self.V)

    # :122-124
    def update_eta(self):
        self.eta = float(sum(np.dot(self.tau[d], self.docs[d][1]) for d in range(self.M)) / sum(c.sum() for _, c in self.docs))

    # :108-118 and :62-105
    def update_elbo(self):
        elbo = 0.0
        a = self.alpha
        for d in range(self.M):
            terms, counts = self.docs[d]
            phi = additive_logistic_cols(self.tau_old[d][None, :] * np.log(self.beta_old[:, terms] + EPSILON) + self.Elogtheta_old[d][:, None])
            El, g, tau = self.Elogtheta[d], self.gamma[d], self.tau[d]
            e_ptheta = gammaln(a.sum()) - gammaln(a).sum() + np.dot(a - 1.0, El)
            tc = float(np.dot(tau, counts))
            e_pc = math.log(EPSILON + self.eta ** tc * (1.0 - self.eta) ** (float(counts.sum()) - tc))
            e_pz = np.dot(phi @ counts, El) if len(terms) else 0.0
            e_pw = (float(np.sum((phi * np.log(self.beta[:, terms] + EPSILON)) @ (counts * tau))) if len(terms) else 0.0) \
                + float(np.dot(counts * (1.0 - tau), np.log(self.kappa[terms] + EPSILON)))
            if self.K == 1:
                ent = 0.0
            else:
                g0 = g.sum()
                ent = gammaln(g).sum() - gammaln(g0) + (g0 - self.K) * digamma(g0) - np.dot(g - 1.0, digamma(g))
            e_qc = -sum(c * _xlogx_entropy(np.array([t, 1.0 - t])) for t, c in zip(tau, counts))
            e_qz = -sum(c * _xlogx_entropy(phi[:, n]) for n, c in enumerate(counts))
            elbo += e_ptheta + e_pc + e_pz + e_pw + ent - e_qc - e_qz
        self.elbo = float(elbo)
        return self.elbo

    # :213-247 + modelutils.jl:574-585
    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        traj = []
        if all(len(t) == 0 for t, _ in self.docs):
            iter = 0
        if checkelbo <= iter:
            self.update_elbo()
        for k in range(1, iter + 1):
            sw = []
            for d in range(self.M):
                s = 0
                for _ in range(viter):
                    s += 1
                    self.update_phi(d)
                    self.update_tau(d)
                    self.update_gamma(d)
                    self.update_Elogtheta(d)
                    if np.linalg.norm(self.Elogtheta[d] - self.Elogtheta_old[d]) < vtol:
                        break
                sw.append(s)
                self.update_beta_doc(d)
                self.update_kappa_doc(d)
            self.sweeps.append(sw)
            self.update_beta()
            self.update_kappa()
            self.update_alpha(niter, ntol)
            self.update_eta()
            if checkelbo != math.inf and k % checkelbo == 0:
                old = self.elbo
                new = self.update_elbo()
                traj.append(new)
                if (new - old) < tol:
                    break
            else:
                traj.append(float("nan"))
        return traj


class CTM:
    """src/CTM.jl:6-53"""

    def __init__(self, docs, V, K, beta0):
        self.docs = [(np.asarray(t, dtype=np.int64), np.asarray(c, dtype=np.float64)) for t, c in docs]
        self.M, self.V, self.K = len(docs), V, K
        self.C = [float(c.sum()) for _, c in self.docs]
        self.mu = np.zeros(K)
        self.sigma = np.eye(K)
        self.invsigma = np.eye(K)
        self.beta = np.array(beta0, dtype=np.float64).reshape(K, V).copy()
        self.beta_old = self.beta.copy()
        self.beta_temp = np.zeros((K, V))
        self.lam = [np.zeros(K) for _ in range(self.M)]
        self.lam_old = [np.zeros(K) for _ in range(self.M)]
        self.vsq = [np.ones(K) for _ in range(self.M)]
        self.logzeta = np.full(self.M, 0.5)
        self.phi = None
        self.elbo = 0.0

    def update_phi(self, d):          # :175-178
        terms, _ = self.docs[d]
        with np.errstate(divide="ignore"):
            self.phi = additive_logistic_cols(np.log(self.beta[:, terms]) + self.lam[d][:, None])

    def update_logzeta(self, d):      # :169-171
        x = self.lam[d] + 0.5 * self.vsq[d]
        m = x.max()
        self.logzeta[d] = m + math.log(np.exp(x - m).sum())

    def update_vsq(self, d, niter, ntol):   # :146-165
        C = self.C[d]
        v = self.vsq[d].copy()
        for i in range(self.K):
            for _ in range(niter):
                rho = 1.0
                ex = math.exp(self.lam[d][i] + 0.5 * v[i] - self.logzeta[d])
                grad = -0.5 * (self.invsigma[i, i] + C * ex - 1.0 / v[i])
                ih = -1.0 / (0.25 * C * ex + 0.5 / v[i] ** 2)
                p = ih * grad
                while v[i] - rho * p <= 0:
                    rho *= 0.5
                v[i] -= rho * p
                if rho * abs(grad) < ntol:
                    break
        self.vsq[d] = v + EPSILON

    def update_lambda(self, d, niter, ntol):  # :129-142
        self.lam_old[d] = self.lam[d]
        _, counts = self.docs[d]
        C = self.C[d]
        for _ in range(niter):
            ex = np.exp(self.lam[d] + 0.5 * self.vsq[d] - self.logzeta[d])
            grad = self.invsigma @ (self.mu - self.lam[d]) + self.phi @ counts - C * ex
            H = self.invsigma + C * np.diag(ex)
            self.lam[d] = self.lam[d] + np.linalg.solve(H, grad)
            if np.linalg.norm(grad) < ntol:
                break

    def update_beta_doc(self, d):     # :122-125
        terms, counts = self.docs[d]
        self.beta_temp[:, terms] = self.beta_temp[:, terms] + self.phi * counts[None, :]

    def update_beta(self):            # :114-118
        self.beta_old = self.beta
        self.beta = self.beta_temp / self.beta_temp.sum(axis=1, keepdims=True)
        self.beta_temp = np.zeros((self.K, self.V))

    def update_sigma(self):           # :108-111
        L = np.stack(self.lam, axis=1) - self.mu[:, None]
        S = (np.diag(np.sum(np.stack(self.vsq, axis=0), axis=0)) + L @ L.T) / self.M
        S = np.triu(S) + np.triu(S, 1).T      # Symmetric() reads the upper triangle
        self.sigma = S
        inv = np.linalg.inv(S)
        self.invsigma = 0.5 * (inv + inv.T)

    def update_mu(self):              # :102-104
        self.mu = np.sum(np.stack(self.lam, axis=0), axis=0) / self.M

    def update_elbo(self):            # :89-98, :56-86
        elbo = 0.0
        _, logdet = np.linalg.slogdet(self.invsigma)
        K = self.K
        for d in range(self.M):
            terms, counts = self.docs[d]
            with np.errstate(divide="ignore"):
                phi = additive_logistic_cols(np.log(self.beta_old[:, terms]) + self.lam_old[d][:, None])
            l, v, lz, C = self.lam[d], self.vsq[d], self.logzeta[d], self.C[d]
            df = l - self.mu
            e_peta = 0.5 * (logdet - K * math.log(2 * math.pi) - np.dot(np.diag(self.invsigma), v) - df @ self.invsigma @ df)
            e_pz = np.dot(phi.T @ l, counts) - C * (np.exp(l + 0.5 * v - lz).sum() + lz - 1.0)
            e_pw = float(np.sum((phi * np.log(self.beta[:, terms] + EPSILON)) @ counts))
            e_qeta = -0.5 * (K * (1 + math.log(2 * math.pi)) + np.log(v).sum())
            e_qz = -sum(c * _xlogx_entropy(phi[:, n]) for n, c in enumerate(counts))
            elbo += e_peta + e_pz + e_pw - e_qeta - e_qz
        self.elbo = float(elbo)
        return self.elbo

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):   # :185-217
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        traj = []
        if all(len(t) == 0 for t, _ in self.docs):
            iter = 0
        if checkelbo <= iter:
            self.update_elbo()
        for k in range(1, iter + 1):
            for d in range(self.M):
                for _ in range(viter):
                    self.update_phi(d)
                    self.update_logzeta(d)
                    self.update_vsq(d, niter, ntol)
                    self.update_lambda(d, niter, ntol)
                    if np.linalg.norm(self.lam[d] - self.lam_old[d]) < vtol:
                        break
                self.update_beta_doc(d)
            self.update_beta()
            self.update_sigma()
            self.update_mu()
            if checkelbo != math.inf and k % checkelbo == 0:
                old = self.elbo
                new = self.update_elbo()
                traj.append(new)
                if (new - old) < tol:
                    break
            else:
                traj.append(float("nan"))
        return traj


class fCTM(CTM):
    """src/fCTM.jl:6-64: filtered CTM.  beta0 and kappa0 must be supplied (Julia RNG in the reference, :40-43)."""

    def __init__(self, docs, V, K, beta0, kappa0):
        super().__init__(docs, V, K, beta0)
        self.eta = 0.5                                                  # :37
        self.kappa = np.array(kappa0, dtype=np.float64).copy()
        self.kappa_old = self.kappa.copy()
        self.kappa_temp = np.zeros(V)
        self.tau = [np.full(len(t), self.eta) for t, _ in self.docs]    # :57
        self.tau_old = [x.copy() for x in self.tau]
        self.sweeps = []

    def update_phi(self, d):          # :216-219
        terms, _ = self.docs[d]
        self.phi = additive_logistic_cols(self.tau[d][None, :] * np.log(self.beta[:, terms] + EPSILON) + self.lam[d][:, None])

    def update_tau(self, d):          # :208-213
        self.tau_old[d] = self.tau[d]
        terms, _ = self.docs[d]
        with np.errstate(divide="ignore", over="ignore"):
            prod = np.prod(self.beta[:, terms] ** (-self.phi), axis=0) if len(terms) else np.zeros(0)
        self.tau[d] = self.eta / (EPSILON + (self.eta + (1.0 - self.eta) * (self.kappa[terms] * prod)))

    def update_beta_doc(self, d):     # :155-158
        terms, counts = self.docs[d]
        self.beta_temp[:, terms] = self.beta_temp[:, terms] + self.phi * (self.tau[d] * counts)[None, :]

    def update_kappa_doc(self, d):    # :141-144
        terms, counts = self.docs[d]
        self.kappa_temp[terms] = self.kappa_temp[terms] + (1.0 - self.tau[d]) * counts

    def update_kappa(self):           # :134-138
        self.kappa_old = self.kappa
        self.kappa = self.kappa_temp / self.kappa_temp.sum()
        self.kappa_temp = np.zeros(self.V)

    def update_elbo(self):            # :105-115, :68-102
        elbo = 0.0
        _, logdet = np.linalg.slogdet(self.invsigma)
        K = self.K
        for d in range(self.M):
            terms, counts = self.docs[d]
            phi = additive_logistic_cols(self.tau_old[d][None, :] * np.log(self.beta_old[:, terms] + EPSILON) + self.lam_old[d][:, None])
            l, v, lz, C, tau = self.lam[d], self.vsq[d], self.logzeta[d], self.C[d], self.tau[d]
            df = l - self.mu
            e_peta = 0.5 * (logdet - K * math.log(2 * math.pi) - np.dot(np.diag(self.invsigma), v) - df @ self.invsigma @ df)
            tc = float(np.dot(tau, counts))
            e_pc = math.log(EPSILON + self.eta ** tc * (1.0 - self.eta) ** (C - tc))
            e_pz = (np.dot(phi.T @ l, counts) if len(terms) else 0.0) - C * (np.exp(l + 0.5 * v - lz).sum() + lz - 1.0)
            e_pw = (float(np.sum((phi * np.log(self.beta[:, terms] + EPSILON)) @ (counts * tau))) if len(terms) else 0.0) \
                + float(np.dot(counts * (1.0 - tau), np.log(self.kappa[terms] + EPSILON)))
            e_qeta = -0.5 * (K * (1 + math.log(2 * math.pi)) + np.log(v).sum())
            e_qc = -sum(c * _xlogx_entropy(np.array([t, 1.0 - t])) for t, c in zip(tau, counts))
            e_qz = -sum(c * _xlogx_entropy(phi[:, n]) for n, c in enumerate(counts))
            elbo += e_peta + e_pc + e_pz + e_pw - e_qeta - e_qc - e_qz
        self.elbo = float(elbo)
        return self.elbo

    def train(self, iter=150, tol=1.0, niter=1000, ntol=None, viter=10, vtol=None, checkelbo=1):   # :226-262
        ntol = 1.0 / self.K ** 2 if ntol is None else ntol
        vtol = 1.0 / self.K ** 2 if vtol is None else vtol
        traj = []
        if all(len(t) == 0 for t, _ in self.docs):
            iter = 0
        if checkelbo <= iter:
            self.update_elbo()
        for k in range(1, iter + 1):
            sw = []
            for d in range(self.M):
                s = 0
                for _ in range(viter):
                    s += 1
                    self.update_phi(d)
                    self.update_tau(d)
                    self.update_logzeta(d)
                    self.update_lambda(d, niter, ntol)
                    self.update_vsq(d, niter, ntol)
                    if np.linalg.norm(self.lam[d] - self.lam_old[d]) < vtol:
                        break
                sw.append(s)
                self.update_beta_doc(d)
                self.update_kappa_doc(d)
            self.sweeps.append(sw)
            self.update_beta()
            self.update_kappa()
            self.update_sigma()
            self.update_mu()                    # update_eta! is commented out in the reference (:253)
            if checkelbo != math.inf and k % checkelbo == 0:
                old = self.elbo
                new = self.update_elbo()
                traj.append(new)
                if (new - old) < tol:
                    break
            else:
                traj.append(float("nan"))
        return traj


def _binom_lgamma_sum(n, p):
    """sum_{y=0}^{n} pdf(Binomial(n,p), y) * lgamma(y+1)   (src/CTPF.jl:116)"""
    n = int(n)
    if n <= 1:
        return 0.0
    from scipy.stats import binom
    y = np.arange(0, n + 1)
    return float(np.sum(binom.pmf(y, n, p) * gammaln(y + 1.0)))


def _multinomial_entropy(n, p):
    return -gammaln(n + 1.0) + n * _xlogx_entropy(p) + sum(_binom_lgamma_sum(n, pi) for pi in p)


def _gamma_entropy(a, scale):
    return a + np.log(scale) + gammaln(a) + (1.0 - a) * digamma(a)


class CTPF:
    """src/CTPF.jl:6-108"""

    def __init__(self, docs, V, U, K, alef0):
        self.docs = [(np.asarray(t, dtype=np.int64), np.asarray(c, dtype=np.float64),
                      np.asarray(r, dtype=np.int64), np.asarray(q, dtype=np.float64)) for t, c, r, q in docs]
        self.M, self.V, self.U, self.K = len(docs), V, U, K
        self.a = self.b = self.c = self.d = self.e = self.f = self.g = self.h = 0.1   # :81
        self.alef = np.array(alef0, dtype=np.float64).reshape(K, V).copy()
        self.alef_old = self.alef.copy()
        self.alef_temp = np.full((K, V), self.a)
        self.he = np.ones((K, U)); self.he_old = self.he.copy(); self.he_temp = np.full((K, U), self.e)
        self.bet = np.ones(K); self.bet_old = np.ones(K)
        self.vav = np.ones(K); self.vav_old = np.ones(K)
        self.dalet = np.ones(K); self.dalet_old = np.ones(K)
        self.het = np.ones(K); self.het_old = np.ones(K)
        self.gimel = [np.ones(K) for _ in range(self.M)]; self.gimel_old = [np.ones(K) for _ in range(self.M)]
        self.zayin = [np.ones(K) for _ in range(self.M)]; self.zayin_old = [np.ones(K) for _ in range(self.M)]
        self.phi = None; self.xi = None
        self.elbo = 0.0

    @staticmethod
    def _xi(gimel, zayin, dalet, het, vav, he, readers):
        top = (digamma(gimel) - np.log(dalet) - np.log(vav))[:, None] + digamma(he[:, readers])
        bot = (digamma(zayin) - np.log(het) - np.log(vav))[:, None] + digamma(he[:, readers])
        return additive_logistic_cols(np.vstack([top, bot]))

    @staticmethod
    def _phi(gimel, dalet, bet, alef, terms):
        return additive_logistic_cols((digamma(gimel) - np.log(dalet) - np.log(bet))[:, None] + digamma(alef[:, terms]))

    def train(self, iter=150, tol=1.0, viter=10, vtol=None, checkelbo=1):   # :344-376
        K = self.K
        vtol = 1.0 / K ** 2 if vtol is None else vtol
        traj = []
        if all(len(t) == 0 for t, _, _, _ in self.docs):
            iter = 0
        if checkelbo <= iter:
            self.update_elbo()
        for k in range(1, iter + 1):
            for d in range(self.M):
                terms, counts, readers, ratings = self.docs[d]
                for _ in range(viter):
                    self.xi = self._xi(self.gimel[d], self.zayin[d], self.dalet, self.het, self.vav, self.he, readers)   # :334
                    self.phi = self._phi(self.gimel[d], self.dalet, self.bet, self.alef, terms)                          # :327
                    self.zayin_old[d] = self.zayin[d]
                    self.zayin[d] = self.g + self.xi[K:, :] @ ratings                                                    # :318
                    self.gimel_old[d] = self.gimel[d]
                    self.gimel[d] = self.c + self.phi @ counts + self.xi[:K, :] @ ratings                                # :309
                    if np.linalg.norm(self.gimel[d] - self.gimel_old[d]) < vtol:
                        break
                self.he_temp[:, readers] = self.he_temp[:, readers] + (self.xi[:K, :] + self.xi[K:, :]) * ratings[None, :]   # :274
                self.alef_temp[:, terms] = self.alef_temp[:, terms] + self.phi * counts[None, :]                            # :259
            # :366-371
            self.he_old = self.he; self.he = self.he_temp; self.he_temp = np.full((K, self.U), self.e)
            self.alef_old = self.alef; self.alef = self.alef_temp; self.alef_temp = np.full((K, self.V), self.a)
            self.dalet_old = self.dalet
            self.dalet = self.d + self.alef.sum(axis=1) / self.bet + self.he.sum(axis=1) / self.vav
            self.het_old = self.het
            self.het = self.h + self.he.sum(axis=1) / self.vav
            gs = np.sum(np.stack(self.gimel, axis=0), axis=0)
            zs = np.sum(np.stack(self.zayin, axis=0), axis=0)
            self.bet_old = self.bet
            self.bet = self.b + gs / self.dalet
            self.vav_old = self.vav
            self.vav = self.f + gs / self.dalet + zs / self.het
            if checkelbo != math.inf and k % checkelbo == 0:
                old = self.elbo
                new = self.update_elbo()
                traj.append(new)
                if (new - old) < tol:
                    break
            else:
                traj.append(float("nan"))
        return traj

    def update_elbo(self):            # :234-247, :111-231
        K, V, U = self.K, self.V, self.U
        lb, lv, ld, lh = np.log(self.bet), np.log(self.vav), np.log(self.dalet), np.log(self.het)
        elbo = V * K * (self.a * math.log(self.b) - gammaln(self.a))
        elbo += np.sum((self.a - 1) * (digamma(self.alef) - lb[:, None]) - self.b * self.alef / self.bet[:, None])
        elbo += U * K * (self.e * math.log(self.f) - gammaln(self.e))
        elbo += np.sum((self.e - 1) * (digamma(self.he) - lv[:, None]) - self.f * self.he / self.vav[:, None])
        elbo += np.sum(_gamma_entropy(self.alef, 1.0 / self.bet[:, None]))
        elbo += np.sum(_gamma_entropy(self.he, 1.0 / self.vav[:, None]))
        rs_he, rs_alef = self.he.sum(axis=1), self.alef.sum(axis=1)
        for d in range(self.M):
            terms, counts, readers, ratings = self.docs[d]
            phi = self._phi(self.gimel_old[d], self.dalet_old, self.bet_old, self.alef_old, terms)
            xi = self._xi(self.gimel_old[d], self.zayin_old[d], self.dalet_old, self.het_old, self.vav_old, self.he_old, readers)
            gi, za = self.gimel[d], self.zayin[d]
            e = -np.dot(gi / (self.dalet * self.vav), rs_he)
            for u, (re, ra) in enumerate(zip(readers, ratings)):
                for i in range(K):
                    e += ra * xi[i, u] * (digamma(gi[i]) - ld[i] + digamma(self.he[i, re]) - lv[i]) - _binom_lgamma_sum(ra, xi[i, u])
            e += -np.dot(za / (self.het * self.vav), rs_he)
            for u, (re, ra) in enumerate(zip(readers, ratings)):
                for i in range(K):
                    e += ra * xi[K + i, u] * (digamma(za[i]) - lh[i] + digamma(self.he[i, re]) - lv[i]) - _binom_lgamma_sum(ra, xi[K + i, u])
            e += -np.dot(gi / (self.dalet * self.bet), rs_alef)
            for n, (j, c) in enumerate(zip(terms, counts)):
                for i in range(K):
                    e += c * phi[i, n] * (digamma(gi[i]) - ld[i] + digamma(self.alef[i, j]) - lb[i]) - _binom_lgamma_sum(c, phi[i, n])
            e += K * (self.c * math.log(self.d) - gammaln(self.c)) + np.sum((self.c - 1) * (digamma(gi) - ld) - self.d * gi / self.dalet)
            e += K * (self.g * math.log(self.h) - gammaln(self.g)) + np.sum((self.g - 1) * (digamma(za) - lh) - self.h * za / self.het)
            e += sum(_multinomial_entropy(ra, xi[:, u]) for u, ra in enumerate(ratings))
            e += sum(_multinomial_entropy(c, phi[:, n]) for n, c in enumerate(counts))
            e += np.sum(_gamma_entropy(gi, 1.0 / self.dalet)) + np.sum(_gamma_entropy(za, 1.0 / self.het))
            elbo += e
        self.elbo = float(elbo)
        return self.elbo
