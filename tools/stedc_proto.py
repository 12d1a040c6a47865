"""NumPy prototype of the library's own divide-and-conquer eigensolver for symmetric tridiagonal matrices (csrc/nls_stedc.h): the algorithm
the HIP kernels implement, kept because its index conventions and tolerances are what the GPU tests compare against.

    T = diag(d) + offdiag(e)  ->  lam (ascending), Q with T Q = Q diag(lam)

Cuppen's divide and conquer as LAPACK's dstedc / dlaed0-dlaed4 organise it, with two simplifications that suit a GPU:
  * the tree is level-synchronous: leaves of LEAF rows, then merges of neighbouring blocks of LEAF 2^l rows;
  * the secular equation is solved for the offset from the nearest pole by dlaed4's iteration ("the middle way": a two-pole rational
    model of the function), every step that leaves the bracket replaced by a bisection step in the logarithm of the offset (one wave per
    8 roots on the GPU, the poles spread over its lanes).
    Orthogonality does not rest on the accuracy of the roots: as in dlaed3, the vector z is recomputed from the
    computed roots (Gu / Eisenstat: the roots are then the EXACT eigenvalues of D + rho zhat zhat^T), and the differences d_i - lam_j are
    formed as (d_i - d_origin) - mu_j, never from the rounded lam_j.
Deflation is dlaed2's: negligible components of z, and pairs of close poles rotated into one (the rotations are applied to the eigenvector
columns)."""

from __future__ import annotations

import numpy as np

EPS = np.finfo(np.float64).eps
LEAF = 32


def secular_roots(dl, w, rho):
    """Roots of 1 + rho sum_i w_i^2 / (dl_i - lam) = 0, dl ascending and distinct, w nonzero, rho > 0.
    Returns (origin index o_j, offset mu_j) with lam_j = dl[o_j] + mu_j."""
    k = dl.size
    w2 = w * w
    org = np.empty(k, dtype=np.intp)
    mu = np.empty(k)
    for j in range(k):
        if j < k - 1:
            gap = dl[j + 1] - dl[j]
            mid = 0.5 * gap
            # secular function at the midpoint of the interval, relative to pole j
            fm = 1.0 + rho * np.sum(w2 / ((dl - dl[j]) - mid))
            o = j if fm >= 0.0 else j + 1
            hi = mid
        else:
            o = k - 1
            hi = rho * np.sum(w2)  # lam_k <= dl_k + rho |w|^2
            if not hi > 0.0:
                hi = np.finfo(float).tiny
        delta = dl - dl[o]
        right = o == j  # root right of its origin: mu in (0, hi]
        # a point of the interval where the function has the sign it has next to the origin pole
        if right:
            others = rho * np.sum(np.where(delta > 0, w2 / np.maximum(delta - hi, np.finfo(float).tiny), 0.0))
            lo = 0.5 * rho * w2[o] / (1.0 + others)
        else:
            others = rho * np.sum(np.where(delta < 0, w2 / np.maximum(-(delta + hi), np.finfo(float).tiny), 0.0))
            lo = 0.5 * rho * w2[o] / max(others - 1.0, 1.0)
        lo = min(lo, hi)
        lo = max(lo, np.finfo(float).tiny)
        # "the middle way" (Li 1994; dlaed4's iteration): with psi = sum over the poles left of the root, phi = over those right of it, the
        # secular function is modelled by c + a / (dL - t) + b / (dR - t) (dL, dR: the two poles that enclose the root) matching value and
        # slope of psi and phi separately; the model's root inside the interval is the next iterate.  Every iterate that leaves the bracket is
        # replaced by a bisection step in the logarithm of the distance to the origin.  tau: signed offset from the origin pole.
        last = j == k - 1
        dL = 0.0 if right else -(dl[j + 1] - dl[j])
        dR = np.inf if last else ((dl[j + 1] - dl[j]) if right else 0.0)
        tlo, thi = (lo, hi) if right else (-hi, -lo)
        tau = min(2.0 * lo, hi) if right else -min(2.0 * lo, hi)
        left_mask = np.arange(k) <= j
        for _ in range(80):
            den = delta - tau
            t = w2 / den
            psi, dpsi = np.sum(t[left_mask]), np.sum((t / den)[left_mask])
            phi, dphi = np.sum(t[~left_mask]), np.sum((t / den)[~left_mask])
            f = 1.0 / rho + psi + phi
            if f < 0.0:
                tlo = tau
            else:
                thi = tau
            if abs(f) <= EPS * (8.0 * (phi - psi) + 2.0 / rho + abs(tau) * (dpsi + dphi)):
                break  # dlaed4's criterion: the function value is at the level of its own rounding error
            DL = dL - tau
            if last:
                c = f - DL * dpsi
                a = dpsi * DL * DL
                eta = DL + a / c if c != 0.0 else np.inf  # root of c + a / (DL - eta)
            else:
                DR = dR - tau
                c = f - DL * dpsi - DR * dphi
                a, b2 = dpsi * DL * DL, dphi * DR * DR
                B = c * (DL + DR) + a + b2
                C = DL * DR * f
                disc = np.sqrt(abs(B * B - 4.0 * c * C))
                if c == 0.0:
                    eta = C / B if B != 0.0 else np.inf
                elif B <= 0.0:
                    eta = (B - disc) / (2.0 * c)
                else:
                    eta = 2.0 * C / (B + disc)
            tnew = tau + eta
            if not (tlo < tnew < thi):
                tnew = np.sign(tau) * np.sqrt(abs(tlo)) * np.sqrt(abs(thi))
            stop = abs(tnew - tau) <= 8.0 * EPS * abs(tnew) or abs(thi - tlo) <= 8.0 * EPS * abs(tnew)
            tau = tnew
            if stop:
                break
        m = abs(tau)
        org[j] = o
        mu[j] = m if right else -m
    return org, mu


def merge(dd, z, rho, Qb):
    """Eigen-decomposition of diag(dd) + rho z z^T (rho > 0) given the current eigenvector block Qb (columns in the order of dd).
    Returns (lam ascending, Q)."""
    m = dd.size
    perm = np.argsort(dd, kind="stable")
    d = dd[perm].copy()
    zz = z[perm].copy()
    Q = Qb[:, perm].copy()
    tol = 8.0 * EPS * max(np.max(np.abs(d)), np.max(np.abs(zz)))
    if rho * np.max(np.abs(zz)) <= tol:
        return d, Q
    keep, defl = [], []
    pj = -1
    for j in range(m):
        if rho * abs(zz[j]) <= tol:
            defl.append(j)
            continue
        if pj < 0:
            pj = j
            continue
        s, c = zz[pj], zz[j]
        tau = np.hypot(c, s)
        t = d[j] - d[pj]
        c, s = c / tau, -s / tau
        if abs(t * c * s) <= tol:  # the two poles are close: rotate z[pj] away
            zz[j], zz[pj] = tau, 0.0
            qp, qj = Q[:, pj].copy(), Q[:, j].copy()
            Q[:, pj] = c * qp + s * qj
            Q[:, j] = c * qj - s * qp
            t = d[pj] * c * c + d[j] * s * s
            d[j] = d[pj] * s * s + d[j] * c * c
            d[pj] = t
            defl.append(pj)
            pj = j
        else:
            keep.append(pj)
            pj = j
    if pj >= 0:
        keep.append(pj)
    keep = np.asarray(keep, dtype=np.intp)
    defl = np.asarray(defl, dtype=np.intp)
    k = keep.size
    dl, w = d[keep], zz[keep]
    assert np.all(np.diff(dl) > 0)
    org, mu = secular_roots(dl, w, rho)
    # delta[i, j] = dl_i - lam_j, formed from the origin and the offset
    delta = (dl[:, None] - dl[org][None, :]) - mu[None, :]
    # zhat_i^2 = prod_j (lam_j - dl_i) / prod_{j != i} (dl_j - dl_i)   (up to the common factor rho)
    num = -delta  # lam_j - dl_i
    den = dl[None, :] - dl[:, None]
    np.fill_diagonal(den, 1.0)
    ratio = num / den
    zhat = np.sqrt(np.abs(np.prod(ratio, axis=1))) * np.where(w < 0, -1.0, 1.0)
    U = zhat[:, None] / delta
    U /= np.linalg.norm(U, axis=0, keepdims=True)
    lam_new = dl[org] + mu
    Qn = Q[:, keep] @ U
    lam = np.concatenate([lam_new, d[defl]])
    Qall = np.concatenate([Qn, Q[:, defl]], axis=1)
    o = np.argsort(lam, kind="stable")
    return lam[o], Qall[:, o]


def leaf_eigh(d, e):
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    return np.linalg.eigh(T)


def stedc(d, e, leaf=LEAF):
    d = np.asarray(d, dtype=np.float64).copy()
    e = np.asarray(e, dtype=np.float64)
    n = d.size
    for p in range(leaf, n, leaf):  # tear: T = diag(T1', T2') + |rho| v v^T, v = (e_last; sign(rho) e_first)
        d[p - 1] -= abs(e[p - 1])
        d[p] -= abs(e[p - 1])
    Q = np.zeros((n, n))
    lam = np.zeros(n)
    for b0 in range(0, n, leaf):
        b1 = min(b0 + leaf, n)
        lam[b0:b1], Q[b0:b1, b0:b1] = leaf_eigh(d[b0:b1], e[b0 : b1 - 1])
    size = leaf
    while size < n:
        for b0 in range(0, n, 2 * size):
            mid, b1 = b0 + size, min(b0 + 2 * size, n)
            if mid >= n:
                continue
            rho = e[mid - 1]
            z = np.concatenate([Q[mid - 1, b0:mid], np.sign(rho) * Q[mid, mid:b1] if rho != 0 else Q[mid, mid:b1]]) / np.sqrt(2.0)
            lam[b0:b1], Q[b0:b1, b0:b1] = merge(lam[b0:b1], z, 2.0 * abs(rho), Q[b0:b1, b0:b1])
        size *= 2
    return lam, Q


def check(d, e, name):
    n = len(d)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    lam, Q = stedc(d, e)
    ref = np.linalg.eigvalsh(T)
    nrm = max(np.max(np.abs(ref)), 1e-300)
    r = (np.max(np.abs(lam - ref)) / nrm, np.max(np.abs(T @ Q - Q * lam[None, :])) / nrm, np.max(np.abs(Q.T @ Q - np.eye(n))))
    print(f"{name:38s} n={n:5d}  eig {r[0]:.2e}  resid {r[1]:.2e}  orth {r[2]:.2e}")
    return r


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (5, 31, 32, 33, 64, 100, 257, 700):
        check(rng.standard_normal(n), rng.standard_normal(n - 1), "random")
    check(np.arange(1.0, 201.0), np.zeros(199), "diagonal")
    check(np.full(300, 2.0), np.full(299, 1.0), "toeplitz (2, 1)")
    m = 50
    wd = np.abs(np.arange(-m, m + 1)).astype(float)
    check(wd, np.ones(2 * m), "Wilkinson W101")
    glued = np.concatenate([wd] * 4)
    ge = np.concatenate([np.ones(2 * m), [1e-8], np.ones(2 * m), [1e-8], np.ones(2 * m), [1e-8], np.ones(2 * m)])
    check(glued, ge, "glued Wilkinson 4 x W101")
    check(np.ones(257), 1e-9 * rng.standard_normal(256), "identity + tiny couplings")
    # the tridiagonal matrix of an RBF kernel (the dual path's spectrum: a few large eigenvalues, most near zero)
    import scipy.linalg as sla

    X = rng.standard_normal((600, 8)) * 0.4
    K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)) + 1.0
    H = sla.hessenberg(K)
    check(np.diag(H).copy(), np.diag(H, 1).copy(), "tridiagonalised RBF kernel")
    # graded
    check(10.0 ** np.linspace(0, -14, 400), 10.0 ** np.linspace(-1, -15, 399), "graded 1 .. 1e-14")
