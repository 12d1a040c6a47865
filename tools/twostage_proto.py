"""NumPy prototype of the two-stage Hermitian tridiagonalisation (dense -> band -> tridiagonal) with eigenvectors, in the
index conventions of csrc/nls_sb*.h.  Development aid and the reference for the stage hooks' GPU tests; not product code.

Stage 1 (sy2sb): panels of b columns; QR of the panel below the band by shifted CholeskyQR3 + Householder reconstruction
(Y unit lower trapezoidal, T upper triangular: Q = I - Y T Y^H), two-sided update A22 <- Q^H A22 Q as a rank-2b update.
Stage 2 (sb2st): bulge chasing, one sweep per column, reflectors of length <= b; V2[r0, s] = tau, V2[r0+1.., s] = v[1:].
Back-transformation: Z <- Q1 (Q2 Z); Q2 applied in diamond blocks of g sweeps (S descending, k ascending).
"""
from __future__ import annotations

import numpy as np


def larfg(x):
    """LAPACK zlarfg / dlarfg: H = I - tau v v^H with H^H x = beta e1, v[0] = 1.  Returns v, tau, beta."""
    x = np.array(x)
    alpha = x[0]
    xn2 = float(np.sum(np.abs(x[1:]) ** 2))
    if xn2 == 0.0 and np.imag(alpha) == 0.0:
        v = np.zeros_like(x)
        v[0] = 1
        return v, x.dtype.type(0), float(np.real(alpha))
    nrm = np.sqrt(np.abs(alpha) ** 2 + xn2)
    beta = -nrm if np.real(alpha) >= 0 else nrm
    tau = (beta - alpha.real) / beta - 1j * np.imag(alpha) / beta if np.iscomplexobj(x) else (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1
    return v, x.dtype.type(tau), float(beta)


def cholqr3_reconstruct(P):
    """P (m x b, m >= 1) -> Y (m x b unit lower trapezoidal), T (b x b upper), R (b x b upper): P = (I - Y T Y^H)[:, :b] R."""
    m, b = P.shape
    kb = b
    assert m > b
    u = np.finfo(np.float64).eps / 2
    Q = P.copy()
    Rtot = np.eye(b, dtype=P.dtype)
    for it in range(3):
        G = Q.conj().T @ Q
        if it == 0:
            G = G + (11.0 * (m * b + b * (b + 1)) * u * np.trace(G).real + np.finfo(np.float64).tiny) * np.eye(b)
        R = np.linalg.cholesky(G).conj().T  # upper
        Q = Q @ np.linalg.inv(R)
        Rtot = R @ Rtot
    orth = np.linalg.norm(Q.conj().T @ Q - np.eye(b))
    # modified LU of Q - [S; 0]
    L = Q.copy()
    S = np.zeros(b, dtype=P.dtype)
    U = np.zeros((b, b), dtype=P.dtype)
    for i in range(kb):
        d = L[i, i]
        S[i] = -(d / abs(d)) if abs(d) > 0 else -1.0
        L[i, i] = d - S[i]
        U[i, i:] = L[i, i:]
        L[i + 1 :, i] /= L[i, i]
        L[i + 1 :, i + 1 :] -= np.outer(L[i + 1 :, i], L[i, i + 1 :])
        L[i, i] = 1.0
        L[i, i + 1 :] = 0.0
    Y = L
    Y1 = Y[:kb, :kb]
    # [I;0] - Q S^-1 = Y (-U S^-1)  ->  T Y1^H = -U S^-1
    T = -U[:kb, :kb] @ np.diag(1.0 / S[:kb]) @ np.linalg.inv(Y1.conj().T)
    Rh = np.diag(S[:kb]) @ Rtot[:kb, :]  # R factor of the Householder QR: (Q S^-1)(S R)
    return Y[:, :kb], T, Rh, orth


def sy2sb(A, b):
    """Dense Hermitian A (full storage used, lower referenced) -> band matrix (dense storage, lower bandwidth b) + panels."""
    A = np.array(A)
    n = A.shape[0]
    panels = []
    j = 0
    while True:
        m = n - j - b  # rows below the band in column j
        kb = min(b, m - 1)  # columns that have something to annihilate (the last panel may be narrower than b)
        if kb <= 0:
            break
        P = A[j + b :, j : j + kb].copy()
        Y, T, R, orth = cholqr3_reconstruct(P)
        A[j + b :, j : j + kb] = 0
        A[j + b : j + b + kb, j : j + kb] = R
        A[j : j + kb, j + b :] = A[j + b :, j : j + kb].conj().T
        # two-sided update of A[j + kb :, j + kb :] with the reflector block zero-padded by b - kb rows on top: this also applies
        # Q^H from the left to the panel columns kb .. b-1 that the narrow last panel does not factor
        z = b - kb
        Yh = np.vstack([np.zeros((z, kb), dtype=A.dtype), Y])
        A22 = A[j + kb :, j + kb :]
        W = A22 @ (Yh @ T)
        M = T.conj().T @ (Yh.conj().T @ W)
        X = W - 0.5 * Yh @ M
        A22 -= X @ Yh.conj().T + Yh @ X.conj().T
        panels.append((j + b, Y, T, orth))
        j += kb
    return A, panels


def apply_q1(panels, C):
    """C <- Q1 C with Q1 = prod_j (I - Y_j T_j Y_j^H) (first panel leftmost)."""
    for r0, Y, T, _ in reversed(panels):
        C[r0:, :] -= Y @ (T @ (Y.conj().T @ C[r0:, :]))
    return C


def sb2st(Ab, b):
    """Band (dense storage, lower bandwidth b, Hermitian) -> d, e, V2.  Works on a full dense copy for clarity."""
    A = np.array(Ab)
    n = A.shape[0]
    cplx = np.iscomplexobj(A)
    V2 = np.zeros((n, n), dtype=A.dtype)

    def two_sided(r0, L, v, tau):
        D = A[r0 : r0 + L, r0 : r0 + L]
        p = tau * (D @ v)
        w = p - 0.5 * np.conj(tau) * (v.conj() @ p) * v
        D -= np.outer(v, w.conj()) + np.outer(w, v.conj())

    for s in range(n - 1):
        r0 = s + 1
        L = min(b, n - r0)
        v, tau, beta = larfg(A[r0 : r0 + L, s])
        A[r0 : r0 + L, s] = 0
        A[r0, s] = beta
        A[s, r0 : r0 + L] = A[r0 : r0 + L, s].conj()
        V2[r0, s] = tau
        V2[r0 + 1 : r0 + L, s] = v[1:]
        two_sided(r0, L, v, tau)
        while True:
            r1 = r0 + b
            if r1 >= n:
                break
            L1 = min(b, n - r1)
            B = A[r1 : r1 + L1, r0 : r0 + L]
            B -= tau * np.outer(B @ v, v.conj())  # right: B H
            v1, tau1, beta1 = larfg(B[:, 0])
            B[:, 0] = 0
            B[0, 0] = beta1
            B[:, 1:] -= np.conj(tau1) * np.outer(v1, v1.conj() @ B[:, 1:])  # left: H'^H B
            A[r0 : r0 + L, r1 : r1 + L1] = B.conj().T
            V2[r1, s] = tau1
            V2[r1 + 1 : r1 + L1, s] = v1[1:]
            two_sided(r1, L1, v1, tau1)
            r0, L, v, tau = r1, L1, v1, tau1
    d = np.real(np.diag(A)).copy()
    e = np.real(np.diag(A, -1)).copy()
    assert not cplx or np.max(np.abs(np.imag(np.diag(A, -1))), initial=0) < 1e-13 * max(1.0, np.max(np.abs(e), initial=0))
    return d, e, V2


def apply_q2_naive(V2, b, C):
    """C <- Q2 C, Q2 = prod_s prod_k H_{s,k} in generation order (so applied last to first)."""
    n = V2.shape[0]
    for s in range(n - 2, -1, -1):
        r0 = s + 1
        blocks = []
        while r0 < n:
            blocks.append(r0)
            r0 += b
        for r0 in blocks:
            L = min(b, n - r0)
            tau = V2[r0, s]
            v = V2[r0 : r0 + L, s].copy()
            v[0] = 1
            C[r0 : r0 + L, :] -= tau * np.outer(v, v.conj() @ C[r0 : r0 + L, :])
    return C


def apply_q2_diamond(V2, b, g, C):
    """Same product, applied in diamond blocks: groups of g sweeps (S descending), k ascending; each block is a compact WY
    transform I - V T V^H with T^-1 = striu(V^H V) + diag(1 / tau) (tau = 0 -> H = I)."""
    n = V2.shape[0]
    nsweeps = n - 1
    ngroups = (nsweeps + g - 1) // g
    for S in range(ngroups - 1, -1, -1):
        s0, s1 = S * g, min((S + 1) * g, nsweeps)
        k = 0
        while True:
            rtop = s0 + 1 + k * b  # first row of the block (reflector of sweep s0)
            if rtop >= n:
                break
            rbot = min(n, (s1 - 1) + 1 + k * b + b)  # one past the last row (reflector of sweep s1 - 1)
            gg = s1 - s0
            V = np.zeros((rbot - rtop, gg), dtype=C.dtype)
            taus = np.zeros(gg, dtype=C.dtype)
            for i in range(gg):
                s = s0 + i
                r0 = s + 1 + k * b
                if r0 >= n:
                    continue
                L = min(b, n - r0)
                taus[i] = V2[r0, s]
                V[r0 - rtop, i] = 1
                V[r0 - rtop + 1 : r0 - rtop + L, i] = V2[r0 + 1 : r0 + L, s]
            Tinv = np.triu(V.conj().T @ V, 1) + np.diag([1.0 / t if t != 0 else 1e300 for t in taus])
            W = np.linalg.solve(Tinv, V.conj().T @ C[rtop:rbot, :])
            C[rtop:rbot, :] -= V @ W
            k += 1
    return C


def eigh_two_stage(A, b, g):
    n = A.shape[0]
    Ab, panels = sy2sb(A, b)
    d, e, V2 = sb2st(Ab, b)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    lam, Z = np.linalg.eigh(T)
    Z = Z.astype(A.dtype)
    Z = apply_q2_diamond(V2, b, g, Z)
    Z = apply_q1(panels, Z)
    return lam, Z, dict(band=Ab, d=d, e=e, V2=V2, panels=panels)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for cplx in (False, True):
        for n, b, g in [(7, 2, 2), (40, 4, 3), (97, 8, 4), (130, 16, 16), (200, 32, 8), (65, 32, 32), (33, 32, 5), (34, 32, 5)]:
            M = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
            A = (M + M.conj().T) / 2
            lam, Q, st = eigh_two_stage(A, b, g)
            bw = max((abs(i - j) for i in range(n) for j in range(n) if abs(st["band"][i, j]) > 1e-13), default=0)
            Zt = rng.standard_normal((n, 5)).astype(A.dtype)
            dq = np.max(np.abs(apply_q2_naive(st["V2"], b, Zt.copy()) - apply_q2_diamond(st["V2"], b, g, Zt.copy())))
            print(f"cplx={cplx} n={n} b={b} g={g}: bandwidth {bw}  |lam-ref| {np.max(np.abs(lam - np.linalg.eigvalsh(A))):.1e} "
                  f"|AQ-QL| {np.max(np.abs(A @ Q - Q * lam)):.1e}  |Q^HQ-I| {np.max(np.abs(Q.conj().T @ Q - np.eye(n))):.1e}  diamond-vs-naive {dq:.1e} "
                  f"panel orth {max((p[3] for p in st['panels']), default=0):.1e}")
