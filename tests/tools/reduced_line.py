"""Prototype (NumPy, CPU) of the REDUCED line solve used by k_line_sweep_q4 / k_line_factor4:
the L-directed unknowns of a line are eliminated analytically, leaving a block-tridiagonal system of the
4 transverse unknowns per node (4x4 blocks, symmetric off-diagonal blocks B_i = D_i + mu_i u_i u_i^T), which
is factorised two-sided.  Checked here against the oracle's line smoothers before the HIP kernel is written;
the HIP kernels follow this file statement by statement.  Development tool, not product code."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))


def axes(direction):
    return {1: (0, 1, 2), 2: (1, 0, 2), 3: (2, 0, 1)}[direction]


class LineSys:
    """Coefficients of the line (jP, jQ) along axis L: u_i (4), d_i (4), m_i, M_TT(i) (4x4), rhs terms."""

    def __init__(self, vnC, e, s, eta, zeta, h, direction):
        self.L, self.P, self.Q = axes(direction)
        self.nC = vnC
        self.h = h
        nx, ny, nz = vnC
        self.sh = [(nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz)]
        off = np.cumsum([0] + [int(np.prod(a)) for a in self.sh])
        self.E = [e[off[c]:off[c + 1]].reshape(self.sh[c], order='F') for c in range(3)]
        self.S = [s[off[c]:off[c + 1]].reshape(self.sh[c], order='F') for c in range(3)]
        self.eta, self.zeta = eta, zeta

    def idx(self, vL, vP, vQ):
        t = [0, 0, 0]
        t[self.L], t[self.P], t[self.Q] = vL, vP, vQ
        return tuple(t)

    def z(self, iL, cP, cQ):
        return self.zeta[self.idx(iL, cP, cQ)]

    def coef(self, i, jP, jQ):
        """a (4 real), d (4 real) of block i (`left`), all from zeta at L-cell i."""
        L, P, Q = self.L, self.P, self.Q
        ihL = 1.0 / self.h[L][i]
        ihP = [1.0 / self.h[P][jP - 1], 1.0 / self.h[P][jP]]
        ihQ = [1.0 / self.h[Q][jQ - 1], 1.0 / self.h[Q][jQ]]
        zz = [[self.z(i, jP - 1 + a, jQ - 1 + b) for b in range(2)] for a in range(2)]   # [P side][Q side]
        rsP = [zz[0][0] + zz[0][1], zz[1][0] + zz[1][1]]     # sum over Q at P side
        rsQ = [zz[0][0] + zz[1][0], zz[0][1] + zz[1][1]]     # sum over P at Q side
        u = np.array([0.5 * ihP[0] * rsP[0] * ihL, -0.5 * ihP[1] * rsP[1] * ihL,
                      0.5 * ihQ[0] * rsQ[0] * ihL, -0.5 * ihQ[1] * rsQ[1] * ihL])
        d = -0.5 * ihL * ihL * np.array([rsP[0], rsP[1], rsQ[0], rsQ[1]])
        return u, d

    def middle(self, i, jP, jQ):
        """m_i (complex scalar: the (0,0) entry) and M_TT(i) (4x4, None for the last block)."""
        L, P, Q = self.L, self.P, self.Q
        nL = self.nC[L]
        hP = [self.h[P][jP - 1], self.h[P][jP]]
        hQ = [self.h[Q][jQ - 1], self.h[Q][jQ]]
        kP = [0.5 / hP[0], 0.5 / hP[1]]
        kQ = [0.5 / hQ[0], 0.5 / hQ[1]]
        iL = min(i + 1, nL - 1)
        kL = [0.5 / self.h[L][i], 0.5 / self.h[L][iL]]
        ihL = [1.0 / self.h[L][i], 1.0 / self.h[L][iL]]
        z = [[[self.z(ii, jP - 1 + a, jQ - 1 + b) for b in range(2)] for a in range(2)] for ii in (i, iL)]
        et = lambda c, ii, a, b: self.eta[c][self.idx(ii, jP - 1 + a, jQ - 1 + b)]
        QP_Lm = [kP[S] * (z[0][S][1] + z[0][S][0]) for S in range(2)]
        PQ_Lm = [kQ[S] * (z[0][1][S] + z[0][0][S]) for S in range(2)]
        QL_Pm = [kL[S] * (z[S][0][1] + z[S][0][0]) for S in range(2)]
        LQ_Pm = [kQ[S] * (z[1][0][S] + z[0][0][S]) for S in range(2)]
        QL_Pp = [kL[S] * (z[S][1][1] + z[S][1][0]) for S in range(2)]
        LQ_Pp = [kQ[S] * (z[1][1][S] + z[0][1][S]) for S in range(2)]
        PL_Qm = [kL[S] * (z[S][1][0] + z[S][0][0]) for S in range(2)]
        LP_Qm = [kP[S] * (z[1][S][0] + z[0][S][0]) for S in range(2)]
        PL_Qp = [kL[S] * (z[S][1][1] + z[S][0][1]) for S in range(2)]
        LP_Qp = [kP[S] * (z[1][S][1] + z[0][S][1]) for S in range(2)]
        st0 = et(L, i, 1, 1) + et(L, i, 1, 0) + et(L, i, 0, 1) + et(L, i, 0, 0)
        m = -0.25 * st0 + (QP_Lm[1] / hP[1] + QP_Lm[0] / hP[0]) + (PQ_Lm[1] / hQ[1] + PQ_Lm[0] / hQ[0])
        if i == nL - 1:
            return m, None
        M = np.zeros((4, 4), dtype=complex)
        st = [et(P, iL, 0, 1) + et(P, iL, 0, 0) + et(P, i, 0, 1) + et(P, i, 0, 0),
              et(P, iL, 1, 1) + et(P, iL, 1, 0) + et(P, i, 1, 1) + et(P, i, 1, 0),
              et(Q, iL, 1, 0) + et(Q, iL, 0, 0) + et(Q, i, 1, 0) + et(Q, i, 0, 0),
              et(Q, iL, 1, 1) + et(Q, iL, 0, 1) + et(Q, i, 1, 1) + et(Q, i, 0, 1)]
        M[0, 0] = -0.25 * st[0] + (QL_Pm[1] * ihL[1] + QL_Pm[0] * ihL[0]) + (LQ_Pm[1] / hQ[1] + LQ_Pm[0] / hQ[0])
        M[1, 1] = -0.25 * st[1] + (QL_Pp[1] * ihL[1] + QL_Pp[0] * ihL[0]) + (LQ_Pp[1] / hQ[1] + LQ_Pp[0] / hQ[0])
        M[2, 2] = -0.25 * st[2] + (PL_Qm[1] * ihL[1] + PL_Qm[0] * ihL[0]) + (LP_Qm[1] / hP[1] + LP_Qm[0] / hP[0])
        M[3, 3] = -0.25 * st[3] + (PL_Qp[1] * ihL[1] + PL_Qp[0] * ihL[0]) + (LP_Qp[1] / hP[1] + LP_Qp[0] / hP[0])
        M[2, 0] = M[0, 2] = -LQ_Pm[0] / hP[0]
        M[3, 0] = M[0, 3] = LQ_Pm[1] / hP[0]
        M[2, 1] = M[1, 2] = LQ_Pp[0] / hP[1]
        M[3, 1] = M[1, 3] = -LQ_Pp[1] / hP[1]
        return m, M

    def rhs(self, i, jP, jQ):
        """b_l (L row of block i) and b_T (4 transverse rows, None for the last block): neighbour lines only."""
        L, P, Q = self.L, self.P, self.Q
        nL = self.nC[L]
        E, S = self.E, self.S
        ihP = [1.0 / self.h[P][jP - 1], 1.0 / self.h[P][jP]]
        ihQ = [1.0 / self.h[Q][jQ - 1], 1.0 / self.h[Q][jQ]]
        kP = [0.5 * ihP[0], 0.5 * ihP[1]]
        kQ = [0.5 * ihQ[0], 0.5 * ihQ[1]]
        zc = lambda ii, a, b: self.z(ii, jP - 1 + a, jQ - 1 + b)
        eL = lambda vL, vP, vQ: E[L][self.idx(vL, vP, vQ)]
        eP = lambda vL, vP, vQ: E[P][self.idx(vL, vP, vQ)]
        eQ = lambda vL, vP, vQ: E[Q][self.idx(vL, vP, vQ)]
        rsP = [zc(i, 0, 0) + zc(i, 0, 1), zc(i, 1, 0) + zc(i, 1, 1)]
        rsQ = [zc(i, 0, 0) + zc(i, 1, 0), zc(i, 0, 1) + zc(i, 1, 1)]
        bl = S[L][self.idx(i, jP, jQ)]
        bl = bl + kP[1] * ihP[1] * rsP[1] * eL(i, jP + 1, jQ) + kP[0] * ihP[0] * rsP[0] * eL(i, jP - 1, jQ)
        bl = bl + kQ[1] * ihQ[1] * rsQ[1] * eL(i, jP, jQ + 1) + kQ[0] * ihQ[0] * rsQ[0] * eL(i, jP, jQ - 1)
        if i == nL - 1:
            return bl, None
        iL = i + 1
        kL0, kL1 = 0.5 / self.h[L][i], 0.5 / self.h[L][iL]
        bT = np.zeros(4, dtype=complex)
        for k in range(4):
            typ, side = (1, k) if k < 2 else (2, k - 2)
            sg = -1.0 if side else 1.0
            if typ == 1:
                pc, pn = jP - 1 + side, (jP + 1 if side else jP - 1)
                zf = [zc(i, side, 0), zc(i, side, 1), zc(iL, side, 0), zc(iL, side, 1)]
                ihA = ihP[side]
                Ev = [eL(iL, pn, jQ), eL(i, pn, jQ), eQ(iL, pn, jQ), eQ(iL, pn, jQ - 1), eP(iL, pc, jQ + 1), eP(iL, pc, jQ - 1)]
                K = [sg * ihA, -sg * ihA, sg * kQ[1] * ihA, -sg * kQ[0] * ihA, kQ[1] * ihQ[1], kQ[0] * ihQ[0]]
                sv = S[P][self.idx(iL, pc, jQ)]
            else:
                qc, qn = jQ - 1 + side, (jQ + 1 if side else jQ - 1)
                zf = [zc(i, 0, side), zc(i, 1, side), zc(iL, 0, side), zc(iL, 1, side)]
                ihA = ihQ[side]
                Ev = [eL(iL, jP, qn), eL(i, jP, qn), eP(iL, jP, qn), eP(iL, jP - 1, qn), eQ(iL, jP + 1, qc), eQ(iL, jP - 1, qc)]
                K = [sg * ihA, -sg * ihA, sg * kP[1] * ihA, -sg * kP[0] * ihA, kP[1] * ihP[1], kP[0] * ihP[0]]
                sv = S[Q][self.idx(iL, jP, qc)]
            rs0, rs1 = zf[0] + zf[1], zf[2] + zf[3]
            cs0, cs1 = zf[0] + zf[2], zf[1] + zf[3]
            y = sv + (K[0] * kL1 * rs1) * Ev[0] + (K[1] * kL0 * rs0) * Ev[1]
            y = y + (K[2] * cs1) * Ev[2] + (K[3] * cs0) * Ev[3] + (K[4] * cs1) * Ev[4] + (K[5] * cs0) * Ev[5]
            bT[k] = y
        return bl, bT

    def write(self, i, jP, jQ, ell, T):
        L, P, Q = self.L, self.P, self.Q
        self.E[L][self.idx(i, jP, jQ)] = ell
        if T is not None:
            iL = i + 1
            self.E[P][self.idx(iL, jP - 1, jQ)] = T[0]
            self.E[P][self.idx(iL, jP, jQ)] = T[1]
            self.E[Q][self.idx(iL, jP, jQ - 1)] = T[2]
            self.E[Q][self.idx(iL, jP, jQ)] = T[3]


def factor_line(ls, jP, jQ, mid):
    """Two-sided factorisation of the reduced system: W_i (4x4) for T-blocks 0..nT-1, mu_i for i = 0..nL-1."""
    nL = ls.nC[ls.L]
    nT = nL - 1
    u, d, mu, MT = [], [], [], []
    for i in range(nL):
        ui, di = ls.coef(i, jP, jQ)
        m, M = ls.middle(i, jP, jQ)
        u.append(ui); d.append(di); mu.append(1.0 / m); MT.append(M)
    R = [mu[i] * np.outer(u[i], u[i]) for i in range(nL)]
    B = [np.diag(d[i]) + R[i] for i in range(nL)]          # B_i couples T_i and T_{i-1} (i >= 1)
    C = [MT[i] - R[i] - R[i + 1] for i in range(nT)]
    W = [None] * nT
    for i in range(0, mid):                                   # left chain
        S = C[i] - (B[i] @ W[i - 1] @ B[i] if i > 0 else 0)
        W[i] = np.linalg.inv(S)
    for i in range(nT - 1, mid, -1):                          # right chain
        S = C[i] - (B[i + 1] @ W[i + 1] @ B[i + 1] if i < nT - 1 else 0)
        W[i] = np.linalg.inv(S)
    S = C[mid].copy()
    if mid > 0:
        S -= B[mid] @ W[mid - 1] @ B[mid]
    if mid < nT - 1:
        S -= B[mid + 1] @ W[mid + 1] @ B[mid + 1]
    W[mid] = np.linalg.inv(S)
    return dict(u=u, d=d, mu=mu, W=W, B=B, mid=mid, nT=nT, nL=nL)


def solve_line(ls, fac, jP, jQ):
    nL, nT, mid = fac['nL'], fac['nT'], fac['mid']
    u, mu, W, B = fac['u'], fac['mu'], fac['W'], fac['B']
    bl, bT = [], []
    for i in range(nL):
        a, b = ls.rhs(i, jP, jQ)
        bl.append(a); bT.append(b)
    beta = [mu[i] * bl[i] for i in range(nL)]
    f = [bT[i] + u[i] * beta[i] - u[i + 1] * beta[i + 1] for i in range(nT)]
    z = [None] * nT
    for i in range(0, mid):
        z[i] = W[i] @ (f[i] - (B[i] @ z[i - 1] if i > 0 else 0))
    for i in range(nT - 1, mid, -1):
        z[i] = W[i] @ (f[i] - (B[i + 1] @ z[i + 1] if i < nT - 1 else 0))
    y = f[mid].copy()
    if mid > 0:
        y -= B[mid] @ z[mid - 1]
    if mid < nT - 1:
        y -= B[mid + 1] @ z[mid + 1]
    x = [None] * nT
    x[mid] = W[mid] @ y
    for i in range(mid - 1, -1, -1):
        x[i] = z[i] - W[i] @ (B[i + 1] @ x[i + 1])
    for i in range(mid + 1, nT):
        x[i] = z[i] - W[i] @ (B[i] @ x[i - 1])
    zero = np.zeros(4, dtype=complex)
    for i in range(nL):
        Ti = x[i] if i < nT else zero
        Tm = x[i - 1] if i > 0 else zero
        ell = beta[i] + mu[i] * (u[i] @ (Ti - Tm))
        ls.write(i, jP, jQ, ell, x[i] if i < nT else None)


def sweep_colour(vnC, e, s, eta, zeta, h, direction, nu=1):
    ls = LineSys(vnC, e, s, eta, zeta, h, direction)
    nP, nQ = vnC[ls.P], vnC[ls.Q]
    nT = vnC[ls.L] - 1
    mid = (nT - 1) // 2
    iback = 0
    for it in range(nu):
        iback = 1 - iback
        for ch in range(4):
            c = (0, 3, 2, 1)[ch] if iback else (1, 3, 0, 2)[ch]       # the colour schedule of the device path and the oracle
            cP, cQ = c & 1, c >> 1
            for jQ in range(1 + cQ, nQ, 2):
                for jP in range(1 + cP, nP, 2):
                    fac = factor_line(ls, jP, jQ, mid)
                    solve_line(ls, fac, jP, jQ)


if __name__ == "__main__":
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(3)
    for shape in [(7, 6, 5), (5, 8, 6), (6, 5, 9), (12, 4, 4)]:
        h = [rng.uniform(0.5, 2, n) for n in shape]
        eta = [np.asfortranarray(rng.uniform(0.5, 2, shape) * (0.3j + 0.05)) for _ in range(3)]
        zeta = np.asfortranarray(rng.uniform(0.5, 2, shape))
        nx, ny, nz = shape
        nE = nx * (ny + 1) * (nz + 1) + (nx + 1) * ny * (nz + 1) + (nx + 1) * (ny + 1) * nz
        s = rng.standard_normal(nE) + 1j * rng.standard_normal(nE)
        e0 = rng.standard_normal(nE) + 1j * rng.standard_normal(nE)
        for direction in (1, 2, 3):
            if shape[direction - 1] < 4:
                continue
            eo = e0.copy()
            oracle.gauss_seidel(shape, eo, s, *eta, zeta, *h, 2, direction=direction, order=1)
            ep = e0.copy()
            sweep_colour(shape, ep, s, eta, zeta, h, direction, nu=2)
            print(shape, direction, "rel.err", np.abs(ep - eo).max() / np.abs(eo).max())
