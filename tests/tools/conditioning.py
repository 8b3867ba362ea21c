"""How well conditioned are the line systems of the BASELINE workloads, and what does that mean for
rounding-level parity between two elimination orders?  (development / evidence tool; CPU only)

For sample lines of the 128^3 (or 256^3) benchmark model: the single-line update is computed
  ref64 : by the oracle (= the reference's one-sided band LDL^T, float64), on the 2x2-cell sub-grid around the line
  two64 : by the two-sided reduced 4x4 block elimination of tests/tools/reduced_line.py (float64)
  truth : by the same two-sided elimination in 80-bit long double (entries taken from their float64 values)
and the relative max-norm differences are printed."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import emg3d_amd as em
import bench
from oracle import oracle
from reduced_line import LineSys, axes

LD = np.longdouble
CLD = np.clongdouble


def inv_ld(S):
    """Gauss-Jordan with partial pivoting in long double."""
    n = S.shape[0]
    A = np.concatenate([S.astype(CLD), np.eye(n, dtype=CLD)], axis=1)
    for c in range(n):
        p = c + int(np.argmax(np.abs(A[c:, c])))
        A[[c, p]] = A[[p, c]]
        A[c] = A[c] / A[c, c]
        for r in range(n):
            if r != c:
                A[r] = A[r] - A[r, c] * A[c]
    return A[:, n:]


def solve_two_sided(ls, jP, jQ, dtype, inv, one_sided=False):
    nL = ls.nC[ls.L]; nT = nL - 1; mid = (nT - 1) if one_sided else (nT - 1) // 2
    u, d, mu, MT, bl, bT = [], [], [], [], [], []
    for i in range(nL):
        ui, di = ls.coef(i, jP, jQ); m, M = ls.middle(i, jP, jQ); a, b = ls.rhs(i, jP, jQ)
        u.append(ui.astype(dtype)); d.append(di.astype(dtype)); mu.append(dtype(1) / dtype(m))
        MT.append(None if M is None else M.astype(dtype)); bl.append(dtype(a)); bT.append(None if b is None else b.astype(dtype))
    R = [mu[i] * np.outer(u[i], u[i]) for i in range(nL)]
    B = [np.diag(d[i]) + R[i] for i in range(nL)]
    C = [MT[i] - R[i] - R[i + 1] for i in range(nT)]
    beta = [mu[i] * bl[i] for i in range(nL)]
    f = [bT[i] + u[i] * beta[i] - u[i + 1] * beta[i + 1] for i in range(nT)]
    W = [None] * nT; z = [None] * nT
    for i in range(0, mid):
        S = C[i] - (B[i] @ W[i - 1] @ B[i] if i > 0 else 0); W[i] = inv(S)
        z[i] = W[i] @ (f[i] - (B[i] @ z[i - 1] if i > 0 else 0))
    for i in range(nT - 1, mid, -1):
        S = C[i] - (B[i + 1] @ W[i + 1] @ B[i + 1] if i < nT - 1 else 0); W[i] = inv(S)
        z[i] = W[i] @ (f[i] - (B[i + 1] @ z[i + 1] if i < nT - 1 else 0))
    S = C[mid] - B[mid] @ W[mid - 1] @ B[mid] - (B[mid + 1] @ W[mid + 1] @ B[mid + 1] if mid < nT - 1 else 0)
    W[mid] = inv(S)
    x = [None] * nT
    x[mid] = W[mid] @ (f[mid] - B[mid] @ z[mid - 1] - (B[mid + 1] @ z[mid + 1] if mid < nT - 1 else 0))
    for i in range(mid - 1, -1, -1):
        x[i] = z[i] - W[i] @ (B[i + 1] @ x[i + 1])
    for i in range(mid + 1, nT):
        x[i] = z[i] - W[i] @ (B[i] @ x[i - 1])
    zero = np.zeros(4, dtype=dtype)
    out = []
    for i in range(nL):
        Ti = x[i] if i < nT else zero; Tm = x[i - 1] if i > 0 else zero
        out.append(beta[i] + mu[i] * (u[i] @ (Ti - Tm)))
        if i < nT:
            out.extend(x[i])
    return np.array(out)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "128F"
    nlines = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    # optional explicit lines "jP,jQ;jP,jQ;..." (same pairs for every direction)
    fixed = [tuple(int(v) for v in t.split(',')) for t in sys.argv[3].split(';')] if len(sys.argv) > 3 else None
    if fixed:
        nlines = len(fixed)
    grid, model, sfield, cycle = bench.build_problem(em, wl, 1.0)
    vm = em.VolumeModel(grid, model, sfield)
    from test_gpu_fullsize import _smooth_field
    e0 = np.array(_smooth_field(grid, 7)); s = np.array(_smooth_field(grid, 8)) * 1e-3
    eta = [np.asfortranarray(a) for a in (vm.eta_x, vm.eta_y, vm.eta_z)]
    zeta = np.asfortranarray(vm.zeta)
    rng = np.random.default_rng(0)
    oracle.build()
    for direction in (1, 2, 3):
        L, P, Q = axes(direction)
        ls = LineSys(tuple(grid.vnC), e0.copy(), s, eta, zeta, grid.h, direction)
        worst = []
        for k in range(nlines):
            jP = int(rng.integers(1, grid.vnC[P])); jQ = int(rng.integers(1, grid.vnC[Q]))
            if fixed:
                jP, jQ = fixed[k]
            # the reference's own single-line update on the sub-grid spanning nodes jP-1..jP+1, jQ-1..jQ+1
            sl = [slice(None)] * 3
            def sub(arr, comp, cell=False):
                ix = [None, None, None]
                for ax, (j, name) in ((P, (jP, 'P')), (Q, (jQ, 'Q'))):
                    if cell or comp == ax:
                        ix[ax] = slice(j - 1, j + 1)
                    else:
                        ix[ax] = slice(j - 1, j + 2)
                ix[L] = slice(None)
                return np.asfortranarray(arr[tuple(ix)])
            sub_e = [sub(ls.E[c], c) for c in range(3)]
            sub_s = [sub(ls.S[c], c) for c in range(3)]
            sub_eta = [sub(eta[c], None, True) for c in range(3)]
            sub_zeta = sub(zeta, None, True)
            hh = [None] * 3
            hh[L] = grid.h[L]; hh[P] = grid.h[P][jP - 1:jP + 1]; hh[Q] = grid.h[Q][jQ - 1:jQ + 1]
            shape = sub_zeta.shape
            eb = np.concatenate([a.ravel(order='F') for a in sub_e]); sb = np.concatenate([a.ravel(order='F') for a in sub_s])
            oracle.gauss_seidel(shape, eb, sb, *sub_eta, sub_zeta, *hh, 1, direction=direction, order=0)
            lsub = LineSys(shape, eb, sb, sub_eta, sub_zeta, hh, direction)
            nL = shape[L]
            ref = []
            for i in range(nL):
                ref.append(lsub.E[L][lsub.idx(i, 1, 1)])
                if i < nL - 1:
                    ref += [lsub.E[P][lsub.idx(i + 1, 0, 1)], lsub.E[P][lsub.idx(i + 1, 1, 1)],
                            lsub.E[Q][lsub.idx(i + 1, 1, 0)], lsub.E[Q][lsub.idx(i + 1, 1, 1)]]
            ref = np.array(ref)
            two = solve_two_sided(ls, jP, jQ, np.complex128, np.linalg.inv)
            one = solve_two_sided(ls, jP, jQ, np.complex128, np.linalg.inv, one_sided=True)
            tru = solve_two_sided(ls, jP, jQ, CLD, inv_ld)
            sc = float(np.abs(tru).max())
            worst.append((float(np.abs(ref - tru).max()) / sc, float(np.abs(two - tru).max()) / sc,
                          float(np.abs(ref - two.astype(np.complex128)).max()) / sc, jP, jQ,
                          float(np.abs(one - tru).max()) / sc, float(np.abs(one - ref).max()) / sc))
        w = np.array([x[:3] for x in worst])
        if fixed:
            for x in worst:
                print(f"      line ({x[3]},{x[4]}): ref {x[0]:.1e} two {x[1]:.1e} ref-two {x[2]:.1e}   one-sided reduced: vs truth {x[5]:.1e} vs ref {x[6]:.1e}")
        print(f"{wl} direction {direction}: {nlines} lines; rel. max-norm error of ONE line solve")
        print(f"   reference order (oracle, float64) vs long-double truth : median {np.median(w[:,0]):.1e}  max {w[:,0].max():.1e}")
        print(f"   two-sided reduced (float64)       vs long-double truth : median {np.median(w[:,1]):.1e}  max {w[:,1].max():.1e}")
        print(f"   reference order vs two-sided (both float64)            : median {np.median(w[:,2]):.1e}  max {w[:,2].max():.1e}", flush=True)


if __name__ == "__main__":
    main()


def solve_block5(ls, jP, jQ, dtype, inv, one_sided):
    """The 5x5 block formulation of k_line_factor / k_line_sweep_* (explicit block inverses), one- or two-sided."""
    nL = ls.nC[ls.L]; mid = nL - 1 if one_sided else (nL - 1) // 2
    M, A, b = [], [], []
    for i in range(nL):
        ui, di = ls.coef(i, jP, jQ); m, MT = ls.middle(i, jP, jQ); bl, bT = ls.rhs(i, jP, jQ)
        Mi = np.zeros((5, 5), dtype=dtype); Ai = np.zeros((5, 5), dtype=dtype); bi = np.zeros(5, dtype=dtype)
        Mi[0, 0] = m; bi[0] = bl
        Ai[0, 1:] = ui
        if MT is not None:
            Mi[1:, 1:] = MT; Mi[1:, 0] = -ui; Mi[0, 1:] = -ui; bi[1:] = bT
            Ai[1:, 1:] = np.diag(di)
        else:
            Mi[1:, 1:] = np.eye(4)          # dummy unknowns of the last block (decoupled)
        M.append(Mi); A.append(Ai); b.append(bi)
    W = [None] * nL; z = [None] * nL
    for i in range(0, mid):
        S = M[i] - (A[i] @ W[i - 1] @ A[i].T if i > 0 else 0); W[i] = inv(S)
        z[i] = W[i] @ (b[i] - (A[i] @ z[i - 1] if i > 0 else 0))
    for i in range(nL - 1, mid, -1):
        S = M[i] - (A[i + 1].T @ W[i + 1] @ A[i + 1] if i < nL - 1 else 0); W[i] = inv(S)
        z[i] = W[i] @ (b[i] - (A[i + 1].T @ z[i + 1] if i < nL - 1 else 0))
    S = M[mid] - A[mid] @ W[mid - 1] @ A[mid].T - (A[mid + 1].T @ W[mid + 1] @ A[mid + 1] if mid < nL - 1 else 0)
    W[mid] = inv(S)
    x = [None] * nL
    x[mid] = W[mid] @ (b[mid] - A[mid] @ z[mid - 1] - (A[mid + 1].T @ z[mid + 1] if mid < nL - 1 else 0))
    for i in range(mid - 1, -1, -1):
        x[i] = z[i] - W[i] @ (A[i + 1].T @ x[i + 1])
    for i in range(mid + 1, nL):
        x[i] = z[i] - W[i] @ (A[i] @ x[i - 1])
    out = []
    for i in range(nL):
        out.append(x[i][0])
        if i < nL - 1:
            out.extend(x[i][1:])
    return np.array(out)


def solve_mirrored(ls, jP, jQ, dtype, inv):
    """Two-sided elimination with MIRRORED right-half blocks: left blocks [l_i; T_i] (i < m) eliminated upwards,
    right blocks [l_j; T_{j-1}] (j > m+1) eliminated downwards, middle block [l_m; T_m; l_{m+1}] (6 unknowns) last."""
    n = ls.nC[ls.L]
    u, d, mm, MT, bl, bT = [], [], [], [], [], []
    for i in range(n):
        ui, di = ls.coef(i, jP, jQ); m_, M = ls.middle(i, jP, jQ); a, b = ls.rhs(i, jP, jQ)
        u.append(ui.astype(dtype)); d.append(di.astype(dtype)); mm.append(dtype(m_))
        MT.append(None if M is None else M.astype(dtype)); bl.append(dtype(a)); bT.append(None if b is None else b.astype(dtype))
    m = (n - 2) // 2
    nleft, nright = m, n - m - 2
    def blockL(i):      # [l_i; T_i]
        B = np.zeros((5, 5), dtype=dtype); B[0, 0] = mm[i]; B[1:, 1:] = MT[i]; B[0, 1:] = -u[i]; B[1:, 0] = -u[i]
        return B, np.concatenate([[bl[i]], bT[i]])
    def blockR(j):      # [l_j; T_{j-1}]
        B = np.zeros((5, 5), dtype=dtype); B[0, 0] = mm[j]; B[1:, 1:] = MT[j - 1]; B[0, 1:] = u[j]; B[1:, 0] = u[j]
        return B, np.concatenate([[bl[j]], bT[j - 1]])
    WL, zL = {}, {}
    for i in range(0, m):
        B, b = blockL(i)
        if i > 0:
            A = np.zeros((5, 5), dtype=dtype); A[0, 1:] = u[i]; A[1:, 1:] = np.diag(d[i])
            B = B - A @ WL[i - 1] @ A.T; b = b - A @ zL[i - 1]
        WL[i] = inv(B); zL[i] = WL[i] @ b
    WR, zR = {}, {}
    for j in range(n - 1, m + 1, -1):
        B, b = blockR(j)
        if j < n - 1:
            C = np.zeros((5, 5), dtype=dtype); C[1:, 0] = -u[j]; C[1:, 1:] = np.diag(d[j])      # rows block j+1, cols block j
            B = B - C.T @ WR[j + 1] @ C; b = b - C.T @ zR[j + 1]
        WR[j] = inv(B); zR[j] = WR[j] @ b
    # middle [l_m; T_m; l_{m+1}]
    S = np.zeros((6, 6), dtype=dtype)
    S[0, 0] = mm[m]; S[1:5, 1:5] = MT[m]; S[5, 5] = mm[m + 1]
    S[0, 1:5] = -u[m]; S[1:5, 0] = -u[m]; S[5, 1:5] = u[m + 1]; S[1:5, 5] = u[m + 1]
    y = np.concatenate([[bl[m]], bT[m], [bl[m + 1]]])
    if m > 0:
        A = np.zeros((6, 5), dtype=dtype); A[0, 1:] = u[m]; A[1:5, 1:] = np.diag(d[m])
        S = S - A @ WL[m - 1] @ A.T; y = y - A @ zL[m - 1]
    if m + 2 <= n - 1:
        C = np.zeros((5, 6), dtype=dtype); C[1:, 5] = -u[m + 1]; C[1:, 1:5] = np.diag(d[m + 1])   # rows block m+2, cols middle
        S = S - C.T @ WR[m + 2] @ C; y = y - C.T @ zR[m + 2]
    xm = inv(S) @ y
    ell = [None] * n; T = [None] * (n - 1)
    ell[m], T[m], ell[m + 1] = xm[0], xm[1:5], xm[5]
    xin = np.concatenate([[ell[m]], T[m]])
    for i in range(m - 1, -1, -1):
        A = np.zeros((5, 5), dtype=dtype); A[0, 1:] = u[i + 1]; A[1:, 1:] = np.diag(d[i + 1])
        x = zL[i] - WL[i] @ (A.T @ xin)
        ell[i], T[i] = x[0], x[1:]; xin = x
    xin = np.concatenate([[ell[m + 1]], T[m]])
    for j in range(m + 2, n):
        C = np.zeros((5, 5), dtype=dtype); C[1:, 0] = -u[j - 1]; C[1:, 1:] = np.diag(d[j - 1])
        x = zR[j] - WR[j] @ (C @ xin)
        ell[j], T[j - 1] = x[0], x[1:]; xin = x
    out = []
    for i in range(n):
        out.append(ell[i])
        if i < n - 1:
            out.extend(T[i])
    return np.array(out)
