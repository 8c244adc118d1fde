"""CPU twin of the batched min-time solver (include/rl_mincurv.h: rl_mintime_solve_batch) --
TEST INFRASTRUCTURE ONLY (never imported by the product).

The NLP is the reference's (min_time_optm/min_time_optimizer.py:93-163 with models/double_track.py:10-204;
the functions themselves are pinned by fixture G8 through oracle/dt_checker.py).  The reference hands it
to IPOPT (casadi.Opti, :158-161), a third-party solver that is not in this image: what is restated here
is not IPOPT's code but the same KIND of method, written out in full -- a primal-dual interior-point
Newton iteration in which every iteration solves ONE equality-constrained QP subproblem (the barrier
QP) through its block-tridiagonal cyclic KKT system:

    variables per node j (9):  n, xi, omega, beta, v | F, delta, gamma | t      (scaled like the reference,
                               :109-113; the abscissa is pinned by :130 and u[1] only appears in the cost,
                               whose minimum puts it at 0 -- both are eliminated)
    equalities per node (7):   Hermite-Simpson defect (6) with node j+1, load-transfer residual
    inequalities per node(17): tyre ellipses (4), power, v >= 1, force lo/hi, steer lo/hi, force-rate lo/hi,
                               steer-rate lo/hi, lateral lo/hi, t >= 0          g(w) + s = 0, s > 0

    [ H + G' S^-1 Z G + delta I    A'     ] [dw]     [ r_d + G'(S^-1 Z r_g - S^-1 r_sz) ]
    [ A                       -eps I      ] [dy] = - [ r_c                               ]

H = Hessian of the Lagrangian (objective + exact constraint curvature: a Gauss-Newton variant without it
does not converge on this problem, measured), delta is a Levenberg / inertia-correction parameter: raised
until the block elimination shows 9 positive and 7 negative pivots per node, adapted by the step length
afterwards.  Step length: fraction to the boundary (0.995) for s and z, then backtracking on (infeasibility,
barrier objective): improve one of them against the current point, do not raise the infeasibility above
max(2 x current, 1e-5 per row), and -- where that floor is what lets it grow -- be acceptable to the last 8
iterates of the barrier problem as well (filter).  mu is lowered monotonically once the barrier problem is
solved to 30 mu.

The HIP kernels (k_mt_*) implement exactly this iteration; this file differs in how it does the work:
derivatives by complex-step differentiation of the numpy model, second derivatives by central differences
of those (the kernels use forward and forward-over-forward dual numbers), the linear solve by scipy's
sparse LU (the kernels use a block LDL' with a cyclic border); only the pivot-sign test that drives delta is
restated step for step (pivot_signs_ok).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import dt_checker as dc

NV, NE, NI = 9, 7, 17
THETA_GROWTH = 2.0
FILTER = 8
TH_FILTER = 1e-4
DUAL_CAP = 1.0
MU_KAPPA = 30.0                                   # the barrier problem counts as solved at MU_KAPPA * mu
D_DOWN, D_UP, A_HI, A_LO = 0.4, 3.0, 0.25, 0.1    # Levenberg parameter x D_DOWN after a step > A_HI, x D_UP after one <= A_LO (round 4: 0.9 / 0.2 before)
THETA_FLOOR = 1e-5     # ... above 1e-5 per row (next to a feasible point twice nothing is nothing)   # a step may not more than double the l1 infeasibility, whatever it does to the objective
# reduced variable a -> (array, column): X cols 1..5, U cols 0,2,3, T
_XCOL = [1, 2, 3, 4, 5]
_UCOL = [0, 2, 3]


class Problem:
    def __init__(self, model, s, kappa, left, right, track_length, average_track_width=7.0, speed_cap=30.0):
        self.m = dict(model)
        self.s = np.asarray(s, float); self.kappa = np.asarray(kappa, float)
        self.left = np.asarray(left, float); self.right = np.asarray(right, float)
        self.L = float(track_length)
        self.N = len(self.s)
        self.margin = self.m["vehicle_width"] / 2.0 + self.m["safety_margin"]
        m = self.m
        self.scale_x = np.array([1.0, average_track_width, 1.0, 1.0, 0.5, speed_cap])      # :109
        self.scale_u = np.array([m["Fd_max"], abs(m["Fb_max"]), m["delta_max"], m["mass"] * 50.0])  # :110-111
        self.sw = np.r_[self.scale_x[_XCOL], self.scale_u[_UCOL], 1.0]                        # scale of the 9 variables
        self.se = np.r_[1.0 / self.scale_x, 1.0 / self.scale_u[3]]                            # row scale of the 7 equalities
        assert np.all(self.right + self.margin < self.left - self.margin)

    # ---- packing
    def unpack(self, w):
        """w [N,9] scaled -> physical X [N,6], U [N,4], T [N]."""
        ph = w * self.sw
        X = np.zeros((self.N, 6), dtype=w.dtype); U = np.zeros((self.N, 4), dtype=w.dtype)
        X[:, 0] = self.s
        X[:, _XCOL] = ph[:, :5]
        U[:, _UCOL] = ph[:, 5:8]
        return X, U, ph[:, 8]

    def pack(self, X, U, T):
        w = np.zeros((self.N, NV))
        w[:, :5] = X[:, _XCOL]; w[:, 5:8] = U[:, _UCOL]; w[:, 8] = T
        return w / self.sw

    # ---- the pair functions, complex-safe, with the right node given separately
    def pair(self, X, U, T, Xn, Un):
        m, L = self.m, self.L
        Xn = Xn.copy()
        d = (Xn[:, 2] - X[:, 2]).real
        Xn[:, 2] = Xn[:, 2] + (np.arctan2(np.sin(d), np.cos(d)) - d)                # utils.py:10-13 (a shift by 2 pi k)
        ds = (X[:, 0] - Xn[:, 0]).real
        kk = np.abs(ds) + L / 2.0
        Xn[:, 0] = Xn[:, 0] + (kk - np.fmod(kk, L)) * np.sign(ds)                   # utils.py:15-18
        k = self.kappa
        f1, (Fx, Fy, Fz) = dc.dynamics(m, X, U, k)
        f2, _ = dc.dynamics(m, Xn, U, k)
        t = T[:, None]
        Xm = 0.5 * (X + Xn) + (t / 8.0) * (f1 - f2)
        fm, _ = dc.dynamics(m, Xm, U, k)
        eq = np.empty((self.N, NE), dtype=X.dtype)
        eq[:, :6] = X + (t / 6.0) * (f1 + 4 * fm + f2) - Xn
        delta, transfer, v = U[:, 2], U[:, 3], X[:, 5]
        eq[:, 6] = transfer - m["hcog"] / (0.5 * (m["twf"] + m["twr"])) * (
            Fy[:, 2] + Fy[:, 3] + (Fx[:, 0] + Fx[:, 1]) * np.sin(delta) + (Fy[:, 0] + Fy[:, 1]) * np.cos(delta))
        eq = eq * self.se
        g = np.empty((self.N, NI), dtype=X.dtype)
        g[:, :4] = (Fx / (m["mu"] * Fz)) ** 2 + (Fy / (m["mu"] * Fz)) ** 2 - 1.0
        drive = U[:, 0] * (np.tanh(U[:, 0]) * 0.5 + 0.5)
        su, sx = self.scale_u, self.scale_x
        g[:, 4] = (v * drive - m["Pmax"]) / m["Pmax"]
        g[:, 5] = (1.0 - v) / sx[5]
        g[:, 6] = (m["Fb_max"] - U[:, 0]) / su[0]; g[:, 7] = (U[:, 0] - m["Fd_max"]) / su[0]
        g[:, 8] = (-m["delta_max"] - delta) / su[2]; g[:, 9] = (delta - m["delta_max"]) / su[2]
        ru = (Un[:, 0] - U[:, 0]) / T; rd = (Un[:, 2] - delta) / T
        g[:, 10] = (m["Fb_max"] / m["Tb"] - ru) / su[0]; g[:, 11] = (ru - m["Fd_max"] / m["Td"]) / su[0]
        g[:, 12] = (-m["delta_max"] / m["Tdelta"] - rd) / su[2]; g[:, 13] = (rd - m["delta_max"] / m["Tdelta"]) / su[2]
        g[:, 14] = ((self.right + self.margin) - X[:, 1]) / sx[1]; g[:, 15] = (X[:, 1] - (self.left - self.margin)) / sx[1]
        g[:, 16] = -T
        return eq, g

    def functions(self, w):
        X, U, T = self.unpack(w)
        Xn, Un = np.roll(X, -1, axis=0), np.roll(U, -1, axis=0)
        return self.pair(X, U, T, Xn, Un)

    def cost(self, w):
        """sum T + 1e-4 sum |U|^2 + 1e-1 sum |dU|^2 in the scaled controls (:119-123), u[1] = 0."""
        us = w[:, 5:8]
        return w[:, 8].sum() + 1e-4 * (us ** 2).sum() + 1e-1 * ((np.roll(us, -1, axis=0) - us) ** 2).sum()

    def cost_grad_hess(self, w):
        us = w[:, 5:8]
        g = np.zeros_like(w)
        g[:, 8] = 1.0
        g[:, 5:8] = 2e-4 * us + 2e-1 * (2 * us - np.roll(us, -1, axis=0) - np.roll(us, 1, axis=0))
        N = self.N
        idx = (np.arange(N)[:, None] * NV + np.arange(5, 8)[None, :]).ravel()
        nxt = (((np.arange(N) + 1) % N)[:, None] * NV + np.arange(5, 8)[None, :]).ravel()
        H = sp.coo_matrix((np.full(len(idx), 2e-4 + 4e-1), (idx, idx)), shape=(N * NV, N * NV))
        H = H + sp.coo_matrix((np.full(len(idx), -2e-1), (idx, nxt)), shape=(N * NV, N * NV))
        H = H + sp.coo_matrix((np.full(len(idx), -2e-1), (nxt, idx)), shape=(N * NV, N * NV))
        return g, H.tocsr()

    def jacobians(self, w):
        """Complex-step derivatives of (eq, g) of every pair with respect to its own 9 variables and the 9 of
        the next node: Jo, Jn  [N, 7+17, 9]."""
        h = 1e-30
        X, U, T = self.unpack(w.astype(complex))
        Xn, Un = np.roll(X, -1, axis=0), np.roll(U, -1, axis=0)
        Jo = np.zeros((self.N, NE + NI, NV)); Jn = np.zeros((self.N, NE + NI, NV))
        for a in range(NV):
            step = 1j * h * self.sw[a]
            Xa, Ua, Ta = X.copy(), U.copy(), T.copy()
            if a < 5: Xa[:, _XCOL[a]] += step
            elif a < 8: Ua[:, _UCOL[a - 5]] += step
            else: Ta = Ta + step
            eq, g = self.pair(Xa, Ua, Ta, Xn, Un)
            Jo[:, :NE, a] = eq.imag / h; Jo[:, NE:, a] = g.imag / h
            if a < 8:
                Xb, Ub = Xn.copy(), Un.copy()
                if a < 5: Xb[:, _XCOL[a]] += step
                else: Ub[:, _UCOL[a - 5]] += step
                eq, g = self.pair(X, U, T, Xb, Ub)
                Jn[:, :NE, a] = eq.imag / h; Jn[:, NE:, a] = g.imag / h
        return Jo, Jn

    def sparse_jac(self, Jo, Jn, rows):
        """Blocks [N, r, 9] (own / next node) -> sparse [N r, N 9]."""
        N = self.N
        r = Jo.shape[1]
        ri = (np.arange(N)[:, None, None] * r + np.arange(r)[None, :, None] + np.zeros((1, 1, NV), int)).ravel()
        co = (np.arange(N)[:, None, None] * NV + np.zeros((1, r, 1), int) + np.arange(NV)[None, None, :]).ravel()
        cn = (((np.arange(N) + 1) % N)[:, None, None] * NV + np.zeros((1, r, 1), int) + np.arange(NV)[None, None, :]).ravel()
        return (sp.coo_matrix((Jo.ravel(), (ri, co)), shape=(N * r, N * NV)) +
                sp.coo_matrix((Jn.ravel(), (ri, cn)), shape=(N * r, N * NV))).tocsr()


def initial_point(P, speed, seg_time):
    """The reference's initial guess without a previous solution (min_time_optimizer.py:146-151): zero lateral
    offset / angles / rates, the QSS speed profile, controls (1, -1, 0.001, 0) in physical units, the QSS
    segment times -- made strictly interior where the NLP's own bounds require it (v >= 1, t > 0)."""
    N = P.N
    X = np.zeros((N, 6)); X[:, 0] = P.s; X[:, 5] = np.maximum(speed, 1.5)
    U = np.tile(np.array([1.0, 0.0, 0.001, 0.0]), (N, 1))
    T = np.maximum(seg_time, 1e-3)
    return P.pack(X, U, T)


def lagrangian_hessian(P, w, y, z):
    """Hessian of sum_j (y_j . eq_j + z_j . g_j) with respect to w: per pair an 18 x 18 block (own 9 + next 9
    variables), by central differences of the complex-step gradient.  Sparse [9N, 9N]."""
    N = P.N
    mult = np.concatenate([y, z], axis=1)                      # [N, 24]
    X, U, T = P.unpack(w.astype(complex))
    Xn, Un = np.roll(X, -1, axis=0), np.roll(U, -1, axis=0)
    Tn = np.roll(T, -1)

    def grad(X, U, T, Xn, Un):
        """gradient of the pair Lagrangian wrt the 18 local variables [N,18] (next-node t does not enter)."""
        h = 1e-30
        out = np.zeros((N, 18))
        for a in range(NV):
            step = 1j * h * P.sw[a]
            Xa, Ua, Ta = X.copy(), U.copy(), T.copy()
            if a < 5: Xa[:, _XCOL[a]] += step
            elif a < 8: Ua[:, _UCOL[a - 5]] += step
            else: Ta = Ta + step
            eq, g = P.pair(Xa, Ua, Ta, Xn, Un)
            out[:, a] = (np.concatenate([eq, g], axis=1).imag / h * mult).sum(axis=1)
            if a < 8:
                Xb, Ub = Xn.copy(), Un.copy()
                if a < 5: Xb[:, _XCOL[a]] += step
                else: Ub[:, _UCOL[a - 5]] += step
                eq, g = P.pair(X, U, T, Xb, Ub)
                out[:, 9 + a] = (np.concatenate([eq, g], axis=1).imag / h * mult).sum(axis=1)
        return out
    Hb = np.zeros((N, 18, 18))
    e = 1e-6
    for b in range(17):
        a = b if b < 9 else b - 9
        d = e * P.sw[a]
        outs = []
        for sgn in (+1, -1):
            Xa, Ua, Ta, Xb, Ub = X.copy(), U.copy(), T.copy(), Xn.copy(), Un.copy()
            tgtX, tgtU = (Xa, Ua) if b < 9 else (Xb, Ub)
            if a < 5: tgtX[:, _XCOL[a]] += sgn * d
            elif a < 8: tgtU[:, _UCOL[a - 5]] += sgn * d
            else: Ta = Ta + sgn * d
            outs.append(grad(Xa, Ua, Ta, Xb, Ub))
        Hb[:, :, b] = (outs[0] - outs[1]) / (2 * e)
    Hb = 0.5 * (Hb + Hb.transpose(0, 2, 1))
    own = np.arange(N)[:, None] * NV + np.arange(NV)[None, :]
    nxt = ((np.arange(N) + 1) % N)[:, None] * NV + np.arange(NV)[None, :]
    idx = np.concatenate([own, nxt], axis=1)                   # [N,18]
    ri = np.repeat(idx[:, :, None], 18, axis=2).ravel(); ci = np.repeat(idx[:, None, :], 18, axis=1).ravel()
    return sp.coo_matrix((Hb.ravel(), (ri, ci)), shape=(N * NV, N * NV)).tocsr()


def pivot_signs_ok(K_ww, A, N, eps_reg):
    """The inertia test of the HIP kernel, restated: eliminate the cyclic block-tridiagonal KKT matrix node by
    node (16 x 16 blocks: 9 unknowns then 7 multipliers, last node as the border) WITHOUT pivoting and
    require every pivot to be non-negligible and exactly 7 N of them negative: the inertia (9N, 7N, 0) of a
    KKT matrix whose reduced Hessian is positive definite (Sylvester: LDL' pivot signs = eigenvalue signs)."""
    nb = NV + NE
    Kd = K_ww.toarray() if N <= 64 else None

    def blockof(i, j):
        Mb = np.zeros((nb, nb))
        ri = slice(i * NV, (i + 1) * NV); cj = slice(j * NV, (j + 1) * NV)
        Mb[:NV, :NV] = K_ww[ri, cj].toarray() if Kd is None else Kd[ri, cj]
        Mb[:NV, NV:] = A[j * NE:(j + 1) * NE, ri].toarray().T          # A' : rows of node i's unknowns, pair j's multipliers
        Mb[NV:, :NV] = A[i * NE:(i + 1) * NE, cj].toarray()
        if i == j:
            Mb[NV:, NV:] = -eps_reg * np.eye(NE)
        return Mb

    count = [0]

    def pivots_ok(S):
        S = S.copy()
        for k in range(nb):
            p = S[k, k]
            if not np.isfinite(p) or abs(p) < 1e-11:
                return False
            count[0] += p < 0
            S[k + 1:, k + 1:] -= np.outer(S[k + 1:, k], S[k, k + 1:]) / p
        return True
    last = N - 1
    S_last = blockof(last, last)
    F = blockof(last, 0)
    S = blockof(0, 0)
    for j in range(N - 1):
        E = blockof(j + 1, j)
        if j == N - 2:
            F = F + E
        if not pivots_ok(S):
            return False
        Sinv = np.linalg.inv(S)
        Q = F @ Sinv
        S_last = S_last - Q @ F.T
        if j < N - 2:
            Pm = E @ Sinv
            S = blockof(j + 1, j + 1) - Pm @ E.T
            F = -Q @ E.T
    return pivots_ok(S_last) and count[0] == N * NE


def solve(P, w0, max_iter=60, tol=1e-6, eps_reg=1e-8, delta0=1e-4, mu0=1e-1, verbose=False):
    """Returns (w, info).  Every pass of the loop is one iteration of the HIP solver (k_mt_derivs, k_mt_kkt,
    k_mt_step), including the passes that only raise delta.  info: iterations, kkt (scaled dual
    infeasibility), viol (max scaled constraint violation), compl, lap_time, status (1 converged, 0 iteration
    limit, 2 failed), history."""
    N = P.N
    w = w0.copy()
    eq, g = P.functions(w)
    s = np.maximum(-g, 1e-2)                       # slacks start interior
    mu = mu0
    z = mu / s
    y = np.zeros((N, NE))
    delta = delta0
    hist = []
    _, H = P.cost_grad_hess(w)
    I9 = sp.identity(N * NV, format="csr")
    status = 0
    filt = []
    kkt = viol = compl = np.inf
    for it in range(max_iter):
        eq, g = P.functions(w)
        Jo, Jn = P.jacobians(w)
        A = P.sparse_jac(Jo[:, :NE], Jn[:, :NE], NE)
        G = P.sparse_jac(Jo[:, NE:], Jn[:, NE:], NI)
        gc, _ = P.cost_grad_hess(w)
        sz = s.ravel(); zz = z.ravel()
        r_d = gc.ravel() + A.T @ y.ravel() + G.T @ zz
        r_c = eq.ravel()
        r_g = g.ravel() + sz
        kkt = np.abs(r_d).max(); viol = max(np.abs(r_c).max(), np.abs(r_g).max()); compl = (sz * zz).max()
        hist.append([kkt, viol, compl, mu, delta, float((w[:, 8] * P.sw[8]).sum()), 0.0])
        if verbose:
            print(f"it {it:3d} kkt {kkt:.2e} viol {viol:.2e} compl {compl:.2e} mu {mu:.1e} delta {delta:.1e} "
                  f"lap {hist[-1][5]:.4f} alpha {hist[-2][6] if len(hist) > 1 else 0:.3f}")
        if max(kkt, viol, compl) <= tol:
            status = 1
            break
        # barrier parameter: monotone (Fiacco-McCormick): lowered once the barrier problem is solved to ~MU_KAPPA mu
        if max(kkt, viol, np.abs(sz * zz - mu).max()) <= MU_KAPPA * mu:
            mu_new = max(min(0.2 * mu, mu ** 1.5), tol / 10.0)     # no lower than needed for compl <= tol
            if mu_new != mu:
                filt = []      # a new barrier problem: another objective, the filter starts empty
            mu = mu_new
        Hk = H + lagrangian_hessian(P, w, y, z)
        W = sp.diags(zz / sz)
        Kww0 = (Hk + G.T @ W @ G).tocsr()
        ok_f = False
        for _ in range(12):
            if pivot_signs_ok((Kww0 + delta * I9).tocsr(), A, N, eps_reg):
                ok_f = True
                break
            delta = max(10.0 * delta, 1e-4)
            if delta > 1e8:
                break
        if not ok_f:
            status = 2
            break
        K = sp.bmat([[Kww0 + delta * I9, A.T], [A, -eps_reg * sp.identity(N * NE)]], format="csc")
        rhs = -np.r_[gc.ravel() + A.T @ y.ravel() + G.T @ (mu / sz + zz / sz * r_g), r_c]
        sol = spla.spsolve(K, rhs)
        dw = sol[:N * NV].reshape(N, NV); dy = sol[N * NV:].reshape(N, NE)
        ds = -r_g - G @ dw.ravel()
        dz = -(sz * zz - mu + zz * ds) / sz
        ap = min(1.0, _ftb(sz, ds)); ad = min(1.0, _ftb(zz, dz))
        # backtracking against the one-entry filter (theta0, phi0): accept when the infeasibility or the barrier
        # objective improves
        th0 = np.abs(r_c).sum() + np.abs(r_g).sum()
        ph0 = P.cost(w) - mu * np.log(sz).sum()
        a = ap
        ok = False
        for _ in range(12):
            wt = w + a * dw; stt = sz + a * ds
            eqt, gt = P.functions(wt)
            tht = np.abs(eqt).sum() + np.abs(gt.ravel() + stt).sum()
            pht = P.cost(wt) - mu * np.log(stt).sum()
            floor = THETA_FLOOR * N * (NE + NI)
            eth, eph = 1e-13 * N * (NE + NI), 1e-13 * abs(ph0)     # what rounding alone moves the two sums by
            acc = np.isfinite(pht) and np.isfinite(tht) and tht <= max(THETA_GROWTH * th0, floor) + 1e-9 and \
                (tht <= (1 - 1e-5) * th0 + eth or pht <= ph0 - 1e-5 * th0 + eph)
            # ... and, once nearly feasible (l1 infeasibility below TH_FILTER per row), acceptable to the (up to FILTER)
            # earlier iterates of this barrier problem: no cycling between two nearly feasible points
            if th0 < TH_FILTER * N * (NE + NI):
                for (thf, phf) in filt[-FILTER:]:
                    acc = acc and (tht <= (1 - 1e-5) * thf + eth or pht <= phf - 1e-5 * thf + eph)
            if acc:
                ok = True
                break
            a *= 0.5
        if not ok:
            delta = max(delta * 10.0, 1e-4)
            if delta > 1e6:
                status = 2
                break
            continue
        hist[-1][6] = a
        filt.append((th0, ph0))     # the point just left joins the filter
        w = wt; s = stt.reshape(N, NI)
        y = y + a * dy
        z = (zz + min(ad, 1.0, max(DUAL_CAP * a, 1e-3)) * dz).reshape(N, NI)   # the dual step does not outrun the primal one
        z = np.clip(z, mu / (1e10 * s), 1e10 * mu / s)
        # Levenberg parameter: a short step means the linearisation was not trusted that far
        delta = min(max(delta * (D_DOWN if a > A_HI else (1.0 if a > A_LO else D_UP)), 1e-6), 1e3)
    X, U, T = P.unpack(w)
    info = {"iterations": len(hist), "kkt": kkt, "viol": viol, "compl": compl, "lap_time": float(T.sum()), "history": hist,
            "status": status, "y": y, "z": z, "s": s, "mu": mu, "delta": delta}
    return w, info


def _ftb(v, dv, tau=0.995):
    neg = dv < 0
    if not neg.any():
        return 1.0
    return float(np.min(-tau * v[neg] / dv[neg]))
