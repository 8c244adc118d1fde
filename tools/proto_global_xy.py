"""Prototype (numpy, dense) of the two-unknowns-per-control-point global QP: formulation questions only."""
import sys, os
LS = None
import numpy as np
from scipy.interpolate import BSpline
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden, spline  # noqa
from test_global_qp import monza_widths  # noqa


def design(t, k, u, nu):
    n = len(t) - k - 1
    out = np.zeros((len(u), n))
    for j in range(n):
        c = np.zeros(n); c[j] = 1.0
        out[:, j] = BSpline(t, c, k)(u, nu)
    return out


def ipm(P, q, A, lo, hi, x, warm=None, tol_mu=1e-10, tol_res=1e-9):
    m = len(lo)
    ax = A @ x
    if warm is None:
        sl, su = np.maximum(ax - lo, 1e-2), np.maximum(hi - ax, 1e-2); ll = np.ones(m); lu = np.ones(m)
    else:
        sl, su, ll, lu = [np.maximum(v, 1e-2) for v in warm]
    qinf = np.abs(q).max()
    its = 0
    for it in range(80):
        ax = A @ x
        rpl, rpu = ax - lo - sl, hi - ax - su
        rd = P @ x + q + A.T @ (lu - ll)
        mu = (sl @ ll + su @ lu) / (2 * m)
        if max(np.abs(rd).max() / (1 + qinf), np.abs(rpl).max(), np.abs(rpu).max()) < tol_res and mu < tol_mu:
            break
        its += 1
        dm = ll / sl + lu / su
        K = P + A.T @ (dm[:, None] * A)
        L = np.linalg.cholesky(K)
        sigma = 0.0; dsl = dsu = dll = dlu = 0.0
        for ps in range(2):
            rcl, rcu = sl * ll, su * lu
            if ps == 1:
                rcl = rcl - sigma * mu + dsl * dll; rcu = rcu - sigma * mu + dsu * dlu
            wv = (-rcl - ll * rpl) / sl - (-rcu - lu * rpu) / su
            rhs = -rd + A.T @ wv
            dx = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            adx = A @ dx
            dsl, dsu = adx + rpl, -adx + rpu
            dll, dlu = (-rcl - ll * dsl) / sl, (-rcu - lu * dsu) / su
            amin = 1.0
            for v, dv in ((sl, dsl), (su, dsu), (ll, dll), (lu, dlu)):
                neg = dv < 0
                if neg.any(): amin = min(amin, 0.995 * (-v[neg] / dv[neg]).min())
            if ps == 0:
                mu_aff = ((sl + amin * dsl) @ (ll + amin * dll) + (su + amin * dsu) @ (lu + amin * dlu)) / (2 * m)
                sigma = (mu_aff / mu) ** 3
        x = x + amin * dx
        sl, su, ll, lu = sl + amin * dsl, su + amin * dsu, ll + amin * dll, lu + amin * dlu
    return x, (sl, su, ll, lu), its, np.linalg.cond(K)


def main(N=2000, n_outer=6, lon=1.0, eps=1e-9, dof=2, rho=0.0):
    fits = golden("G1_spline_fits.npz")
    t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", N)
    n = len(cx); npp = n - k
    B0, B1, B2 = design(t, k, u, 0), design(t, k, u, 1), design(t, k, u, 2)
    fold = lambda M: M[:, :npp] + np.pad(M[:, npp:], ((0, 0), (0, npp - k)))   # periodic: columns np.. alias 0..k-1
    B0, B1, B2 = fold(B0), fold(B1), fold(B2)
    c0x, c0y = cx[:npp], cy[:npp]
    dx, dy = B1 @ c0x, B1 @ c0y
    s = np.hypot(dx, dy)
    tx, ty = dx / s, dy / s
    nx, ny = -ty, tx
    if dof == 2:
        A = np.zeros((2 * N, 2 * npp))
        A[:N, 0::2] = B0 * nx[:, None]; A[:N, 1::2] = B0 * ny[:, None]
        A[N:, 0::2] = B0 * tx[:, None]; A[N:, 1::2] = B0 * ty[:, None]
        lo = np.concatenate([-(wr - 0.25), -lon * np.ones(N)]); hi = np.concatenate([wl - 0.25, lon * np.ones(N)])
    z = np.zeros(2 * npp)
    warm = None
    tot = 0
    for outer in range(n_outer + 1):
        X, Y = c0x + z[0::2], c0y + z[1::2]
        dx, dy, ddx, ddy = B1 @ X, B1 @ Y, B2 @ X, B2 @ Y
        s2 = dx * dx + dy * dy
        inv3 = 1.0 / (s2 * np.sqrt(s2))
        kap = (dx * ddy - dy * ddx) * inv3
        print(f"outer {outer}: sum kappa^2 = {kap @ kap:.9e}")
        if outer == n_outer: break
        # d kappa / d(dx, dy, ddx, ddy)
        g_dx = ddy * inv3 - 3 * kap * dx / s2
        g_dy = -ddx * inv3 - 3 * kap * dy / s2
        g_ddx = -dy * inv3
        g_ddy = dx * inv3
        G = np.zeros((N, 2 * npp))
        G[:, 0::2] = g_dx[:, None] * B1 + g_ddx[:, None] * B2
        G[:, 1::2] = g_dy[:, None] * B1 + g_ddy[:, None] * B2
        P = 2 * G.T @ G
        q = 2 * G.T @ (kap - G @ z)
        sc = len(z) / np.trace(P)
        P = P * sc + (eps + rho) * np.eye(len(z)); q = q * sc - rho * z
        ev = np.linalg.eigvalsh(P)
        last = outer + 1 >= n_outer
        znew, warm, its, cond = ipm(P, q, A, lo, hi, z.copy(), warm, 1e-10 if last else 1e-5, 1e-9 if last else 1e-4)
        tot += its
        lat = A[:N] @ znew; lo_ = A[N:] @ znew
        dz = znew - z
        dl = A[:N] @ dz; im = int(np.abs(dl).argmax())
        print(f"   argmax sample {im} (u={u[im]:.4f}), kappa there {kap[im]:.3e}, lat {lat[im]:.3f} in [{lo[im]:.2f},{hi[im]:.2f}], dl around: {np.round(dl[max(0,im-60):im+61:15],3)}")
        print(f"   sample step max: lateral {np.abs(A[:N] @ dz).max():.3e} m, longitudinal {np.abs(A[N:] @ dz).max():.3e} m")
        print(f"   ipm its {its}, step max {np.abs(znew - z).max():.3e}, cond(K) {cond:.2e}, P eig min/max {ev[0]:.2e}/{ev[-1]:.2e}, "
              f"active lat {(np.minimum(lat - lo[:N], hi[:N] - lat) < 1e-6).sum()}, active lon {(np.minimum(lo_ + lon, lon - lo_) < 1e-6).sum()}")
        if LS:
            def cost(zz):
                X, Y = c0x + zz[0::2], c0y + zz[1::2]
                dx, dy, ddx, ddy = B1 @ X, B1 @ Y, B2 @ X, B2 @ Y
                s2 = dx * dx + dy * dy
                kk = (dx * ddy - dy * ddx) / (s2 * np.sqrt(s2))
                return kk @ kk
            cands = [(cost(z + tau * (znew - z)), tau) for tau in LS]
            best = min(cands)
            print("   line search:", [(f"{c:.10e}", tau) for c, tau in cands], "->", best[1])
            znew = z + best[1] * (znew - z)
        z = znew
    print("total ipm its", tot)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:2]])
