"""CPU checker (numpy, vectorised) for the double-track function evaluation of
include/rl_mincurv.h: rl_dt_eval_nodes -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Restates, formula by formula, models/double_track.py:10-140 (dynamics) and :143-204 (node
constraints), min_time_optimizer.py:93-163 (pairing of node i-1 with node i, cost) and
utils/utils.py:10-18 of the reference.  PARITY UNPINNED: the reference evaluates these through CasADi,
which is not importable here, and ships no numeric fixture for them; what pins this
file is its agreement with the published model and the collocation-order test in
tests/test_double_track.py."""
import numpy as np

G = 9.8
PARAMS = ["kd_f", "kb_f", "mass", "Jzz", "lf", "lr", "twf", "twr", "delta_max", "fr", "hcog", "kroll_f",
          "cl_f", "cl_r", "rho", "A", "cd", "mu", "Bf", "Cf", "Br", "Cr", "Pmax", "Fd_max", "Fb_max",
          "Td", "Tb", "Tdelta"]


def dynamics(m, x, u, k):
    """x [...,6] = s, n, xi, omega, beta, v; u [...,4]; k [...] curvature.  Returns xdot [...,6] and
    (Fx, Fy, Fz) each [...,4] in the order fl, fr, rl, rr."""
    n, phi, omega, beta, v = x[..., 1], x[..., 2], x[..., 3], x[..., 4], x[..., 5]
    fd = u[..., 0] * (np.tanh(u[..., 0]) * 0.5 + 0.5)
    fb = u[..., 0] * (np.tanh(-u[..., 0]) * 0.5 + 0.5)
    delta, gam = u[..., 2], u[..., 3]
    lf, lr = m["lf"], m["lr"]
    l = lf + lr
    mass = m["mass"]
    Fx_f = 0.5 * m["kd_f"] * fd + 0.5 * m["kb_f"] * fb - 0.5 * m["fr"] * mass * G * lr / l
    Fx_r = 0.5 * (1 - m["kd_f"]) * fd + 0.5 * (1 - m["kb_f"]) * fb - 0.5 * m["fr"] * mass * G * lf / l
    ax = (fd + fb - 0.5 * m["cd"] * m["A"] * v ** 2 - m["fr"] * mass * G) / mass
    Fz_f = 0.5 * mass * G * lr / l - 0.5 * m["hcog"] / l * mass * ax + 0.25 * m["cl_f"] * m["rho"] * m["A"] * v ** 2
    Fz_r = 0.5 * mass * G * lr / l + 0.5 * m["hcog"] / l * mass * ax + 0.25 * m["cl_r"] * m["rho"] * m["A"] * v ** 2
    Fz = np.stack([Fz_f - m["kroll_f"] * gam, Fz_f + m["kroll_f"] * gam,
                   Fz_r - (1 - m["kroll_f"]) * gam, Fz_r + (1 - m["kroll_f"]) * gam], axis=-1)
    sb, cb = np.sin(beta), np.cos(beta)
    a_fl = delta - np.arctan((lf * omega + v * sb) / (v * cb - 0.5 * m["twf"] * omega))
    a_fr = delta - np.arctan((lf * omega + v * sb) / (v * cb + 0.5 * m["twf"] * omega))
    a_rl = np.arctan((lr * omega - v * sb) / (v * cb - 0.5 * m["twr"] * omega))
    a_rr = np.arctan((lr * omega - v * sb) / (v * cb + 0.5 * m["twr"] * omega))
    mu = m["mu"]
    Fy = np.stack([mu * Fz[..., 0] * np.sin(m["Cf"] * np.arctan(m["Bf"] * a_fl)),
                   mu * Fz[..., 1] * np.sin(m["Cf"] * np.arctan(m["Bf"] * a_fr)),
                   mu * Fz[..., 2] * np.sin(m["Cr"] * np.arctan(m["Br"] * a_rl)),
                   mu * Fz[..., 3] * np.sin(m["Cr"] * np.arctan(m["Br"] * a_rr))], axis=-1)
    Fx = np.stack([Fx_f, Fx_f, Fx_r, Fx_r], axis=-1)
    drag = 0.5 * m["cd"] * m["rho"] * m["A"] * v ** 2
    FxF, FxR = Fx[..., 0] + Fx[..., 1], Fx[..., 2] + Fx[..., 3]
    FyF, FyR = Fy[..., 0] + Fy[..., 1], Fy[..., 2] + Fy[..., 3]
    v_dot = (FxR * cb + FxF * np.cos(delta - beta) + FyR * sb - FyF * np.sin(delta - beta) - drag * cb) / mass
    beta_dot = -omega + (-FxR * sb + FxF * np.sin(delta - beta) + FyR * cb + FyF * np.cos(delta - beta)
                         + drag * sb) / (mass * v)
    omega_dot = ((Fx[..., 3] - Fx[..., 2]) * m["twr"] / 2 - FyR * lr
                 + ((Fx[..., 1] - Fx[..., 0]) * np.cos(delta) + (Fy[..., 0] - Fy[..., 1]) * np.sin(delta)) * m["twf"] / 2
                 + (FyF * np.cos(delta) + FxF * np.sin(delta)) * lf) / m["Jzz"]
    s_dot = v * np.cos(phi + beta) / (1 - n * k)
    n_dot = v * np.sin(phi + beta)
    phi_dot = omega - k * s_dot
    return np.stack([s_dot, n_dot, phi_dot, omega_dot, beta_dot, v_dot], axis=-1), (Fx, Fy, Fz)


def eval_nodes(m, s, kappa, left, right, margin, track_length, X, U, T):
    """X [B,N,6], U [B,N,4], T [B,N].  Returns eq [B,N,8], ineq [B,N,14], cost [B]."""
    Xn, Un = np.roll(X, -1, axis=1).copy(), np.roll(U, -1, axis=1)
    d = Xn[..., 2] - X[..., 2]
    Xn[..., 2] = np.arctan2(np.sin(d), np.cos(d)) + X[..., 2]
    ds = X[..., 0] - Xn[..., 0]
    kk = np.abs(ds) + track_length / 2.0
    Xn[..., 0] = Xn[..., 0] + (kk - np.fmod(kk, track_length)) * np.sign(ds)
    k = kappa[None, :]
    f1, (Fx, Fy, Fz) = dynamics(m, X, U, k)
    f2, _ = dynamics(m, Xn, U, k)
    t = T[..., None]
    Xm = 0.5 * (X + Xn) + (t / 8.0) * (f1 - f2)
    fm, _ = dynamics(m, Xm, U, k)
    eq = np.empty(X.shape[:2] + (8,))
    eq[..., :6] = X + (t / 6.0) * (f1 + 4 * fm + f2) - Xn
    delta, gam, v = U[..., 2], U[..., 3], X[..., 5]
    eq[..., 6] = gam - m["hcog"] / (0.5 * (m["twf"] + m["twr"])) * (
        Fy[..., 2] + Fy[..., 3] + (Fx[..., 0] + Fx[..., 1]) * np.sin(delta) + (Fy[..., 0] + Fy[..., 1]) * np.cos(delta))
    eq[..., 7] = X[..., 0] - s[None, :]
    g = np.empty(X.shape[:2] + (14,))
    g[..., :4] = (Fx / (m["mu"] * Fz)) ** 2 + (Fy / (m["mu"] * Fz)) ** 2 - 1.0
    fd = U[..., 0] * (np.tanh(U[..., 0]) * 0.5 + 0.5)
    g[..., 4] = v * fd - m["Pmax"]
    g[..., 5] = 1.0 - v
    g[..., 6] = m["Fb_max"] - U[..., 0]; g[..., 7] = U[..., 0] - m["Fd_max"]
    g[..., 8] = -m["delta_max"] - delta; g[..., 9] = delta - m["delta_max"]
    ru = (Un[..., 0] - U[..., 0]) / T; rd = (Un[..., 2] - delta) / T
    g[..., 10] = np.maximum(m["Fb_max"] / m["Tb"] - ru, ru - m["Fd_max"] / m["Td"])
    g[..., 11] = np.maximum(-m["delta_max"] / m["Tdelta"] - rd, rd - m["delta_max"] / m["Tdelta"])
    g[..., 12] = (right[None, :] + margin) - X[..., 1]; g[..., 13] = X[..., 1] - (left[None, :] - margin)
    su = np.array([m["Fd_max"], abs(m["Fb_max"]), m["delta_max"], m["mass"] * 50.0])
    Us, Uns = U / su, Un / su
    cost = T.sum(axis=1) + 1e-4 * (Uns ** 2).sum(axis=(1, 2)) + 1e-1 * ((Uns - Us) ** 2).sum(axis=(1, 2))
    return eq, g, cost
