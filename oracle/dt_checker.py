"""CPU checker (numpy, vectorised) for the double-track function evaluation of
include/rl_mincurv.h: rl_dt_eval_nodes -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Restates, formula by formula, models/double_track.py:10-140 (dynamics) and :143-204 (node
constraints), min_time_optimizer.py:93-163 (pairing of node i-1 with node i, cost) and
utils/utils.py:10-18 of the reference.  Pinned by fixture G8 (tests/golden/G8_double_track.npz): the
reference's own functions executed on numbers through a numeric stand-in for CasADi
(tests/golden/make_golden.py); tests/test_oracle_golden.py::test_g8_* holds this file to it (1e-13)."""
import numpy as np

G = 9.8
PARAMS = ["kd_f", "kb_f", "mass", "Jzz", "lf", "lr", "twf", "twr", "delta_max", "fr", "hcog", "kroll_f",
          "cl_f", "cl_r", "rho", "A", "cd", "mu", "Bf", "Cf", "Br", "Cr", "Pmax", "Fd_max", "Fb_max",
          "Td", "Tb", "Tdelta"]


def dynamics(m, x, u, k):
    """x [...,6] = s, n, xi, omega, beta, v; u [...,4]; k [...] curvature.  Returns xdot [...,6] and
    (Fx, Fy, Fz) each [...,4] in the order fl, fr, rl, rr."""
    n, phi, omega, beta, v = x[..., 1], x[..., 2], x[..., 3], x[..., 4], x[..., 5]
    drive = u[..., 0] * (np.tanh(u[..., 0]) * 0.5 + 0.5)
    brake = u[..., 0] * (np.tanh(-u[..., 0]) * 0.5 + 0.5)
    delta, transfer = u[..., 2], u[..., 3]
    lf, lr = m["lf"], m["lr"]
    l = lf + lr
    mass = m["mass"]
    fx_front = 0.5 * m["kd_f"] * drive + 0.5 * m["kb_f"] * brake - 0.5 * m["fr"] * mass * G * lr / l
    fx_rear = 0.5 * (1 - m["kd_f"]) * drive + 0.5 * (1 - m["kb_f"]) * brake - 0.5 * m["fr"] * mass * G * lf / l
    ax = (drive + brake - 0.5 * m["cd"] * m["A"] * v ** 2 - m["fr"] * mass * G) / mass
    fz_front = 0.5 * mass * G * lr / l - 0.5 * m["hcog"] / l * mass * ax + 0.25 * m["cl_f"] * m["rho"] * m["A"] * v ** 2
    fz_rear = 0.5 * mass * G * lr / l + 0.5 * m["hcog"] / l * mass * ax + 0.25 * m["cl_r"] * m["rho"] * m["A"] * v ** 2
    Fz = np.stack([fz_front - m["kroll_f"] * transfer, fz_front + m["kroll_f"] * transfer,
                   fz_rear - (1 - m["kroll_f"]) * transfer, fz_rear + (1 - m["kroll_f"]) * transfer], axis=-1)
    sb, cb = np.sin(beta), np.cos(beta)
    slip_fl = delta - np.arctan((lf * omega + v * sb) / (v * cb - 0.5 * m["twf"] * omega))
    slip_fr = delta - np.arctan((lf * omega + v * sb) / (v * cb + 0.5 * m["twf"] * omega))
    slip_rl = np.arctan((lr * omega - v * sb) / (v * cb - 0.5 * m["twr"] * omega))
    slip_rr = np.arctan((lr * omega - v * sb) / (v * cb + 0.5 * m["twr"] * omega))
    mu = m["mu"]
    Fy = np.stack([mu * Fz[..., 0] * np.sin(m["Cf"] * np.arctan(m["Bf"] * slip_fl)),
                   mu * Fz[..., 1] * np.sin(m["Cf"] * np.arctan(m["Bf"] * slip_fr)),
                   mu * Fz[..., 2] * np.sin(m["Cr"] * np.arctan(m["Br"] * slip_rl)),
                   mu * Fz[..., 3] * np.sin(m["Cr"] * np.arctan(m["Br"] * slip_rr))], axis=-1)
    Fx = np.stack([fx_front, fx_front, fx_rear, fx_rear], axis=-1)
    drag = 0.5 * m["cd"] * m["rho"] * m["A"] * v ** 2
    sum_fx_front, sum_fx_rear = Fx[..., 0] + Fx[..., 1], Fx[..., 2] + Fx[..., 3]
    sum_fy_front, sum_fy_rear = Fy[..., 0] + Fy[..., 1], Fy[..., 2] + Fy[..., 3]
    dv = (sum_fx_rear * cb + sum_fx_front * np.cos(delta - beta) + sum_fy_rear * sb - sum_fy_front * np.sin(delta - beta) - drag * cb) / mass
    dbeta = -omega + (-sum_fx_rear * sb + sum_fx_front * np.sin(delta - beta) + sum_fy_rear * cb + sum_fy_front * np.cos(delta - beta)
                         + drag * sb) / (mass * v)
    domega = ((Fx[..., 3] - Fx[..., 2]) * m["twr"] / 2 - sum_fy_rear * lr
                 + ((Fx[..., 1] - Fx[..., 0]) * np.cos(delta) + (Fy[..., 0] - Fy[..., 1]) * np.sin(delta)) * m["twf"] / 2
                 + (sum_fy_front * np.cos(delta) + sum_fx_front * np.sin(delta)) * lf) / m["Jzz"]
    ds_dt = v * np.cos(phi + beta) / (1 - n * k)
    dn_dt = v * np.sin(phi + beta)
    dxi_dt = omega - k * ds_dt
    return np.stack([ds_dt, dn_dt, dxi_dt, domega, dbeta, dv], axis=-1), (Fx, Fy, Fz)


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
    delta, transfer, v = U[..., 2], U[..., 3], X[..., 5]
    eq[..., 6] = transfer - m["hcog"] / (0.5 * (m["twf"] + m["twr"])) * (
        Fy[..., 2] + Fy[..., 3] + (Fx[..., 0] + Fx[..., 1]) * np.sin(delta) + (Fy[..., 0] + Fy[..., 1]) * np.cos(delta))
    eq[..., 7] = X[..., 0] - s[None, :]
    g = np.empty(X.shape[:2] + (14,))
    g[..., :4] = (Fx / (m["mu"] * Fz)) ** 2 + (Fy / (m["mu"] * Fz)) ** 2 - 1.0
    drive = U[..., 0] * (np.tanh(U[..., 0]) * 0.5 + 0.5)
    g[..., 4] = v * drive - m["Pmax"]
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
