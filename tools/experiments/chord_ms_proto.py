"""CPU prototype (NumPy, batched over rods) of chord multiple shooting.

Question it answers before any kernel is written: if the interval Jacobians A_i of a
P-interval multiple-shooting discretisation of the rod are NOT recomputed every
Newton iteration but reused for `age` time steps (and held / applied in fp32), how fast
does the chord iteration  x <- x - Jt^{-1} F(x)  contract on the bench workload
(B rods, N = 100, setup_robot parameters, sine tensions with periods 0.5..3 s)?

Run:  python tools/experiments/chord_ms_proto.py [P] [refresh_every] [steps]
Development tool only; imports the oracle for the parameters and the control draw.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import cosserat_oracle as co  # noqa: E402


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def ode_vec(D, y, qh, wh, zh, tf):
    """Batched ODE (diagonal material matrices). y[...,19]; qh, wh [...,3]; zh[...,6]; tf[...,3]."""
    h, n, m, q, w = y[..., 3:7], y[..., 7:10], y[..., 10:13], y[..., 13:16], y[..., 16:19]
    a, b, c, d = h[..., 0], h[..., 1], h[..., 2], h[..., 3]
    s = 2.0 / (a * a + b * b + c * c + d * d)
    R = np.empty(h.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 + s * (-c * c - d * d); R[..., 0, 1] = s * (b * c - d * a); R[..., 0, 2] = s * (b * d + c * a)
    R[..., 1, 0] = s * (b * c + d * a); R[..., 1, 1] = 1 + s * (-b * b - d * d); R[..., 1, 2] = s * (c * d - b * a)
    R[..., 2, 0] = s * (b * d - c * a); R[..., 2, 1] = s * (c * d + b * a); R[..., 2, 2] = 1 + s * (-b * b - c * c)
    RT = np.swapaxes(R, -1, -2)
    vh, uh = zh[..., 0:3], zh[..., 3:6]
    mv = lambda M, x: np.einsum("...ij,...j->...i", M, x)
    v = (mv(RT, n) + D.Kse_vstar - vh @ D.Bse.T) @ D.Kse_inv.T
    u = (mv(RT, m) - uh @ D.Bbt.T) @ D.Kbt_inv.T
    qt = D.c0 * q + qh
    wt = D.c0 * w + wh
    vt = D.c0 * v + vh
    ut = D.c0 * u + uh
    f = D.rhoAg - mv(R, D.C * q * np.abs(q)) + tf
    ps = mv(R, v)
    ns = D.rhoA * mv(R, cross(w, q) + qt) - f
    ms = mv(R, cross(w, w @ D.rhoJ.T) + wt @ D.rhoJ.T) - cross(ps, n)
    qs = vt - cross(u, q) + cross(w, v)
    ws = ut - cross(u, w)
    u0, u1, u2 = u[..., 0], u[..., 1], u[..., 2]
    hs = 0.5 * np.stack([-u0 * b - u1 * c - u2 * d, u0 * a + u2 * c - u1 * d,
                         u1 * a - u2 * b + u0 * d, u2 * a + u1 * b - u0 * c], -1)
    return np.concatenate([ps, hs, ns, ms, qs, ws], -1), np.concatenate([v, u], -1)


class ChordMS:
    def __init__(self, D, B, P, fp32_jac=True):
        self.D, self.B, self.P = D, B, P
        N = D.N
        self.N = N
        self.starts = np.round(np.arange(P + 1) * (N - 1) / P).astype(int)  # grid index of interval starts; last = N-1
        self.len = np.diff(self.starts)
        self.Lmax = self.len.max()
        self.fp32 = fp32_jac
        y, z = co.straight_state(D)
        self.y = np.broadcast_to(y.T, (B, N, 19)).copy()
        self.z = np.broadcast_to(z.T, (B, N, 6)).copy()
        self.y_prev, self.z_prev = self.y.copy(), self.z.copy()
        self.G = np.zeros((B, 6))
        self.A = None   # [B,P,19,16(+...)] interval Jacobians wrt the 16 non-p comps (interval 0: wrt G in cols 0..5)
        self.hist = []

    def integrate(self, Y0, qh, wh, zh, tf, store=None):
        """Y0[B,P,C,19] start states (C columns); integrates every interval to its end.
        Returns end states [B,P,C,19]. store: (y,z) arrays to fill from column 0."""
        D = self.D
        Y = Y0.copy()
        for k in range(self.Lmax):
            idx = self.starts[:-1] + k                         # grid index per interval
            act = k < self.len                                 # [P]
            idc = np.minimum(idx, self.N - 2)
            ys, z = ode_vec(D, Y, qh[:, idc][:, :, None], wh[:, idc][:, :, None], zh[:, idc][:, :, None], tf[:, None, None])
            if store is not None:
                sy, sz = store
                for i in np.nonzero(act)[0]:
                    sy[:, idx[i]] = Y[:, i, 0]
                    sz[:, idx[i]] = z[:, i, 0]
            Y = np.where(act[None, :, None, None], Y + D.ds * ys, Y)
        return Y

    def start_states(self, G, Yint):
        """G[B,6], Yint[B,P-1,19] -> Y0[B,P,19]"""
        D = self.D
        y0 = np.concatenate([np.broadcast_to(D.y0_head, (self.B, 7)), G, np.broadcast_to(D.y0_tail, (self.B, 6))], -1)
        return np.concatenate([y0[:, None], Yint], 1)

    def jacobians(self, G, Yint, qh, wh, zh, tf, eps=1e-7):
        B, P = self.B, self.P
        Y0 = self.start_states(G, Yint)                       # [B,P,19]
        cols = np.repeat(Y0[:, :, None], 17, 2)               # col 0 base, 1..16 perturbed comps 3..18
        hstep = np.zeros((B, P, 16))
        for c in range(16):
            comp = 3 + c
            if True:
                st = eps * np.maximum(1.0, np.abs(Y0[:, :, comp]))
                # interval 0: only G comps (7..12) are unknowns; others unused
                cols[:, :, 1 + c, comp] += st
                hstep[:, :, c] = st
        E = self.integrate(cols, qh, wh, zh, tf)
        A = (E[:, :, 1:] - E[:, :, :1]) / hstep[..., None]   # [B,P,16(col),19(row)]
        A = np.swapaxes(A, -1, -2)                            # [B,P,19,16]
        if self.fp32:
            A = A.astype(np.float32).astype(np.float64)
        return A

    def residual(self, G, Yint, qh, wh, zh, tf, store=None):
        Y0 = self.start_states(G, Yint)
        E = self.integrate(Y0[:, :, None], qh, wh, zh, tf, store)[:, :, 0]   # [B,P,19]
        c = E[:, :-1] - Yint                                   # interface jumps [B,P-1,19]
        tip = np.concatenate([self.D.P.F_tip - E[:, -1, 7:10], self.D.P.M_tip - E[:, -1, 10:13]], -1)
        return c, tip, E

    def solve_linear(self, A, c, tip):
        """Block substitution with the (frozen) Jacobians. Returns dG[B,6], dY[B,P-1,19] (the Newton update to ADD)."""
        B, P = self.B, self.P
        f = (lambda x: x.astype(np.float32).astype(np.float64)) if self.fp32 else (lambda x: x)
        # unknown update dx: dY_1 = c_0 + A_0[:, :, G cols] dG ; dY_{i+1} = c_i + A_i dY_i(3:)
        # write dY_i = a_i + M_i dG
        a = np.zeros((B, P + 1, 19)); M = np.zeros((B, P + 1, 19, 6))
        # interval 0: unknown G = comps 7..12 -> columns 4..9 of the 16
        M0 = A[:, 0][:, :, 4:10]
        a[:, 1] = c[:, 0] if P > 1 else 0
        a_cur = None
        # generic recurrence
        avec = np.zeros((B, 19)); Mmat = None
        # i = 0
        e_a = np.zeros((B, 19)); e_M = M0.copy()               # end-of-interval-0 sensitivity (before jump)
        for i in range(1, P + 1):
            # dY_i = (jump c_{i-1}) + end sens of interval i-1
            if i <= P - 1:
                a_i = f(c[:, i - 1] + e_a); M_i = e_M
                a[:, i] = a_i; M[:, i] = M_i
                Ai = A[:, i]                                   # [B,19,16]
                e_a = f(np.einsum("brc,bc->br", Ai, a_i[:, 3:]))
                e_a[:, :3] += a_i[:, :3]                       # p rows: identity on p
                e_M = f(np.einsum("brc,bcg->brg", Ai, M_i[:, 3:]))
                e_M[:, :3] += M_i[:, :3]
            else:
                pass
        # tip: residual_tip(x+dx) ~ tip - (e_a + e_M dG)[n,m rows] = 0
        T = e_M[:, 7:13]                                       # [B,6,6]
        rhs = tip - e_a[:, 7:13]
        dG = np.linalg.solve(T, rhs[..., None])[..., 0]
        dG = f(dG)
        dY = a[:, 1:P] + np.einsum("bprg,bg->bpr", M[:, 1:P], dG)
        return dG, f(dY)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    refresh = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    pred_order = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    Prm = co.setup_params(None, N=100)
    D = Prm.derived()
    ctl = co.batch_sine_controls(1024, T, Prm.del_t, 1235)
    # pick the B fastest-period rods? use the first B and report per-rod worst
    ctl = ctl[:B]
    S = ChordMS(D, B, P)
    tol = 1e-8
    Yint = S.y[:, S.starts[1:-1]].copy()
    G = S.G.copy()
    past = []
    tot_sweeps = 0
    log = []
    age = 10 ** 9
    for t in range(T):
        tf = ctl[:, t] @ D.P.tendon_dirs
        yh = D.c1 * S.y + D.c2 * S.y_prev
        zh = D.c1 * S.z + D.c2 * S.z_prev
        qh, wh = yh[..., 13:16], yh[..., 16:19]
        S.y_prev, S.z_prev = S.y.copy(), S.z.copy()
        # predictor: polynomial extrapolation of the unknowns
        x = np.concatenate([G, Yint.reshape(B, -1)], -1)
        past.append(x)
        k = min(pred_order, len(past) - 1)
        if k == 0:
            xp = past[-1]
        elif k == 1:
            xp = 2 * past[-1] - past[-2]
        elif k == 2:
            xp = 3 * past[-1] - 3 * past[-2] + past[-3]
        else:
            xp = 4 * past[-1] - 6 * past[-2] + 4 * past[-3] - past[-4]
        G = xp[:, :6].copy(); Yint = xp[:, 6:].reshape(B, P - 1, 19).copy()
        if age >= refresh:
            S.A = S.jacobians(G, Yint, qh, wh, zh, tf)
            age = 0
        norms = []
        ny, nz = S.y.copy(), S.z.copy()
        for it in range(12):
            c, tip, E = S.residual(G, Yint, qh, wh, zh, tf, store=(ny, nz))
            dG, dY = S.solve_linear(S.A, c, tip)
            scale = 1.0  # max(1,|x|) ~ 1 here for most comps; use relative to max(1,|x|)
            dn = np.maximum(np.abs(dG / np.maximum(1, np.abs(G))).max(-1),
                            np.abs(dY / np.maximum(1, np.abs(Yint))).reshape(B, -1).max(-1))
            norms.append(dn)
            if (dn <= tol).all():
                break
            upd = dn > tol
            G = np.where(upd[:, None], G + dG, G)
            Yint = np.where(upd[:, None, None], Yint + dY, Yint)
        norms = np.array(norms)        # [its, B]
        sweeps = (norms > tol).sum(0) + 1
        tot_sweeps += sweeps
        # contraction rate estimate from consecutive norms where both > 1e-11
        with np.errstate(divide="ignore", invalid="ignore"):
            rho = norms[1:] / norms[:-1]
        S.y, S.z = ny, nz
        S.y[:, -1] = E[:, -1]
        S.z[:, -1] = S.z_prev[:, -1]
        age += 1
        worst = np.argmax(sweeps)
        log.append((t, age - 1, sweeps.max(), sweeps.mean(), norms[0].max(),
                    norms[1].max() if len(norms) > 1 else 0, norms[2].max() if len(norms) > 2 else 0,
                    norms[3].max() if len(norms) > 3 else 0))
        print("t=%3d age=%2d sweeps max %d mean %.2f | d0 %.1e d1 %.1e d2 %.1e d3 %.1e" % log[-1], flush=True)
    print("mean sweeps/step over rods, steps>=30:", np.mean([l[3] for l in log[30:]]),
          "max-rod mean:", (tot_sweeps / T).max())
    # accuracy vs single-shooting newton oracle on rod 0
    tr = co.simulate(D, ctl[0], solver="newton", tol=1e-12)
    tip_ref = tr[-1][0:3, -1]
    print("rod0 tip now", S.y_prev[0, -1, :3], "oracle (state before last step)", tip_ref)


if __name__ == "__main__":
    main()
