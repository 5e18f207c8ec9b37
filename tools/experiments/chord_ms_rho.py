"""Contraction of the chord iteration as a function of Jacobian age (CPU prototype, see chord_ms_proto.py).
Usage: python chord_ms_rho.py [P] [T] [B]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from chord_ms_proto import ChordMS, co

def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    mode = sys.argv[4] if len(sys.argv) > 4 else "chord"
    Prm = co.setup_params(None, N=100); D = Prm.derived()
    ctl_all = co.batch_sine_controls(1024, T, Prm.del_t, 1235)
    rng = np.random.default_rng(1235); per = rng.uniform(0.5, 3.0, size=1024)
    order = np.argsort(per)[:B]          # the B fastest rods
    ctl = ctl_all[order]
    print("periods", np.sort(per)[:B].round(2))
    S = ChordMS(D, B, P)
    Yint = S.y[:, S.starts[1:-1]].copy(); G = S.G.copy()
    sols, jacs, ctx = [], [], []
    for t in range(T):
        tf = ctl[:, t] @ D.P.tendon_dirs
        yh = D.c1 * S.y + D.c2 * S.y_prev; zh = D.c1 * S.z + D.c2 * S.z_prev
        qh, wh = yh[..., 13:16], yh[..., 16:19]
        S.y_prev, S.z_prev = S.y.copy(), S.z.copy()
        ny, nz = S.y.copy(), S.z.copy()
        for it in range(20):
            A = S.jacobians(G, Yint, qh, wh, zh, tf)
            c, tip, E = S.residual(G, Yint, qh, wh, zh, tf, store=(ny, nz))
            dG, dY = S.solve_linear(A, c, tip)
            dn = max(np.abs(dG).max(), np.abs(dY).max())
            if dn < 1e-11: break
            G = G + dG; Yint = Yint + dY
        S.y, S.z = ny, nz; S.y[:, -1] = E[:, -1]; S.z[:, -1] = S.z_prev[:, -1]
        sols.append((G.copy(), Yint.copy())); jacs.append(A); ctx.append((qh, wh, zh, tf))
    # analysis
    def err(G, Y, t):
        Gs, Ys = sols[t]
        return np.maximum(np.abs((G - Gs) / np.maximum(1, np.abs(Gs))).max(-1),
                          np.abs((Y - Ys) / np.maximum(1, np.abs(Ys))).reshape(B, -1).max(-1))
    for age in (0, 1, 2, 3, 4, 6, 8):
        rows = []
        for t in range(30, T):
            qh, wh, zh, tf = ctx[t]
            Gs, Ys = sols[t]
            # start error: direction of the quadratic-extrapolation error, scaled to 6e-5 (max norm)
            gp = 3 * sols[t-1][0] - 3 * sols[t-2][0] + sols[t-3][0]; yp = 3 * sols[t-1][1] - 3 * sols[t-2][1] + sols[t-3][1]
            eG, eY = gp - Gs, yp - Ys
            n0 = err(gp, yp, t)
            sc = 6e-5 / n0
            G = Gs + eG * sc[:, None]; Y = Ys + eY * sc[:, None, None]
            A = jacs[t - age].copy()
            es = [err(G, Y, t)]
            prev = None
            for k in range(5):
                c, tip, E = S.residual(G, Y, qh, wh, zh, tf)
                if mode == "broyden" and prev is not None:
                    # per-interval secant update: dE_i = E_i(new) - E_i(old) ; A_i += (dE - A dY) dY^T / dY^T dY  (non-p columns)
                    Gp, Yp, Ep = prev
                    dE = E - Ep                                        # [B,P,19]
                    dx = np.zeros((B, P, 16))
                    dx[:, 0, 4:10] = G - Gp
                    dx[:, 1:] = (Y - Yp)[..., 3:]
                    dp = np.zeros((B, P, 3)); dp[:, 1:] = (Y - Yp)[..., :3]
                    pred = np.einsum("bprc,bpc->bpr", A, dx); pred[..., :3] += dp
                    r = dE - pred
                    den = (dx * dx).sum(-1, keepdims=True) + 1e-300
                    A = A + r[..., :, None] * (dx / den)[..., None, :]
                prev = (G.copy(), Y.copy(), E.copy())
                dG, dY = S.solve_linear(A, c, tip)
                G = G + dG; Y = Y + dY
                es.append(err(G, Y, t))
            rows.append(np.array(es))
        R = np.array(rows)        # [steps, its, B]
        worst = R.max(axis=(0, 2)); med = np.median(R, axis=(0, 2))
        print("age %d  worst err by iteration: %s | median %s" % (age, " ".join("%.1e" % v for v in worst), " ".join("%.1e" % v for v in med)))

main()
