"""CPU oracle: NumPy/SciPy restatement of the KNODE-Cosserat rod hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``knode-cosserat_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors produced by importing
the unmodified reference in the build container (``tests/golden/make_golden.py``
is the generating script; the reference itself never travels).

Algorithm citations are ``file:line`` relative to the reference checkout
(``knode_cosserat/``):

* parameters / derived terms .......... cosserat_ode.py:5-78
* residual MLP evaluator .............. cosserat_ode.py:90-112
* per-segment spatial derivative ...... cosserat_ode.py:114-186
* Euler / RK4 shooting residuals ...... cosserat_ode.py:188-255
* parameter presets ................... knode.py:6-53
* BDF2 time loop + fsolve shooting .... knode.py:55-102
* tendon-tension generators ........... physics_controls.py:3-33
* quaternion -> Euler (loss term) ..... Utils/transformations.py:3-31

The state convention is the reference's: ``y = [p(3) h(4) n(3) m(3) q(3) w(3)]``
(19 rows) and ``z = [v(3) u(3)]`` (6 rows), arrays shaped ``[rows, N]``.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field

import numpy as np

NY = 19
NZ = 6
NS = NY + NZ


# --------------------------------------------------------------------------
# parameters (cosserat_ode.py:5-78, knode.py:6-53)
# --------------------------------------------------------------------------
def _default_tendon_dirs(n_tendons: int = 4) -> np.ndarray:
    # cosserat_ode.py:35-41 - four tendons at theta = pi/4 + k*pi/2, no z part
    th0 = math.pi / n_tendons
    rows = []
    for k in range(4):
        a = th0 + k * math.pi / 2
        rows.append([math.cos(a), math.sin(a), 0.0])
    return np.array(rows, dtype=np.float64)


@dataclass
class RodParams:
    """Independent parameters; defaults are the class defaults of the
    reference (cosserat_ode.py:15-47)."""

    L: float = 0.4
    N: int = 10
    E: float = 109e9
    r: float = 0.0012
    rho: float = 8000.0
    vstar: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 1.0]))
    g: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    Bse: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))
    Bbt: np.ndarray = field(default_factory=lambda: np.diag([3e-2, 3e-2, 3e-2]))
    C: np.ndarray = field(default_factory=lambda: np.array([1e-4, 1e-4, 1e-4]))
    del_t: float = 0.005
    F_tip: np.ndarray = field(default_factory=lambda: np.zeros(3))
    M_tip: np.ndarray = field(default_factory=lambda: np.zeros(3))
    tendon_dirs: np.ndarray = field(default_factory=_default_tendon_dirs)
    p0: np.ndarray = field(default_factory=lambda: np.zeros(3))
    h0: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0, 0.0]))
    q0: np.ndarray = field(default_factory=lambda: np.zeros(3))
    w0: np.ndarray = field(default_factory=lambda: np.zeros(3))

    def derived(self) -> "Derived":
        return Derived(self)


class Derived:
    """Dependent terms, cosserat_ode.py:58-78."""

    def __init__(self, P: RodParams):
        self.P = P
        self.N = int(P.N)
        r2 = P.r * P.r
        self.A = math.pi * r2
        self.G = P.E / (2 * (1 + 0.3))
        self.ds = P.L / (P.N - 1)
        Ixx = math.pi * r2 * r2 / 4
        self.J = np.diag([Ixx, Ixx, 2 * Ixx])
        self.Kse = np.diag([self.G * self.A, self.G * self.A, P.E * self.A])
        self.Kbt = np.diag([P.E * Ixx, P.E * Ixx, self.G * 2 * Ixx])
        self.c0 = 1.5 / P.del_t
        self.c1 = -2.0 / P.del_t
        self.c2 = 0.5 / P.del_t
        self.Kse_inv = np.linalg.inv(self.Kse + self.c0 * np.asarray(P.Bse, float))
        self.Kbt_inv = np.linalg.inv(self.Kbt + self.c0 * np.asarray(P.Bbt, float))
        self.Kse_vstar = self.Kse @ np.asarray(P.vstar, float)
        self.rhoA = P.rho * self.A
        self.rhoAg = self.rhoA * np.asarray(P.g, float)
        self.rhoJ = P.rho * self.J
        self.Bse = np.asarray(P.Bse, float)
        self.Bbt = np.asarray(P.Bbt, float)
        self.C = np.asarray(P.C, float)
        self.y0_head = np.concatenate([P.p0, P.h0]).astype(float)
        self.y0_tail = np.concatenate([P.q0, P.w0]).astype(float)


MODS = (None, "noair", "nsw", "short", "damping", "dampstiff", "lengthstiff", "youngs")


def setup_params(mod=None, N: int = 10, base: RodParams | None = None) -> RodParams:
    """Experimental preset of knode.py:6-53 applied to a parameter set."""
    P = copy.deepcopy(base) if base is not None else RodParams()
    P.N = N
    P.del_t = 0.05
    P.L = 0.635
    P.r = 0.003175
    P.rho = 1411.6751
    P.E = 2.757903e9
    bbt = 3e-2
    if mod is None:
        pass
    elif mod == "noair":
        P.C = np.zeros(3)
    elif mod == "nsw":
        P.g = np.zeros(3)
    elif mod == "short":
        P.L = 0.4
    elif mod == "damping":
        bbt = 0.2
    elif mod == "dampstiff":
        bbt = 0.2
        P.E = 10e9
    elif mod == "lengthstiff":
        P.L = 0.4
        P.E = 10e9
    elif mod == "youngs":
        P.E = 10e9
    else:
        raise Exception("Unknown mod " + str(mod))
    P.Bbt = np.diag([bbt, bbt, bbt])
    return P


# --------------------------------------------------------------------------
# residual MLP (cosserat_ode.py:90-112)
# --------------------------------------------------------------------------
ACT_NONE, ACT_TANH, ACT_SOFTPLUS, ACT_RELU, ACT_ELU = 0, 1, 2, 3, 4
_ACT_BY_NAME = {"tanh": ACT_TANH, "softplus": ACT_SOFTPLUS, "relu": ACT_RELU, "elu": ACT_ELU,
                "none": ACT_NONE, "identity": ACT_NONE}


@dataclass
class Mlp:
    """Dense stack ``x -> act_k(W_k x + b_k)``; the activation code of the last
    layer is normally ACT_NONE.  ``history`` selects the 53-wide input."""

    weights: list
    biases: list
    acts: list
    history: bool = False

    @property
    def in_dim(self):
        return self.weights[0].shape[1]


def _activate(code: int, x: np.ndarray) -> np.ndarray:
    if code == ACT_NONE:
        return x
    if code == ACT_TANH:
        return np.tanh(x)
    if code == ACT_SOFTPLUS:  # stable form used at cosserat_ode.py:92
        return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0)
    if code == ACT_RELU:
        return np.maximum(0, x)
    if code == ACT_ELU:
        return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1)
    raise ValueError(code)


def mlp_eval(mlp: Mlp, x: np.ndarray) -> np.ndarray:
    a = x
    for W, b, act in zip(mlp.weights, mlp.biases, mlp.acts):
        a = _activate(act, W @ a + b)  # float32 weights promote to float64 like numpy does in the reference
    return a


def make_mlp(sizes, acts="elu", seed=0, history=False, dtype=np.float32) -> Mlp:
    """Random weights following cosserat_ode_torch.py:76-105
    (|N(0.01,0.01)| weights, N(0,0.01) biases), drawn with NumPy so that the
    same numbers can be handed to every implementation."""
    rng = np.random.default_rng(seed)
    if isinstance(acts, str):
        acts = [acts] * (len(sizes) - 2)
    Ws, bs, codes = [], [], []
    for k in range(len(sizes) - 1):
        Ws.append(np.abs(rng.normal(0.01, 0.01, size=(sizes[k + 1], sizes[k]))).astype(dtype))
        bs.append(rng.normal(0.0, 0.01, size=(sizes[k + 1],)).astype(dtype))
        codes.append(_ACT_BY_NAME[acts[k]] if k < len(sizes) - 2 else ACT_NONE)
    return Mlp(Ws, bs, codes, history)


# --------------------------------------------------------------------------
# per-segment spatial derivative (cosserat_ode.py:114-186)
# --------------------------------------------------------------------------
def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def quat_rotation(h) -> np.ndarray:
    """Eq. 10 with an un-normalised quaternion, cosserat_ode.py:133-137."""
    a, b, c, d = h
    s = 2.0 / (a * a + b * b + c * c + d * d)
    return np.array([
        [1 + s * (-c * c - d * d), s * (b * c - d * a), s * (b * d + c * a)],
        [s * (b * c + d * a), 1 + s * (-b * b - d * d), s * (c * d - b * a)],
        [s * (b * d - c * a), s * (c * d + b * a), 1 + s * (-b * b - c * c)],
    ])


def ode(D: Derived, y, yh, zh, tendon_forces, mlp: Mlp | None = None):
    """One evaluation of the rod's arc-length derivative.

    Returns ``(ys[19], z[6])``.  With an MLP the network correction is added to
    ``ys`` and ``z`` *after* every physics term has been formed from the
    uncorrected ``z`` (cosserat_ode.py:178-184)."""
    h, n, m, q, w = y[3:7], y[7:10], y[10:13], y[13:16], y[16:19]
    vh, uh = zh[0:3], zh[3:6]
    R = quat_rotation(h)

    v = D.Kse_inv @ (R.T @ n + D.Kse_vstar - D.Bse @ vh)
    u = D.Kbt_inv @ (R.T @ m - D.Bbt @ uh)

    qt = D.c0 * q + yh[13:16]
    wt = D.c0 * w + yh[16:19]
    vt = D.c0 * v + vh
    ut = D.c0 * u + uh

    f = D.rhoAg - R @ (D.C * q * np.abs(q)) + tendon_forces
    ps = R @ v
    ns = D.rhoA * (R @ (_cross(w, q) + qt)) - f
    ms = R @ (_cross(w, D.rhoJ @ w) + D.rhoJ @ wt) - _cross(ps, n)
    qs = vt - _cross(u, q) + _cross(w, v)
    ws = ut - _cross(u, w)
    hs = 0.5 * np.array([
        -u[0] * h[1] - u[1] * h[2] - u[2] * h[3],
        u[0] * h[0] + u[2] * h[2] - u[1] * h[3],
        u[1] * h[0] - u[2] * h[1] + u[0] * h[3],
        u[2] * h[0] + u[1] * h[1] - u[0] * h[2],
    ])
    ys = np.concatenate([ps, hs, ns, ms, qs, ws])
    z = np.concatenate([v, u])
    if mlp is not None:
        if mlp.history:
            x = np.concatenate([y, yh, z, zh, tendon_forces])
        else:
            x = np.concatenate([y, z, tendon_forces])
        out = mlp_eval(mlp, x)
        ys = ys + out[:NY]
        z = z + out[NY:]
    return ys, z


# --------------------------------------------------------------------------
# shooting residuals (cosserat_ode.py:188-255)
# --------------------------------------------------------------------------
def tendon_force(D: Derived, tensions) -> np.ndarray:
    return np.asarray(tensions, float) @ D.P.tendon_dirs  # cosserat_ode.py:195


def residual_euler(D: Derived, G, y, z, yh, zh, tensions, mlp=None):
    """Explicit-Euler sweep.  Mutates ``y`` and ``z`` in place exactly like the
    reference: column 0 of ``y`` is rebuilt from the boundary conditions and
    ``G``; ``z[:, N-1]`` is never written."""
    G = np.asarray(G, float)
    y[:, 0] = np.concatenate([D.y0_head, G[0:3], G[3:6], D.y0_tail])
    tf = tendon_force(D, tensions)
    for j in range(D.N - 1):
        ys, z[:, j] = ode(D, y[:, j], yh[:, j], zh[:, j], tf, mlp)
        y[:, j + 1] = y[:, j] + D.ds * ys
    return np.concatenate([D.P.F_tip - y[7:10, -1], D.P.M_tip - y[10:13, -1]])


def residual_rk4(D: Derived, G, y, z, yh, yh_int, zh, zh_int, tensions, mlp=None):
    """Classical RK4 sweep, cosserat_ode.py:215-255: stages 2-3 use the
    midpoint histories, stage 4 the history of column j+1; ``z[:, j]`` is taken
    from stage 1 only."""
    G = np.asarray(G, float)
    y[:, 0] = np.concatenate([D.y0_head, G[0:3], G[3:6], D.y0_tail])
    tf = tendon_force(D, tensions)
    ds = D.ds
    for j in range(D.N - 1):
        yj = y[:, j]
        k1, z[:, j] = ode(D, yj, yh[:, j], zh[:, j], tf, mlp)
        k2, _ = ode(D, yj + k1 * ds / 2, yh_int[:, j], zh_int[:, j], tf, mlp)
        k3, _ = ode(D, yj + k2 * ds / 2, yh_int[:, j], zh_int[:, j], tf, mlp)
        k4, _ = ode(D, yj + k3 * ds, yh[:, j + 1], zh[:, j + 1], tf, mlp)
        y[:, j + 1] = yj + ds * (k1 + 2 * (k2 + k3) + k4) / 6
    return np.concatenate([D.P.F_tip - y[7:10, -1], D.P.M_tip - y[10:13, -1]])


# --------------------------------------------------------------------------
# time loop (knode.py:55-102)
# --------------------------------------------------------------------------
def straight_state(D: Derived):
    """knode.py:58-64: straight rod along +z, unit quaternion, v = e3."""
    N = D.N
    y = np.zeros((NY, N))
    y[2] = np.linspace(0, D.P.L, N)
    y[3] = 1.0
    z = np.zeros((NZ, N))
    z[2] = 1.0
    return y, z


def newton_shoot(fun, G0, tol=1e-12, maxit=50, fd_eps=1e-7, damped=True):
    """Newton on the 6 shooting unknowns with a forward-difference Jacobian; the
    stopping rule is on the Newton update.  ``damped``: a step that does not
    reduce the residual norm is halved until it does (backtracking) - where the
    full step already reduces it, which is every step of every fixture but the
    untrained 512-wide network, the iterates are those of plain Newton.  ``fun``
    must leave the swept state at the point it was last called with, so after
    convergence one more call at the accepted ``G`` leaves the caller's y, z
    consistent."""
    G = np.array(G0, float)
    it = 0
    ok = False
    r0 = fun(G)
    for it in range(1, maxit + 1):
        Jm = np.empty((6, 6))
        for c in range(6):
            e = fd_eps * max(abs(G[c]), 1.0)
            Gp = G.copy()
            Gp[c] += e
            Jm[:, c] = (fun(Gp) - r0) / e
        with np.errstate(all="ignore"):
            try:
                d = np.linalg.solve(Jm, r0)
            except np.linalg.LinAlgError:
                break
        if not np.all(np.isfinite(d)):
            break
        if np.max(np.abs(d)) <= tol * max(1.0, np.max(np.abs(G))):
            ok = True
            break
        lam, n0 = 1.0, np.linalg.norm(r0)
        while True:
            with np.errstate(all="ignore"):
                r1 = fun(G - lam * d)
            n1 = np.linalg.norm(r1)
            if not damped or (np.isfinite(n1) and n1 <= n0 * (1.0 - 1e-4 * lam)) or lam < 1e-4:
                break
            lam *= 0.5
        G = G - lam * d
        r0 = r1
    fun(G)  # final sweep at the accepted point
    return G, ok, it


def simulate(D: Derived, ctl, mlp: Mlp | None = None, scheme: str = "euler", solver: str = "fsolve",
             tol: float = 1e-12, return_info: bool = False, xtol: float = 1.49012e-8):
    """BDF2 time stepping with shooting at every step.

    ``solver='fsolve'`` is the reference's own choice (MINPACK hybrd through
    SciPy, knode.py:89); the returned y, z are whatever the last residual call
    left behind.  ``solver='lbfgs'`` is its ``use_fsolve=False`` branch.  ``solver='newton'`` is the tightly converged variant the HIP
    kernels implement.  Output: ``float64[T, 50, N]`` with rows
    ``[y; z; yh; zh]``; entry 0 is the initial state and the last solved step
    is dropped (knode.py:102)."""
    from scipy.optimize import fsolve

    ctl = np.asarray(ctl, float)
    y, z = straight_state(D)
    y_prev, z_prev = y.copy(), z.copy()
    G = np.zeros(6)
    out = [np.vstack([y, z, y, z])]
    info = {"ier": [], "nfev": [], "G": []}
    for tensions in ctl:
        yh = D.c1 * y + D.c2 * y_prev
        zh = D.c1 * z + D.c2 * z_prev
        y_prev, z_prev = y.copy(), z.copy()
        if scheme == "euler":
            fun = lambda g: residual_euler(D, g, y, z, yh, zh, tensions, mlp)
        elif scheme == "rk4":
            yh_int = 0.5 * (yh[:, :-1] + yh[:, 1:])
            zh_int = 0.5 * (zh[:, :-1] + zh[:, 1:])
            fun = lambda g: residual_rk4(D, g, y, z, yh, yh_int, zh, zh_int, tensions, mlp)
        else:
            raise ValueError(scheme)
        if solver == "fsolve":
            G, fo, ier, _ = fsolve(fun, G, full_output=True, xtol=xtol)
            info["ier"].append(ier)
            info["nfev"].append(fo["nfev"])
        elif solver == "lbfgs":
            # the use_fsolve=False branch, knode.py:91-94: L-BFGS-B on the sum of squared residuals
            # (cosserat_ode.py:212-213); y, z are left at the minimiser's last function call
            from scipy.optimize import minimize
            res = minimize(lambda g: float(np.sum(fun(g) ** 2)), G, method="L-BFGS-B")
            G = res.x
            info["ier"].append(1 if res.success else 5)
            info["nfev"].append(res.nfev)
        else:
            G, ok, it = newton_shoot(fun, G, tol=tol)
            info["ier"].append(1 if ok else 5)
            info["nfev"].append(7 * it + 1)
        info["G"].append(np.array(G))
        out.append(np.vstack([y.copy(), z.copy(), yh, zh]))
    traj = np.array(out)[:-1]
    if return_info:
        info = {k: np.array(v) for k, v in info.items()}
        return traj, info
    return traj


# --------------------------------------------------------------------------
# inputs (physics_controls.py:3-33) and the loss helper (transformations.py:3-31)
# --------------------------------------------------------------------------
def calc_controls(control_type, control_arg, del_t, train_len):
    np.random.seed(int(control_arg))
    rows = []
    for i in range(1, train_len + 1):
        if control_type == "sine":
            period_steps = control_arg / del_t
            rows.append([6 + np.sin(2 * np.pi * i / period_steps + k * (2 * np.pi / 4)) for k in range(4)])
        elif control_type == "step":
            s = 0 if i * del_t < 1.5 else control_arg
            rows.append([5 + s, 5, 5, 5 + s])
        elif control_type == "random":
            rows.append([5 + 5 * np.random.rand() for _ in range(4)])
        else:
            raise Exception("Unknown control type " + control_type)
    return rows


def quaternion_to_euler(quat: np.ndarray) -> np.ndarray:
    """[4, a] -> [3, a]; the (non-standard) angle formulas of
    Utils/transformations.py:25-27 reproduced literally."""
    qn = quat / np.sqrt(np.sum(quat * quat, axis=0, keepdims=True))
    w, x, y, z = qn
    roll = np.arctan2(2 * (w * y + x * z), 1 - 2 * (y * y + z * z))
    pitch = np.arcsin(np.clip(2 * (w * z - x * y), -1.0, 1.0))
    yaw = np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + z * z))
    return np.stack([roll, pitch, yaw], axis=0)


def batch_sine_controls(B: int, T: int, del_t: float, seed: int):
    """Synthetic per-rod tensions of SURVEY section 8d (cfg2/cfg3): rod b gets
    ``6 + sin(2*pi*i*dt/P_b + phi_b + k*pi/2)``, P_b ~ U[0.5,3] s,
    phi_b ~ U[0,2*pi).  Returns float64[B, T, 4]."""
    rng = np.random.default_rng(seed)
    P = rng.uniform(0.5, 3.0, size=B)
    phi = rng.uniform(0.0, 2 * np.pi, size=B)
    i = np.arange(1, T + 1)[None, :, None]
    k = np.arange(4)[None, None, :]
    return 6.0 + np.sin(2 * np.pi * i * del_t / P[:, None, None] + phi[:, None, None] + k * (np.pi / 2))


def mlp_from_arrays(d, prefix: str) -> Mlp:
    """Rebuild an :class:`Mlp` from the ``{prefix}_W{k}`` / ``_b{k}`` / ``_acts`` /
    ``_history`` entries the golden fixtures store."""
    acts = [int(a) for a in d[f"{prefix}_acts"]]
    Ws = [np.asarray(d[f"{prefix}_W{k}"]) for k in range(len(acts))]
    bs = [np.asarray(d[f"{prefix}_b{k}"]) for k in range(len(acts))]
    return Mlp(Ws, bs, acts, bool(int(d[f"{prefix}_history"])))


def params_for(mod, N: int) -> RodParams:
    """``'default'`` = class defaults, anything else = knode.setup_robot preset."""
    if mod == "default":
        P = RodParams()
        P.N = N
        return P
    return setup_params(None if mod in (None, "None") else mod, N)
