"""``setup_robot`` / ``simulate`` - drop-in for ``knode_cosserat/knode.py``.

``simulate(robot, ctl)`` keeps the reference's contract (knode.py:55-102):
``ctl`` is a sequence of T tension 4-vectors, the result is
``float64[T, 50, N]`` with rows ``[y; z; yh; zh]``, entry 0 is the straight
initial rod (whose "history" rows are a copy of the state, knode.py:68) and the
last solved step is dropped (knode.py:102).  The time loop, the BDF2 history
terms and the shooting solve all run on the MI355X (``kr_simulate_batch``);
the shooting unknowns are found by Newton iteration instead of MINPACK hybrd /
L-BFGS-B - same root, see DESIGN.md.

``simulate_batch`` is the batched form the reference lacks: B rods with
individual tension histories in one call.
"""
from __future__ import annotations

import numpy as np

import krod_native as kn
from cosserat_ode import CosseratRod

_MODS = (None, "noair", "nsw", "short", "damping", "dampstiff", "lengthstiff", "youngs")


def setup_robot(robot, mod=None, original=False):
    """Experimental parameter set of the physical robot plus one of the
    model-mismatch variants; reference knode.py:6-53.  Works on ``CosseratRod``
    and ``CosseratRodTorch`` objects."""
    if original:
        raise Exception("--original parameter no longer supported")
    p = kn.KrParams()
    kn.check(kn.load().kr_default_params(p))
    # start from the robot's own values for the fields a modifier may leave untouched
    for k in range(3):
        p.C[k] = float(np.asarray(_to_numpy(robot.C)).reshape(-1)[k])
        p.g[k] = float(np.asarray(_to_numpy(robot.g)).reshape(-1)[k])
    rc = kn.load().kr_apply_preset(p, None if mod is None else str(mod).encode())
    if rc != 0:
        raise Exception("Unknown mod " + str(mod))
    robot.del_t = p.del_t
    robot.L = p.L
    robot.tendon_offset = 0.04445
    robot.r = p.r
    robot.rho = p.rho
    robot.E = p.E
    is_np = isinstance(robot, CosseratRod)
    if mod == "noair":
        robot.C = _like(robot, [0, 0, 0], is_np)
    elif mod == "nsw":
        robot.g = _like(robot, [0, 0, 0], is_np)
    bbt = p.Bbt[0]
    if is_np:
        robot.Bbt = np.diag([bbt, bbt, bbt])
    else:
        import torch
        robot.Bbt = torch.diag(torch.tensor([bbt, bbt, bbt], device=robot.device))
    robot.compute_intermediate_terms()


def setup_robot_original(robot, mod=None):
    """The legacy parameter set, knode_cosserat_realworld/prepare.py:35-73 (``setup_robot(robot, mod,
    original=True)`` there): del_t 0.005, L 0.4, E 209e9, r 0.0012, rho 8000, Bbt 5e-4, modifiers
    None / nsw / short / damping / diameter / youngs / dampstiff / lengthstiff."""
    p = kn.KrParams()
    kn.check(kn.load().kr_default_params(p))
    for k in range(3):
        p.g[k] = float(np.asarray(_to_numpy(robot.g)).reshape(-1)[k])
    rc = kn.load().kr_apply_preset_original(p, None if mod is None else str(mod).encode())
    if rc != 0:
        raise Exception("Unknown mod " + str(mod))
    robot.del_t, robot.L, robot.E, robot.r, robot.rho = p.del_t, p.L, p.E, p.r, p.rho
    is_np = isinstance(robot, CosseratRod)
    if mod == "nsw":
        robot.g = _like(robot, [0, 0, 0], is_np)
    bbt = p.Bbt[0]
    if is_np:
        robot.Bbt = np.diag([bbt, bbt, bbt])
    else:
        import torch
        robot.Bbt = torch.diag(torch.tensor([bbt, bbt, bbt], device=robot.device))
    robot.compute_intermediate_terms()


def _to_numpy(a):
    if hasattr(a, "detach"):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def _like(robot, values, is_np):
    if is_np:
        return np.array(values)
    import torch
    return torch.tensor(values, device=robot.device)


def simulate_batch(robot, ctl, dtype="f64", scheme="euler", return_states=True, tol=0.0, maxit=0, tip_only=False):
    """B rods, each with its own tension history.

    ctl: array-like [B, T, 4].  Returns a dict with
      ``tip``    float[B, T, 3]  tip position after each solved step,
      ``status`` int32[B, T]     0 converged / 1 iteration cap / 2 non-finite,
      ``traj``   float[B, T+1, 25, N] (reference row order, entry 0 = initial state) unless ``tip_only``.
    All T steps are solved (no off-by-one drop here)."""
    import torch
    h = robot._native()
    dev = f"cuda:{robot.device}"
    tdt = torch.float64 if dtype in ("f64", torch.float64, np.float64) else torch.float32
    ctl_t = torch.as_tensor(np.asarray(ctl, dtype=np.float64), device=dev).to(tdt).contiguous()
    B, T = ctl_t.shape[0], ctl_t.shape[1]
    n_slots = 3 if tip_only else T + 1
    states = h.new_state(B, tdt, n_slots=n_slots)
    h.init_straight(states[0])
    G = torch.zeros((B, 6), dtype=tdt, device=dev)  # knode.py:67
    tip = torch.empty((B, T, 3), dtype=tdt, device=dev)
    status = torch.zeros((B, T), dtype=torch.int32, device=dev)
    h.simulate(ctl_t, states, G, ring=tip_only, tip=tip, status=status,
               scheme=kn.KR_RK4 if scheme == "rk4" else kn.KR_EULER, tol=tol, maxit=maxit, use_nn=robot._use_nn)
    out = {"tip": tip.cpu().numpy(), "status": status.cpu().numpy(), "G": G.cpu().numpy()}
    if not tip_only and return_states:
        N = h.N
        traj = torch.empty((B, T + 1, 25, N), dtype=tdt, device=dev)
        for t in range(T + 1):
            y, z = h.unpack(states[t])
            traj[:, t, :19] = y
            traj[:, t, 19:] = z
        out["traj"] = traj.cpu().numpy()
    return out


def simulate(robot, ctl, robot_reference=None):
    """Reference knode.py:55-102.  The time loop is one ``kr_simulate_batch`` call, the ``[y; z; yh; zh]`` rows of all
    steps one ``kr_state_unpack50`` launch.  A step whose shooting solve did not converge raises a ``RuntimeWarning``
    (the reference's ``fsolve`` does the same through SciPy and carries on)."""
    import warnings
    import torch
    if robot_reference is None:
        robot_reference = robot
    ctl = np.asarray([np.asarray(c, dtype=np.float64) for c in ctl], dtype=np.float64).reshape(-1, 4)
    T = ctl.shape[0]
    N = int(robot_reference.N)
    if int(robot.N) != N:
        raise kn.KrError("robot and robot_reference must share N")
    if T == 0:  # np.array([initial])[:-1] in the reference
        return np.empty((0, 50, N), dtype=np.float64)
    # side effect of the reference's loop (knode.py:71): the robot is left holding the LAST control, including the
    # one whose solve is dropped - a following robot.getResidualEuler(...) by the caller reads it
    robot.tendon_tensions = ctl[-1].copy()
    h = robot._native()
    dev = f"cuda:{robot.device}"
    states = h.new_state(1, torch.float64, n_slots=max(T, 2))
    # the initial rod takes its length from robot_reference (knode.py:59)
    if robot_reference is not robot:
        robot_reference._native().init_straight(states[0])
    else:
        h.init_straight(states[0])
    status = None
    if T > 1:  # the T-th solve is dropped by the reference (knode.py:102), so it is not run
        G = torch.zeros((1, 6), dtype=torch.float64, device=dev)
        status = torch.zeros((1, T - 1), dtype=torch.int32, device=dev)
        ctl_t = torch.as_tensor(ctl[: T - 1].reshape(1, T - 1, 4), device=dev).contiguous()
        h.simulate(ctl_t, states, G, ring=False, use_nn=robot._use_nn, status=status)
    # entry t = [state t; c1 * state t-1 + c2 * state t-2] (knode.py:74-75,96): the T entries as one batch of T "rods"
    st = states[:T, 0]
    m1 = torch.cat([st[:1], st[: T - 1]])
    m2 = torch.cat([st[:1], st[:1], st[: T - 2]]) if T > 1 else st[:1]
    out = h.unpack50(st.contiguous(), m1.contiguous(), m2.contiguous())
    out[0, 25:] = out[0, :25]  # knode.py:68: entry 0 is vstack([y, z, y, z])
    res = out.cpu().numpy()
    if status is not None:
        bad = np.flatnonzero(status.cpu().numpy()[0])
        if bad.size:
            warnings.warn(f"shooting solve did not converge at step(s) {bad[:8].tolist()}"
                          f"{' ...' if bad.size > 8 else ''} of {T - 1}", RuntimeWarning)
    return res
