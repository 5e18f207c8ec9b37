"""TEST INFRASTRUCTURE (CPU oracle; only tests/ may import it) - full-state estimate from measured poses, vectorised restatement of
``knode_cosserat_realworld/estimate_state.py:158-242``.

``estimate_state(data[T, 7, n], tensions[T, 4], robot) -> [T, 25, N]`` keeps the reference's recipe
step for step: positions/quaternions copied, velocities by ``np.gradient``, angular velocities from
consecutive quaternions (:97-123), accelerations by second-order ``np.gradient``, strains ``v, u`` from
finite differences along the arc and the logarithm of the relative rotation (:48-95), internal force and
moment integrated from the tip backwards with the step ``L / N`` (:126-156), then ``v, u`` re-estimated
from the constitutive law with the BDF2 history of the previous estimate (:222-230).  Quirks kept:
row 21 initialised to 1, the base x/y and base quaternion vector part zeroed, ``v[:, 0] = (0, 0, 1)``,
the backward integration skips the literal index ``i == 9`` (and, for N != 10, wraps into the tip column,
which is then not copied), at t = 0 the "previous" strains are the current ones.

Everything that the reference evaluates per (t, i) in Python loops (and one ``scipy.linalg.logm`` per
segment and step) is evaluated for all steps and grid points at once; only the two-term recurrence of the
re-estimated strains runs over t.  Pinned by tests/golden/estimate_state.npz (produced by the reference); the
product path is the device implementation behind ``kr_estimate_state`` (knode-cosserat_amd/krod_estimate.py).
"""
from __future__ import annotations

import numpy as np


def _rotations(h):
    """h[..., 4] (scalar first, not normalised) -> R[..., 3, 3], cosserat_ode.py:133-137."""
    a, b, c, d = (h[..., k] for k in range(4))
    s = 2.0 / (a * a + b * b + c * c + d * d)
    R = np.empty(h.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 + s * (-c * c - d * d); R[..., 0, 1] = s * (b * c - d * a); R[..., 0, 2] = s * (b * d + c * a)
    R[..., 1, 0] = s * (b * c + d * a); R[..., 1, 1] = 1 + s * (-b * b - d * d); R[..., 1, 2] = s * (c * d - b * a)
    R[..., 2, 0] = s * (b * d - c * a); R[..., 2, 1] = s * (c * d + b * a); R[..., 2, 2] = 1 + s * (-b * b - c * c)
    return R


def _log_so3(R):
    """Matrix logarithm of rotation matrices R[..., 3, 3] (what scipy.linalg.logm returns for them)."""
    tr = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(tr)
    skew = 0.5 * (R - np.swapaxes(R, -1, -2))
    small = th < 1e-6
    f = np.where(small, 1.0 + th * th / 6.0, th / np.where(small, 1.0, np.sin(th)))
    return f[..., None, None] * skew


def angular_velocities(quats, del_t):
    """estimate_state.py:97-123: quats[T, 4, N] -> w[T, 3, N]."""
    q1, q2 = quats[:-1], quats[1:]
    w = np.zeros((quats.shape[0], 3, quats.shape[2]))
    w[1:, 0] = q1[:, 0] * q2[:, 1] - q1[:, 1] * q2[:, 0] - q1[:, 2] * q2[:, 3] + q1[:, 3] * q2[:, 2]
    w[1:, 1] = q1[:, 0] * q2[:, 2] + q1[:, 1] * q2[:, 3] - q1[:, 2] * q2[:, 0] - q1[:, 3] * q2[:, 1]
    w[1:, 2] = q1[:, 0] * q2[:, 3] - q1[:, 1] * q2[:, 2] + q1[:, 2] * q2[:, 1] - q1[:, 3] * q2[:, 0]
    w *= 2.0 / del_t
    w[0] = w[1]
    return w


def estimate_state(data, tensions, robot):
    data = np.asarray(data, dtype=np.float64)
    tensions = np.asarray(tensions, dtype=np.float64)
    N = int(robot.N)
    T = data.shape[0]
    arc = np.linspace(0, robot.L, N)
    est = np.zeros((T, 25, N))
    est[:, 21, :] = 1
    est[:, :3, :] = data[:, :3, :]
    est[:, :2, 0] = 0
    est[:, 3:7, :] = data[:, 3:7, :]
    vel = np.gradient(est[:, :3, :], robot.del_t, axis=0, edge_order=1)
    est[:, 13:16, :] = vel
    ang = angular_velocities(est[:, 3:7, :], robot.del_t)
    est[:, 16:19, :] = ang
    qt = np.gradient(vel, robot.del_t, axis=0, edge_order=2)
    wt = np.gradient(ang, robot.del_t, axis=0, edge_order=2)

    pos = est[:, :3, :]                                   # [T, 3, N]
    R = _rotations(np.moveaxis(est[:, 3:7, :], 1, 2))     # [T, N, 3, 3]
    ds = np.diff(arc)                                     # [N-1]
    # strains from the measured curve (compute_v_u, :48-95)
    p_s = np.empty_like(pos)
    p_s[:, :, :-1] = (pos[:, :, 1:] - pos[:, :, :-1]) / ds
    p_s[:, :, -1] = p_s[:, :, -2]
    Rrel = R[:, 1:] @ np.swapaxes(R[:, :-1], -1, -2)
    Rs = np.empty_like(R)
    Rs[:, :-1] = R[:, :-1] @ (_log_so3(Rrel) / ds[None, :, None, None])
    Rs[:, -1] = Rs[:, -2]
    v = np.einsum("tnji,tjn->tin", R, p_s)                # R^T p_s
    uhat = np.swapaxes(R, -1, -2) @ Rs
    u = np.stack([uhat[..., 2, 1], uhat[..., 0, 2], uhat[..., 1, 0]], axis=1)   # [T, 3, N]
    v[:, 0:2, 0] = 0
    v[:, 2, 0] = 1

    # internal force and moment, tip to root (compute_internal_forces_and_moments, :126-156)
    tf = tensions @ np.asarray(robot.tendon_dirs, dtype=np.float64)            # [T, 3]
    C = np.asarray(robot.C, dtype=np.float64)
    drag = np.einsum("tnij,tjn->tin", R, C[None, :, None] * vel * np.abs(vel))
    f = np.asarray(robot.rhoAg)[None, :, None] - drag + tf[:, :, None]
    ns = robot.rhoA * np.einsum("tnij,tjn->tin", R, np.cross(ang, vel, axis=1) + qt) - f
    step = robot.L / N
    n = np.zeros((T, 3, N))
    for i in range(N):
        if i != 9:
            n[:, :, N - i - 2] = n[:, :, N - i - 1] - ns[:, :, N - i - 1] * step
    rhoJ = np.asarray(robot.rhoJ, dtype=np.float64)
    Jw = np.einsum("ij,tjn->tin", rhoJ, ang)
    ms = (np.einsum("tnij,tjn->tin", R, np.cross(ang, Jw, axis=1) + np.einsum("ij,tjn->tin", rhoJ, wt))
          - np.cross(p_s, n, axis=1))
    m = np.zeros((T, 3, N))
    for i in range(N):
        if i != 9:
            m[:, :, N - i - 2] = m[:, :, N - i - 1] - ms[:, :, N - i - 1] * step
    est[:, 7:10, :-1] = n[:, :, :-1]
    est[:, 10:13, :-1] = m[:, :, :-1]

    # strains re-estimated from the constitutive law with BDF2 history of the previous estimate (:222-230)
    Ksei = np.asarray(robot.Kse_plus_c0_Bse_inv, dtype=np.float64)
    Kbti = np.asarray(robot.Kbt_plus_c0_Bbt_inv, dtype=np.float64)
    Bse = np.asarray(robot.Bse, dtype=np.float64)
    Bbt = np.asarray(robot.Bbt, dtype=np.float64)
    Kse_vstar = np.asarray(robot.Kse_vstar, dtype=np.float64)
    Rtn = np.einsum("tnji,tjn->tin", R, est[:, 7:10, :])
    Rtm = np.einsum("tnji,tjn->tin", R, est[:, 10:13, :])
    v_prev = u_prev = None
    for t in range(T):
        vt_, ut_ = v[t], u[t]
        if t == 0:
            v_prev, u_prev = vt_, ut_
        vh = robot.c1 * vt_ + robot.c2 * v_prev
        uh = robot.c1 * ut_ + robot.c2 * u_prev
        vt_ = Ksei @ (Rtn[t] + Kse_vstar[:, None] - Bse @ vh)
        ut_ = Kbti @ (Rtm[t] - Bbt @ uh)
        est[t, 19:22, :] = vt_
        est[t, 22:, :] = ut_
        v_prev, u_prev = vt_, ut_
    est[:, 4:7, 0] = 0
    # :197 `robot.vstar = estimated_state[0, 19:22, 0]` is a view: it ends up as the re-estimated root strain
    # of the first step (dependent terms such as Kse_vstar are NOT recomputed by the reference)
    robot.vstar = est[0, 19:22, 0]
    return est
