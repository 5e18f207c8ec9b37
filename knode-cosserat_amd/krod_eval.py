"""Trajectory files and evaluation metrics of the reference's drivers.

* ``save_trajectory`` / ``load_trajectory``: the ``.npy`` dictionary ``{"traj": [T, 50, N], "controls": [T-1, 4]}``
  of ``knode_cosserat_realworld/simulate.py:97-100`` (read back with ``np.load(..., allow_pickle=True).item()``,
  cosserat_ode.py:265) and the evaluation record of ``physics_multitrain.py:201-205``.
* ``dtw_distance``: the tip-trajectory metric of ``physics_train.py:161`` and ``physics_multitrain.py:211``.
  The reference calls ``fastdtw(a, b)[0]`` (radius 1, L1 point distance for vector samples), an
  *approximation* of dynamic time warping whose package is not available in this build container; this is
  the exact DTW with the same point distance - a lower bound of what fastdtw returns, equal to it whenever
  fastdtw's coarse path contains the optimal one.  Parity with fastdtw is therefore **unpinned**.
* ``pos_euler_mse``: ``physics_multitrain.py:213-222`` (position + zyx Euler angles, x 1000).
* ``evaluate``: the closed-loop rollout with live weights of ``physics_train.py:136-167``; the rollout is
  ``knode.simulate`` on the MI355X with the MLP inside the shooting sweeps.
"""
from __future__ import annotations

import numpy as np


def save_trajectory(path, traj, controls):
    np.save(path, {"traj": np.asarray(traj, dtype=np.float64), "controls": np.asarray(controls, dtype=np.float64)})


def load_trajectory(path):
    d = np.load(path, allow_pickle=True).item()
    return d["traj"], d["controls"]


def save_eval_record(path, tensions, reference, predicted):
    np.save(path, {"tensions": tensions, "reference": reference, "predicted": predicted})


def dtw_distance(a, b, p=1):
    """Exact dynamic-time-warping distance between the sample sequences a[Ta, d] and b[Tb, d] with the
    p-norm as point distance (1-D inputs: absolute difference).  O(Ta*Tb) work, vectorised along the
    anti-diagonals of the cost table."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    if b.ndim == 1:
        b = b[:, None]
    Ta, Tb = len(a), len(b)
    if Ta == 0 or Tb == 0:
        raise ValueError("empty sequence")
    diff = np.abs(a[:, None, :] - b[None, :, :])
    cost = diff.sum(-1) if p == 1 else (diff ** p).sum(-1) ** (1.0 / p)
    D = np.full((Ta + 1, Tb + 1), np.inf)
    D[0, 0] = 0.0
    for k in range(2, Ta + Tb + 1):  # cells (i, j), 1-based, with i + j = k
        i = np.arange(max(1, k - Tb), min(Ta, k - 1) + 1)
        j = k - i
        D[i, j] = cost[i - 1, j - 1] + np.minimum(np.minimum(D[i - 1, j], D[i, j - 1]), D[i - 1, j - 1])
    return float(D[Ta, Tb])


def pos_euler_mse(trajectory, reference):
    """physics_multitrain.py:213-222: mean of the squared position errors and the squared zyx-Euler-angle
    errors over all grid points and steps, times 1000.  trajectory, reference: [T, >=7, N]."""
    from scipy.spatial.transform import Rotation
    trajectory = np.asarray(trajectory)
    reference = np.asarray(reference)
    se_pos = (trajectory[:, :3] - reference[:, :3]).reshape((-1, 3)) ** 2
    eq = trajectory[:, 3:7].transpose((0, 2, 1)).reshape((-1, 4))
    rq = reference[:, 3:7].transpose((0, 2, 1)).reshape((-1, 4))
    ee = Rotation.from_quat(eq, scalar_first=True).as_euler("zyx")
    re = Rotation.from_quat(rq, scalar_first=True).as_euler("zyx")
    return float(np.mean(np.concatenate([(ee - re) ** 2, se_pos])) * 1000)


def evaluate(robot_eval, torch_robot, controls, reference, eval_len=None, tip_index=-1):
    """physics_train.py:136-167: put the current weights of `torch_robot` into the NumPy-side robot,
    roll it out in closed loop over `controls` and score the tip path against `reference[:, :3, tip]`.
    Returns (dtw, traj[T, 25, N])."""
    from knode import simulate
    if torch_robot is not None:
        nn_model = torch_robot.nn_models
        robot_eval.nn_model = nn_model
        robot_eval.param_ls = [t.detach().cpu().numpy() for _, t in nn_model.state_dict().items()]
        robot_eval.nn_path = "whatever"  # forces the robot to use the MLP (physics_train.py:144)
    controls = np.asarray(controls)
    n = len(controls) if eval_len is None else eval_len
    traj = simulate(robot_eval, controls[:n])[:n, :25]
    return dtw_distance(traj[:, :3, tip_index], np.asarray(reference)[:n, :3, tip_index]), traj
