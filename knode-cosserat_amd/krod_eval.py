"""Trajectory files and evaluation metrics of the reference's drivers.

* ``save_trajectory`` / ``load_trajectory``: the ``.npy`` dictionary ``{"traj": [T, 50, N], "controls": [T-1, 4]}``
  of ``knode_cosserat_realworld/simulate.py:97-100`` (read back with ``np.load(..., allow_pickle=True).item()``,
  cosserat_ode.py:265) and the evaluation record of ``physics_multitrain.py:201-205``.
* ``fastdtw_distance``: the tip-trajectory metric of ``physics_train.py:159-161`` and ``physics_multitrain.py:211``.
  The reference calls ``fastdtw(a, b)[0]`` of the third-party package ``fastdtw`` (slaypni/fastdtw; no version is
  pinned by the reference and the package is absent from this build container): radius 1, L1 point distance for
  vector samples.  This is a restatement of the published algorithm (S. Salvador, P. Chan, "FastDTW: Toward accurate
  dynamic time warping in linear time and space", Intelligent Data Analysis 11(5), 2007): halve both series by
  averaging adjacent samples until one is shorter than radius + 2, solve exactly there, and refine the warp path on
  every finer level inside the projected path widened by the radius.  Ties between predecessors are broken in the
  order (i-1, j), (i, j-1), (i-1, j-1) - the order of the package's ``min`` - which matters for the window of the next
  level.  **Parity with the package is unpinned** (nothing to run it against); ``dtw_distance`` is the exact DTW
  with the same point distance, a lower bound of it (equal whenever the coarse path contains the optimal one).
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


def _l1(a, b):
    return float(np.abs(a - b).sum())


def _dtw_window(x, y, window):
    """DTW restricted to the cells of `window` (sorted by row, then column); returns (distance, path)."""
    inf = float("inf")
    D = {(0, 0): (0.0, 0, 0)}
    get = D.get
    for i, j in window:
        i1, j1 = i + 1, j + 1
        d = _l1(x[i], y[j])
        a = get((i1 - 1, j1), (inf,))[0]
        b = get((i1, j1 - 1), (inf,))[0]
        c = get((i1 - 1, j1 - 1), (inf,))[0]
        # first minimum in the order up, left, diagonal
        if a <= b and a <= c:
            D[i1, j1] = (a + d, i1 - 1, j1)
        elif b <= c:
            D[i1, j1] = (b + d, i1, j1 - 1)
        else:
            D[i1, j1] = (c + d, i1 - 1, j1 - 1)
    path = []
    i, j = len(x), len(y)
    while not (i == 0 and j == 0):
        path.append((i - 1, j - 1))
        _, i, j = D[i, j]
    path.reverse()
    return D[len(x), len(y)][0], path


def _expand_window(path, len_x, len_y, radius):
    cells = set(path)
    for i, j in path:
        for a in range(-radius, radius + 1):
            for b in range(-radius, radius + 1):
                cells.add((i + a, j + b))
    fine = set()
    for i, j in cells:
        fine.update(((2 * i, 2 * j), (2 * i, 2 * j + 1), (2 * i + 1, 2 * j), (2 * i + 1, 2 * j + 1)))
    window = []
    start_j = 0
    for i in range(len_x):
        new_start = None
        for j in range(start_j, len_y):
            if (i, j) in fine:
                window.append((i, j))
                if new_start is None:
                    new_start = j
            elif new_start is not None:
                break
        start_j = new_start
    return window


def _fastdtw(x, y, radius):
    if len(x) < radius + 2 or len(y) < radius + 2:
        return _dtw_window(x, y, [(i, j) for i in range(len(x)) for j in range(len(y))])
    half = lambda s: [(s[i] + s[i + 1]) / 2 for i in range(0, len(s) - len(s) % 2, 2)]
    _, path = _fastdtw(half(x), half(y), radius)
    return _dtw_window(x, y, _expand_window(path, len(x), len(y), radius))


def fastdtw_distance(a, b, radius=1):
    """FastDTW (see the module docstring) between the sample sequences a[Ta, d] and b[Tb, d], L1 point distance."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    if b.ndim == 1:
        b = b[:, None]
    if len(a) == 0 or len(b) == 0:
        raise ValueError("empty sequence")
    return float(_fastdtw([r for r in a], [r for r in b], int(radius))[0])


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


def evaluate(robot_eval, torch_robot, controls, reference, eval_len=None, tip_index=-1, exact=False):
    """physics_train.py:136-167: put the current weights of `torch_robot` into the NumPy-side robot,
    roll it out in closed loop over `controls` and score the tip path against `reference[:, :3, tip]` with FastDTW
    (radius 1, what the reference calls; ``exact=True``: the exact DTW).  Returns (dtw, traj[T, 25, N])."""
    from knode import simulate
    if torch_robot is not None:
        nn_model = torch_robot.nn_models
        robot_eval.nn_model = nn_model
        robot_eval.param_ls = [t.detach().cpu().numpy() for _, t in nn_model.state_dict().items()]
        robot_eval.nn_path = "whatever"  # forces the robot to use the MLP (physics_train.py:144)
    controls = np.asarray(controls)
    n = len(controls) if eval_len is None else eval_len
    traj = simulate(robot_eval, controls[:n])[:n, :25]
    metric = dtw_distance if exact else fastdtw_distance
    return metric(traj[:, :3, tip_index], np.asarray(reference)[:n, :3, tip_index]), traj
