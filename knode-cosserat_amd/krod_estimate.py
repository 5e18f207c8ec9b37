"""Full-state estimate from measured poses on the MI355X.

``estimate_state(data[T, 7, n], tensions[T, 4], robot) -> float64[T, 25, N]`` is the call of
``knode_cosserat_realworld/estimate_state.py:158-242``: positions / quaternions copied, velocities by
``np.gradient``, angular velocities from consecutive quaternions (:97-123), accelerations by second-order
``np.gradient``, strains from finite differences along the arc and the logarithm of the relative rotation
(:48-95), internal force and moment integrated from the tip backwards (:126-156, with the literal ``i != 9``
skip), strains re-estimated from the constitutive law with the BDF2 history of the previous estimate (:222-230).
All of it runs in four small fp64 kernels behind ``kr_estimate_state`` (csrc/kr_estimate.hip); like the reference
the call leaves ``robot.vstar`` at the re-estimated root strain of the first step (:197).  No CPU path.
"""
from __future__ import annotations

import numpy as np

import krod_native as kn


def estimate_state(data, tensions, robot):
    import torch
    data = np.ascontiguousarray(data, dtype=np.float64)
    tensions = np.ascontiguousarray(tensions, dtype=np.float64)
    T, seven, n = data.shape
    N = int(robot.N)
    if seven != 7 or n != N:
        raise kn.KrError(f"data must be [T, 7, N] with N = robot.N = {N}; got {data.shape}")
    if tensions.shape != (T, 4):
        raise kn.KrError(f"tensions must be [T, 4]; got {tensions.shape}")
    h = robot._native()
    dev = f"cuda:{robot.device}"
    d = torch.as_tensor(data, device=dev)
    tt = torch.as_tensor(tensions, device=dev)
    est = torch.empty((T, 25, N), dtype=torch.float64, device=dev)
    ws = torch.empty(max(int(h.lib.kr_estimate_ws_bytes(T, N)), 16), dtype=torch.uint8, device=dev)
    kn.check(h.lib.kr_estimate_state(h._h, T, kn._ptr(d), kn._ptr(tt), kn._ptr(est), kn._ptr(ws), kn._stream()))
    out = est.cpu().numpy()
    if T:
        robot.vstar = out[0, 19:22, 0]  # estimate_state.py:197 (a view in the reference as well)
    return out
