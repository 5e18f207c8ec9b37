"""ctypes wrapper of oracle/cosserat_oracle_c.c (TEST INFRASTRUCTURE - only tests/ and bench.py's cpu_baseline
leg import it).  ``simulate(P, ctl, traj=True)`` mirrors ``cosserat_oracle.simulate(D, ctl, solver='newton')``
except that all T controls are solved and the initial state is entry 0 of the trajectory."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OrcParams(C.Structure):
    _fields_ = [("L", C.c_double), ("E", C.c_double), ("r", C.c_double), ("rho", C.c_double), ("del_t", C.c_double),
                ("N", C.c_int), ("vstar", C.c_double * 3), ("g", C.c_double * 3), ("Bse", C.c_double * 9),
                ("Bbt", C.c_double * 9), ("C", C.c_double * 3), ("F_tip", C.c_double * 3), ("M_tip", C.c_double * 3),
                ("tendon_dirs", C.c_double * 12), ("p0", C.c_double * 3), ("h0", C.c_double * 4),
                ("q0", C.c_double * 3), ("w0", C.c_double * 3)]


_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "lib", "liboracle_c.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _lib = C.CDLL(path)
        _lib.orc_simulate.restype = C.c_int
        _lib.orc_simulate.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def params_from(P) -> OrcParams:
    """P: cosserat_oracle.RodParams"""
    o = OrcParams()
    o.L, o.E, o.r, o.rho, o.del_t, o.N = float(P.L), float(P.E), float(P.r), float(P.rho), float(P.del_t), int(P.N)
    for name in ("vstar", "g", "C", "F_tip", "M_tip", "p0", "h0", "q0", "w0", "Bse", "Bbt", "tendon_dirs"):
        a = np.asarray(getattr(P, name), dtype=np.float64).ravel()
        getattr(o, name)[:] = a.tolist()
    return o


def simulate(P, ctl, traj=True):
    """-> (tip[T, 3], traj[T+1, 25, N] or None, n_unconverged).  Releases the GIL: call it from threads."""
    lib = load()
    ctl = np.ascontiguousarray(ctl, dtype=np.float64).reshape(-1, 4)
    T = ctl.shape[0]
    tip = np.empty((T, 3))
    tr = np.empty((T + 1, 25, int(P.N))) if traj else None
    o = params_from(P)
    bad = lib.orc_simulate(C.byref(o), T, ctl.ctypes.data, tip.ctypes.data, tr.ctypes.data if traj else None)
    if bad < 0:
        raise ValueError("bad parameters")
    return tip, tr, bad
