"""ctypes binding of libknode_rod.so (C ABI: include/knode_rod.h).

This module is the only place where the shared library is touched.  It fails
loudly: if the library is missing or a symbol is absent, importing the solver
classes raises - there is no CPU fallback anywhere in this package.

torch is used only as the owner of device memory and streams; every pointer
handed to the library is ``tensor.data_ptr()`` of a contiguous CUDA(HIP) tensor
and every call is queued on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KR_LIB_PATH") or os.path.join(_HERE, "lib", "libknode_rod.so")

KR_SLOTS = 28
KR_F32, KR_F64 = 0, 1
KR_EULER, KR_RK4 = 0, 1
ACT_NONE, ACT_TANH, ACT_SOFTPLUS, ACT_RELU, ACT_ELU = range(5)
ST_CONVERGED, ST_MAXIT, ST_NONFINITE = 0, 1, 2
KR_MAX_LAYERS = 8
KR_E_ARG, KR_E_UNSUPPORTED = -1, -4

# reference row (0..24 of [y; z]) -> packed slot, see knode_rod.h
ROW_TO_SLOT = np.array([12 + r for r in range(13)] + [r - 13 for r in range(13, 19)] + [6 + (r - 19) for r in range(19, 25)])


class KrParams(C.Structure):
    _fields_ = [
        ("L", C.c_double), ("N", C.c_int32), ("nn_input_history", C.c_int32),
        ("E", C.c_double), ("r", C.c_double), ("rho", C.c_double),
        ("vstar", C.c_double * 3), ("g", C.c_double * 3), ("Bse", C.c_double * 9), ("Bbt", C.c_double * 9),
        ("C", C.c_double * 3), ("del_t", C.c_double), ("F_tip", C.c_double * 3), ("M_tip", C.c_double * 3),
        ("tendon_dirs", C.c_double * 12), ("p0", C.c_double * 3), ("h0", C.c_double * 4),
        ("q0", C.c_double * 3), ("w0", C.c_double * 3),
    ]


class KrDerived(C.Structure):
    _fields_ = [
        ("A", C.c_double), ("G", C.c_double), ("ds", C.c_double), ("c0", C.c_double), ("c1", C.c_double),
        ("c2", C.c_double), ("rhoA", C.c_double),
        ("J", C.c_double * 9), ("Kse", C.c_double * 9), ("Kbt", C.c_double * 9),
        ("Kse_plus_c0_Bse_inv", C.c_double * 9), ("Kbt_plus_c0_Bbt_inv", C.c_double * 9),
        ("Kse_vstar", C.c_double * 3), ("rhoAg", C.c_double * 3), ("rhoJ", C.c_double * 9),
    ]


_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_PROTOS = {
    "kr_last_error": (C.c_char_p, []),
    "kr_version": (_int, []),
    "kr_default_params": (_int, [C.POINTER(KrParams)]),
    "kr_apply_preset": (_int, [C.POINTER(KrParams), C.c_char_p]),
    "kr_apply_preset_original": (_int, [C.POINTER(KrParams), C.c_char_p]),
    "kr_create": (_int, [C.POINTER(KrParams), _int, C.POINTER(_vp)]),
    "kr_destroy": (_int, [_vp]),
    "kr_set_option": (_int, [_vp, C.c_char_p, _int]),
    "kr_get_option": (_int, [_vp, C.c_char_p, C.POINTER(C.c_int)]),
    "kr_debug_buffer": (_int, [_vp, _vp]),
    "kr_set_params": (_int, [_vp, C.POINTER(KrParams)]),
    "kr_get_derived": (_int, [_vp, C.POINTER(KrDerived)]),
    "kr_derive": (_int, [C.POINTER(KrParams), C.POINTER(KrDerived)]),
    "kr_mlp_eval_batch": (_int, [_vp, _i64, _vp, _vp, _int, _vp]),
    "kr_set_mlp": (_int, [_vp, _int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_vp), C.POINTER(_vp), _int, _vp]),
    "kr_ode_batch": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp]),
    "kr_ode_vjp_batch": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp]),
    "kr_ode_jacobian_batch": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _int, _vp, _int, _vp]),
    "kr_state_init_straight": (_int, [_vp, _i64, _vp, _int, _vp]),
    "kr_state_pack": (_int, [_vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "kr_state_unpack": (_int, [_vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "kr_state_unpack50": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _int, _vp]),
    "kr_state_tip": (_int, [_vp, _i64, _vp, _vp, _int, _vp]),
    "kr_residual_batch": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "kr_simulate_prepare": (_int, [_vp, _i64, _int]),
    "kr_residual_mid_batch": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp]),
    "kr_step_batch": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, C.c_double, _int, _vp, _vp, _int, _vp, _int, _int, _vp]),
    "kr_simulate_batch": (_int, [_vp, _i64, _i64, _int, _vp, _vp, _int, _vp, _vp, C.c_double, _int, _vp, _int, _vp, _int, _vp]),
    "kr_next_segment_physics": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _int, _vp]),
    "kr_mlp_ws_bytes": (C.c_size_t, [_int, C.POINTER(C.c_int32), _i64]),
    "kr_mlp_forward": (_int, [_vp, _i64, _int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_vp), C.POINTER(_vp), _vp, _int, _vp, _vp, _vp]),
    "kr_mlp_forward_loss": (_int, [_vp, _i64, _int, _int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_vp), C.POINTER(_vp),
                                   _vp, _int, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "kr_mlp_backward": (_int, [_vp, _i64, _int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_vp), _vp, _int, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "kr_loss_fwd_bwd": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "kr_gather_targets": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "kr_adam_step": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double,
                            C.c_double, _i64, _i64, _vp]),
    "kr_adam_plateau_step": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double,
                                    C.c_double, _i64, _i64, _i64, C.c_double, _int, C.c_double, C.c_double, _vp, _vp]),
    "kr_train_epoch": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp,
                              C.c_double, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, _i64, C.c_double,
                              _int, C.c_double, C.c_double, _vp, _int, _int, _vp]),
    "kr_train_epochs": (_int, [_vp, _i64, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp,
                               C.c_double, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, _i64, C.c_double,
                               _int, C.c_double, C.c_double, _vp, _int, _vp]),
    "kr_loss_rows_fwd_bwd": (_int, [_vp, _i64, _int, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "kr_estimate_ws_bytes": (C.c_size_t, [_i64, _int]),
    "kr_estimate_state": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


class KrError(RuntimeError):
    code = 0


def load():
    """dlopen the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KrError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C knode-cosserat_amd/csrc`.  There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so both sides share one HIP runtime)
    except Exception:  # pragma: no cover - torch is optional for pure C users
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().kr_last_error()
        err = KrError(f"libknode_rod error {rc}: {msg.decode() if msg else '?'}")
        err.code = rc
        raise err


def dtype_code(t) -> int:
    import torch
    if t in (torch.float32, np.float32, "f32"):
        return KR_F32
    if t in (torch.float64, np.float64, "f64"):
        return KR_F64
    raise KrError(f"unsupported dtype {t}")


def _ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise KrError("libknode_rod works on device memory only: tensor is on " + str(t.device))
    if not t.is_contiguous():
        raise KrError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def derive(params: KrParams) -> KrDerived:
    """Host-only derivation of the dependent terms (no GPU needed)."""
    d = KrDerived()
    check(load().kr_derive(C.byref(params), C.byref(d)))
    return d


def params_from_dict(d: dict) -> KrParams:
    lib = load()
    p = KrParams()
    check(lib.kr_default_params(C.byref(p)))
    for k, v in d.items():
        cur = getattr(p, k)
        if isinstance(cur, C.Array):
            arr = np.asarray(v, dtype=np.float64).reshape(-1)
            if arr.size != len(cur):
                raise KrError(f"parameter {k}: expected {len(cur)} values, got {arr.size}")
            for i, x in enumerate(arr):
                cur[i] = float(x)
        else:
            setattr(p, k, type(cur)(v))
    return p


class Handle:
    """Owns one kr_handle (one device, one parameter set)."""

    def __init__(self, params: KrParams, device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise KrError("no HIP device visible: the rod solver runs on MI355X only (no CPU fallback)")
        self.lib = load()
        self.device = device
        torch.cuda.set_device(device)
        torch.cuda.current_stream()  # make sure torch's context exists on this device
        self._h = _vp()
        check(self.lib.kr_create(C.byref(params), device, C.byref(self._h)))
        self.N = params.N
        self._mlp_keep = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.kr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters --------------------------------------------------------
    def set_params(self, params: KrParams):
        check(self.lib.kr_set_params(self._h, C.byref(params)))
        self.N = params.N

    def derived(self) -> KrDerived:
        d = KrDerived()
        check(self.lib.kr_get_derived(self._h, C.byref(d)))
        return d

    def set_mlp(self, weights, biases, acts):
        """weights[k]: float32 [out, in] numpy arrays (nn.Linear layout)."""
        n = len(weights)
        if n == 0:
            check(self.lib.kr_set_mlp(self._h, 0, None, None, None, None, 0, _stream()))
            return
        Ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        dims = (C.c_int32 * (n + 1))(*([Ws[0].shape[1]] + [w.shape[0] for w in Ws]))
        for k in range(n):
            if Ws[k].shape[1] != dims[k] or bs[k].shape != (dims[k + 1],):
                raise KrError(f"MLP layer {k}: inconsistent shapes {Ws[k].shape} / {bs[k].shape}")
        acts_c = (C.c_int32 * n)(*[int(a) for a in acts])
        Wp = (_vp * n)(*[w.ctypes.data for w in Ws])
        bp = (_vp * n)(*[b.ctypes.data for b in bs])
        check(self.lib.kr_set_mlp(self._h, n, dims, acts_c, Wp, bp, 0, _stream()))

    # -- kernels -------------------------------------------------------------
    def ode_batch(self, y, yh, zh, tf, use_nn=False):
        import torch
        Q = y.shape[0]
        dys = torch.empty((Q, 19), dtype=y.dtype, device=y.device)
        z = torch.empty((Q, 6), dtype=y.dtype, device=y.device)
        check(self.lib.kr_ode_batch(self._h, Q, _ptr(y), _ptr(yh), _ptr(zh), _ptr(tf), _ptr(dys), _ptr(z),
                                    int(bool(use_nn)), dtype_code(y.dtype), _stream()))
        return dys, z

    def ode_vjp(self, y, yh, zh, tf, g_dys, g_z, cut=False, need=(True, True, True, True)):
        """J^T g of the physics of ``ode_batch`` with respect to (y, yh, zh, tf); entries not needed come back None."""
        import torch
        Q = y.shape[0]
        outs = [torch.empty((Q, n), dtype=y.dtype, device=y.device) if w else None for n, w in zip((19, 19, 6, 3), need)]
        check(self.lib.kr_ode_vjp_batch(self._h, Q, _ptr(y), _ptr(yh), _ptr(zh), _ptr(tf), _ptr(g_dys), _ptr(g_z),
                                        int(bool(cut)), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]),
                                        dtype_code(y.dtype), _stream()))
        return outs

    def ode_jacobian(self, y, yh, zh, tf, cut=False):
        """[Q, 25, 19] Jacobian d(dys, z) / dy of the physics of ``ode_batch``."""
        import torch
        Q = y.shape[0]
        jac = torch.empty((Q, 25, 19), dtype=y.dtype, device=y.device)
        check(self.lib.kr_ode_jacobian_batch(self._h, Q, _ptr(y), _ptr(yh), _ptr(zh), _ptr(tf), int(bool(cut)),
                                             _ptr(jac), dtype_code(y.dtype), _stream()))
        return jac

    def new_state(self, B, dtype, n_slots=1):
        import torch
        shape = (n_slots, B, self.N, KR_SLOTS) if n_slots > 1 else (B, self.N, KR_SLOTS)
        return torch.zeros(shape, dtype=dtype, device=f"cuda:{self.device}")

    def init_straight(self, state):
        B = state.shape[0]
        check(self.lib.kr_state_init_straight(self._h, B, _ptr(state), dtype_code(state.dtype), _stream()))
        return state

    def pack(self, y_fm, z_fm, state=None):
        B = y_fm.shape[0]
        if state is None:
            state = self.new_state(B, y_fm.dtype)
        check(self.lib.kr_state_pack(self._h, B, _ptr(y_fm), _ptr(z_fm), _ptr(state), dtype_code(y_fm.dtype), _stream()))
        return state

    def unpack(self, state):
        import torch
        B = state.shape[0]
        y = torch.empty((B, 19, self.N), dtype=state.dtype, device=state.device)
        z = torch.empty((B, 6, self.N), dtype=state.dtype, device=state.device)
        check(self.lib.kr_state_unpack(self._h, B, _ptr(state), _ptr(y), _ptr(z), dtype_code(state.dtype), _stream()))
        return y, z

    def unpack50(self, state, m1, m2, out=None):
        import torch
        B = state.shape[0]
        if out is None:
            out = torch.empty((B, 50, self.N), dtype=state.dtype, device=state.device)
        check(self.lib.kr_state_unpack50(self._h, B, _ptr(state), _ptr(m1), _ptr(m2), _ptr(out),
                                         dtype_code(state.dtype), _stream()))
        return out

    def tip(self, state):
        import torch
        B = state.shape[0]
        out = torch.empty((B, 3), dtype=state.dtype, device=state.device)
        check(self.lib.kr_state_tip(self._h, B, _ptr(state), _ptr(out), dtype_code(state.dtype), _stream()))
        return out

    def residual(self, G, prev, cur, nxt, tensions, scheme=KR_EULER, use_nn=False, hist_is_explicit=False):
        import torch
        B = G.shape[0]
        r = torch.empty((B, 6), dtype=G.dtype, device=G.device)
        check(self.lib.kr_residual_batch(self._h, B, scheme, _ptr(G), _ptr(prev), _ptr(cur), _ptr(nxt), _ptr(tensions),
                                         _ptr(r), int(bool(use_nn)), int(bool(hist_is_explicit)),
                                         dtype_code(G.dtype), _stream()))
        return r

    def residual_mid(self, G, hist, hist_mid, nxt, tensions, scheme=KR_RK4, use_nn=False):
        """Residual sweep from explicit histories; ``hist_mid`` (may be None) = the caller's midpoint histories."""
        import torch
        B = G.shape[0]
        r = torch.empty((B, 6), dtype=G.dtype, device=G.device)
        check(self.lib.kr_residual_mid_batch(self._h, B, scheme, _ptr(G), _ptr(hist), _ptr(hist_mid), _ptr(nxt),
                                             _ptr(tensions), _ptr(r), int(bool(use_nn)), dtype_code(G.dtype), _stream()))
        return r

    def mlp_eval(self, x):
        import torch
        Q = x.shape[0]
        out = torch.empty((Q, 25), dtype=x.dtype, device=x.device)
        check(self.lib.kr_mlp_eval_batch(self._h, Q, _ptr(x), _ptr(out), dtype_code(x.dtype), _stream()))
        return out

    def step(self, prev, cur, nxt, G, tensions, scheme=KR_EULER, tol=0.0, maxit=0, status=None, iters=None,
             use_nn=False, prev2=None, predictor=-1):
        B = G.shape[0]
        check(self.lib.kr_step_batch(self._h, B, scheme, _ptr(prev), _ptr(cur), _ptr(nxt), _ptr(G), _ptr(tensions),
                                     float(tol), int(maxit), _ptr(status), _ptr(iters), int(bool(use_nn)),
                                     _ptr(prev2), int(predictor), dtype_code(G.dtype), _stream()))

    def simulate_prepare(self, B, dtype):
        """One-time host work of the first ``simulate`` call for batches of B rods, ahead of time (launches nothing)."""
        check(self.lib.kr_simulate_prepare(self._h, int(B), dtype_code(dtype)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        check(self.lib.kr_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def set_option(self, name: str, value: int):
        check(self.lib.kr_set_option(self._h, name.encode(), int(value)))

    def simulate(self, ctl, states, G, ring=False, tip=None, status=None, scheme=KR_EULER, tol=0.0, maxit=0,
                 use_nn=False, prev_init=None):
        B, T = ctl.shape[0], ctl.shape[1]
        check(self.lib.kr_simulate_batch(self._h, B, T, scheme, _ptr(ctl), _ptr(states), int(bool(ring)), _ptr(G),
                                         _ptr(tip), float(tol), int(maxit), _ptr(status), int(bool(use_nn)),
                                         _ptr(prev_init), dtype_code(ctl.dtype), _stream()))
