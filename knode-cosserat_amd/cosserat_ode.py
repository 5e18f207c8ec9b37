"""``CosseratRod`` - the NumPy-facing rod model of the reference, backed by HIP.

Drop-in for ``knode_cosserat/cosserat_ode.py`` (reference lines 4-255): same
constructor, same mutable attributes, same method names, argument meaning and
in-place mutation of the caller's ``y`` / ``z``.  All arithmetic runs in fp64 on
the MI355X through ``libknode_rod.so``; there is no CPU implementation in this
package, so every compute method raises ``KrError`` without a GPU.

Attribute handling: the independent parameters (``N, L, E, r, rho, g, C, Bbt,
del_t, ...``) are plain attributes that callers overwrite (knode.py:11-52);
``compute_intermediate_terms()`` refreshes the dependent ones
(``ds, c0, c1, c2, Kse, Kbt, rhoA, rhoAg, rhoJ, Kse_plus_c0_Bse_inv, ...``)
with the host-side routine of the library and every compute call pushes the
current parameter set to the device handle.
"""
from __future__ import annotations

import numpy as np

import krod_native as kn

_ACT_BY_STR = {
    "Tanh()": kn.ACT_TANH,
    "Softplus(beta=1.0, threshold=20.0)": kn.ACT_SOFTPLUS,
    "ReLU()": kn.ACT_RELU,
    "ELU(alpha=1.0)": kn.ACT_ELU,
}


def mlp_from_layer_strings(model, param_ls):
    """(weights, biases, acts) from the reference's representation of the MLP:
    a sequence of torch modules identified by ``str(layer)`` plus the flat
    state-dict parameter list (cosserat_ode.py:95-111)."""
    weights, biases, acts = [], [], []
    cnt = 0
    for i in range(len(model)):
        name = str(model[i])
        if name in _ACT_BY_STR:
            if not acts or acts[-1] != kn.ACT_NONE:
                raise kn.KrError(f"unsupported layer order: activation {name} does not follow a Linear layer")
            acts[-1] = _ACT_BY_STR[name]
        elif name.startswith("Dropout("):
            continue  # identity at evaluation time, cosserat_ode.py:107-108
        else:  # any other module is treated as affine, exactly like the reference does
            weights.append(np.asarray(param_ls[cnt], dtype=np.float32))
            biases.append(np.asarray(param_ls[cnt + 1], dtype=np.float32))
            acts.append(kn.ACT_NONE)
            cnt += 2
    return weights, biases, acts


class CosseratRod:
    def __init__(self, nn_path=None, use_fsolve=False, nn_input_history=False, device=0):
        self.verbose = False
        self.use_fsolve = use_fsolve
        self.nn_path = nn_path
        self.nn_input_history = nn_input_history
        self.device = device
        # independent parameters, reference defaults (cosserat_ode.py:15-47)
        self.L = 0.4
        self.N = 10
        self.E = 109e9
        self.r = 0.0012
        self.rho = 8000
        self.vstar = np.array([0, 0, 1])
        self.g = np.array([0, 0, -9.81])
        self.Bse = np.zeros((3, 3))
        self.Bbt = np.diag([3e-2, 3e-2, 3e-2])
        self.C = np.array([1e-4, 1e-4, 1e-4])
        self.del_t = 0.005
        self.F_tip = np.zeros(3)
        self.M_tip = np.zeros(3)
        self.T0 = 5
        self.n_tendons = 4
        self.tendon_tensions = None
        self.tendon_offset = 0.02
        th = np.pi / self.n_tendons
        self.tendon_dirs = np.array([[np.cos(th + k * np.pi / 2), np.sin(th + k * np.pi / 2), 0] for k in range(4)])
        self.p0 = np.zeros(3)
        self.h0 = np.array([1, 0, 0, 0])
        self.q0 = np.zeros(3)
        self.w0 = np.zeros(3)
        self._handle = None
        self._mlp_key = None
        self.compute_intermediate_terms()
        if self.nn_path is not None:
            self.nn_model, self.param_ls = self.get_nn_from_file()

    # ------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------
    def _params(self, live_vstar=False) -> kn.KrParams:
        # The device derives Kse_vstar from vstar on every push.  In the reference Kse_vstar only changes inside
        # compute_intermediate_terms() (cosserat_ode.py:75), while estimate_state.py:201 overwrites robot.vstar
        # without recomputing it - so the compute paths see the vstar of the last compute_intermediate_terms().
        vstar = self.vstar if live_vstar or getattr(self, "_vstar_derived", None) is None else self._vstar_derived
        return kn.params_from_dict(dict(
            L=self.L, N=int(self.N), nn_input_history=int(bool(self.nn_input_history)), E=self.E, r=self.r,
            rho=self.rho, vstar=vstar, g=self.g, Bse=self.Bse, Bbt=self.Bbt, C=self.C, del_t=self.del_t,
            F_tip=self.F_tip, M_tip=self.M_tip, tendon_dirs=self.tendon_dirs, p0=self.p0, h0=self.h0, q0=self.q0,
            w0=self.w0))

    def compute_intermediate_terms(self):
        """Dependent parameters, reference cosserat_ode.py:58-78 (computed by
        kr_derive on the host side of the library)."""
        self._vstar_derived = np.array(self.vstar, dtype=np.float64).copy()
        d = kn.derive(self._params(live_vstar=True))
        m3 = lambda a: np.array(a, dtype=np.float64).reshape(3, 3)
        self.A, self.G, self.ds = d.A, d.G, d.ds
        self.J, self.Kse, self.Kbt = m3(d.J), m3(d.Kse), m3(d.Kbt)
        self.c0, self.c1, self.c2 = d.c0, d.c1, d.c2
        self.Kse_plus_c0_Bse_inv = m3(d.Kse_plus_c0_Bse_inv)
        self.Kbt_plus_c0_Bbt_inv = m3(d.Kbt_plus_c0_Bbt_inv)
        self.Kse_vstar = np.array(d.Kse_vstar, dtype=np.float64)
        self.rhoA = d.rhoA
        self.rhoAg = np.array(d.rhoAg, dtype=np.float64)
        self.rhoJ = m3(d.rhoJ)

    def _native(self) -> kn.Handle:
        """Device handle carrying the current parameters (and MLP, if any)."""
        p = self._params()
        if self._handle is None:
            self._handle = kn.Handle(p, self.device)
            self._mlp_key = None
        else:
            self._handle.set_params(p)
        if self.nn_path is not None:
            self._push_mlp(self.nn_model, self.param_ls)
        return self._handle

    def _push_mlp(self, model, param_ls):
        """Packs and uploads the network only when it changed: the key is a digest of every parameter byte and
        of the layer strings, so repeated ``simulate`` / ``get_nn_output`` calls with the same weights reuse the
        packed copies the handle already holds (kr_set_mlp allocates and copies synchronously)."""
        import hashlib
        dg = hashlib.blake2b(digest_size=16)
        dg.update(("|".join(str(m) for m in model) + f"|{bool(self.nn_input_history)}").encode())
        for p_ in param_ls:
            a = np.ascontiguousarray(np.asarray(p_, dtype=np.float32))
            dg.update(str(a.shape).encode())
            dg.update(a.tobytes())
        key = dg.digest()
        if key != self._mlp_key:
            self._handle.set_mlp(*mlp_from_layer_strings(model, param_ls))
            self._mlp_key = key

    @property
    def _use_nn(self) -> bool:
        return self.nn_path is not None

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_handle"] = None
        st["_mlp_key"] = None
        return st

    # ------------------------------------------------------------------
    # residual MLP
    # ------------------------------------------------------------------
    def get_nn_from_file(self):
        """Reference cosserat_ode.py:81-88 (which hard-codes map_location='mps')."""
        import torch
        nn_model = torch.load(self.nn_path, map_location="cpu", weights_only=False)["robot"].nn_models
        param_ls = [t.detach().cpu().numpy() for _, t in nn_model.state_dict().items()]
        return nn_model, param_ls

    def get_nn_output(self, input, model, param_ls):
        """Reference cosserat_ode.py:90-112, evaluated on the device in fp64."""
        import torch
        if self._handle is None:
            saved, self.nn_path = self.nn_path, None  # (the handle is created without pushing self.nn_model)
            try:
                self._native()
            finally:
                self.nn_path = saved
        h = self._handle
        self._push_mlp(model, param_ls)
        x = torch.as_tensor(np.asarray(input, dtype=np.float64).reshape(1, -1), device=f"cuda:{self.device}")
        return h.mlp_eval(x.contiguous())[0].cpu().numpy()

    # ------------------------------------------------------------------
    # physics
    # ------------------------------------------------------------------
    def ODE(self, y, yh, zh, tendon_forces):
        """Arc-length derivative at one grid point -> (ys[19], z[6]);
        reference cosserat_ode.py:114-186."""
        import torch
        h = self._native()
        dev = f"cuda:{self.device}"
        t = lambda a, n: torch.as_tensor(np.asarray(a, dtype=np.float64).reshape(1, n), device=dev).contiguous()
        dys, z = h.ode_batch(t(y, 19), t(yh, 19), t(zh, 6), t(tendon_forces, 3), use_nn=self._use_nn)
        return dys[0].cpu().numpy(), z[0].cpu().numpy()

    def _residual(self, scheme, G, y, z, yh, zh, yh_int=None, zh_int=None):
        import torch
        h = self._native()
        dev = f"cuda:{self.device}"
        N = int(self.N)
        td = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
        hist = h.pack(td(yh).reshape(1, 19, N), td(zh).reshape(1, 6, N))
        nxt = h.new_state(1, torch.float64)
        tens = td(np.asarray(self.tendon_tensions, dtype=np.float64).reshape(1, 4))
        Gd = td(np.asarray(G, dtype=np.float64).reshape(1, 6))
        if yh_int is not None and zh_int is not None:
            # the caller's midpoint histories, one column per segment (cosserat_ode.py:225,233-234)
            ym = np.zeros((19, N)); zm = np.zeros((6, N))
            ym[:, : N - 1] = np.asarray(yh_int, dtype=np.float64)[:, : N - 1]
            zm[:, : N - 1] = np.asarray(zh_int, dtype=np.float64)[:, : N - 1]
            mid = h.pack(td(ym).reshape(1, 19, N), td(zm).reshape(1, 6, N))
            r = h.residual_mid(Gd, hist, mid, nxt, tens, scheme=scheme, use_nn=self._use_nn)
        else:
            r = h.residual(Gd, None, hist, nxt, tens, scheme=scheme, use_nn=self._use_nn, hist_is_explicit=True)
        y_new, z_new = h.unpack(nxt)
        # in-place mutation of the caller's arrays (cosserat_ode.py:194,200-201); the last column of z
        # is never written by a sweep
        y[:, :] = y_new[0].cpu().numpy()
        z[:, : N - 1] = z_new[0, :, : N - 1].cpu().numpy()
        res = r[0].cpu().numpy()
        if self.use_fsolve:
            return res
        return float(np.sum(res * res))

    def getResidualEuler(self, G, y, z, yh, yh_int, zh, zh_int):
        """One explicit-Euler shooting sweep; reference cosserat_ode.py:188-213.
        ``yh_int`` / ``zh_int`` are accepted and ignored, like in the reference."""
        return self._residual(kn.KR_EULER, G, y, z, yh, zh)

    def getResidualRK4(self, G, y, z, yh, yh_int, zh, zh_int):
        """Classical RK4 sweep; reference cosserat_ode.py:215-255.  Stages 2 and 3 read the
        midpoint histories the caller passes (``yh_int[:, j]``, ``zh_int[:, j]``, reference
        lines 225 and 233-234) - whatever they are, not a re-derived interpolation; ``None``
        for either selects the interpolation knode.simulate forms (knode.py:80-81)."""
        return self._residual(kn.KR_RK4, G, y, z, yh, zh, yh_int, zh_int)
