"""``CosseratRodTorch`` - the trainable twin of the rod model, backed by HIP.

Drop-in for ``knode_cosserat/cosserat_ode_torch.py`` (reference lines 5-437):
same constructor ``CosseratRodTorch(device, n_layers, nn_input_history=False)``,
same attributes (``use_nn, y, z, tendon_tensions, residualArgs, nn_models,
layers, c0, c1, c2, ds, ...``) and methods (``forward, ODE, ODE_parallel,
getResidualEuler, getNextSegmentEuler, parallelGetNextSegmentEuler``).  The
object stays picklable for ``torch.save({'robot': robot})``
(physics_train.py:165,284).

What runs where:
  * rod physics ............ ``kr_ode_batch`` / ``kr_next_segment_physics`` (fp32, forward only:
                             the physics has no trainable parameter)
  * residual MLP ........... ``kr_mlp_forward`` / ``kr_mlp_backward`` - fp32 GEMMs on the matrix
                             cores, wrapped in a ``torch.autograd.Function`` so the reference's
                             training loop (torch loss -> ``backward()`` -> Adam) works unchanged
  * parameters, optimizer .. plain torch tensors (``nn.ModuleList`` of ``nn.Linear`` / ``nn.ELU``)

Gradients: the training paths produce them for the MLP parameters (what the reference's loops use).
``getResidualEuler(G)`` is differentiable with respect to G and the parameters through an adjoint sweep
(``_SweepFunction``), and ``ODE_parallel`` hands gradients to its *inputs* when they are asked for
(cosserat_ode_torch.py:264-306 lets autograd flow there; no caller of the reference uses it) - both off the hot
path: vector-Jacobian products of an fp64 torch graph of one grid point whose values are pinned to the ODE kernel.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

import krod_native as kn

_ACT_CODE = {nn.Tanh: kn.ACT_TANH, nn.Softplus: kn.ACT_SOFTPLUS, nn.ReLU: kn.ACT_RELU, nn.ELU: kn.ACT_ELU}


_ACT_FN = {kn.ACT_NONE: lambda t: t, kn.ACT_TANH: torch.tanh, kn.ACT_SOFTPLUS: nn.functional.softplus,
           kn.ACT_RELU: torch.relu, kn.ACT_ELU: nn.functional.elu}


def mlp_structure(modules):
    """[(linear, act_code), ...] from an ``nn.ModuleList`` of Linear / activation /
    Dropout modules (the structures ``forward`` of the reference can express)."""
    layers = []
    for m in modules:
        if isinstance(m, nn.Linear):
            layers.append([m, kn.ACT_NONE])
        elif type(m) in _ACT_CODE:
            if not layers or layers[-1][1] != kn.ACT_NONE:
                raise kn.KrError("unsupported MLP structure: activation without a preceding Linear layer")
            if isinstance(m, nn.Softplus) and (m.beta != 1.0 or m.threshold != 20.0):
                raise kn.KrError("only Softplus(beta=1, threshold=20) is supported")
            if isinstance(m, nn.ELU) and m.alpha != 1.0:
                raise kn.KrError("only ELU(alpha=1) is supported")
            layers[-1][1] = _ACT_CODE[type(m)]
        elif isinstance(m, nn.Dropout):
            continue
        else:
            raise kn.KrError(f"unsupported module in the residual MLP: {m}")
    return layers


class _MlpFunction(torch.autograd.Function):
    """out[Q, 32] = MLP(x[Q, in_pad]) on the matrix cores; backward fills the
    parameter gradients with kr_mlp_backward."""

    @staticmethod
    def forward(ctx, handle, x, dims, acts, *params):
        import ctypes as C
        n = len(acts)
        Ws = [p.contiguous() for p in params[0::2]]
        bs = [p.contiguous() for p in params[1::2]]
        Q = x.shape[0]
        dims_c = (C.c_int32 * (n + 1))(*dims)
        acts_c = (C.c_int32 * n)(*acts)
        Wp = (C.c_void_p * n)(*[w.data_ptr() for w in Ws])
        bp = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
        ws_bytes = handle.lib.kr_mlp_ws_bytes(n, dims_c, max(Q, 1))
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
        out = torch.empty((Q, 32), dtype=torch.float32, device=x.device)
        kn.check(handle.lib.kr_mlp_forward(handle._h, Q, n, dims_c, acts_c, Wp, bp, kn._ptr(x), x.shape[1],
                                           kn._ptr(out), kn._ptr(ws), kn._stream()))
        ctx.handle, ctx.dims, ctx.acts = handle, dims, acts
        ctx.biases = [b.detach() for b in bs]
        ctx.save_for_backward(x, ws, *Ws)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        import ctypes as C
        x, ws, *Ws = ctx.saved_tensors
        handle, dims, acts = ctx.handle, ctx.dims, ctx.acts
        n = len(acts)
        Q = x.shape[0]
        g = grad_out.contiguous().float()
        # zero-filled: with the handle option "mlp_grad_accumulate" (set by KnodeTrainer) the library adds into them
        dW = [torch.zeros_like(w) for w in Ws]
        db = [torch.zeros(w.shape[0], dtype=torch.float32, device=x.device) for w in Ws]
        dims_c = (C.c_int32 * (n + 1))(*dims)
        acts_c = (C.c_int32 * n)(*acts)
        Wp = (C.c_void_p * n)(*[w.data_ptr() for w in Ws])
        dWp = (C.c_void_p * n)(*[w.data_ptr() for w in dW])
        dbp = (C.c_void_p * n)(*[b.data_ptr() for b in db])
        kn.check(handle.lib.kr_mlp_backward(handle._h, Q, n, dims_c, acts_c, Wp, kn._ptr(x), x.shape[1], kn._ptr(g),
                                            kn._ptr(ws), dWp, dbp, kn._stream()))
        grads = []
        for a, b in zip(dW, db):
            grads += [a, b]
        dx = None
        if ctx.needs_input_grad[1]:
            # Gradient with respect to the input rows - no caller of the reference asks for it (its training loops feed
            # data), so there is no kernel for it: the layer stack once more as library GEMMs under autograd.
            with torch.enable_grad():
                xin = x[:, :dims[0]].detach().requires_grad_(True)
                a_ = xin
                for k in range(n):
                    a_ = _ACT_FN[acts[k]](a_ @ Ws[k].t() + ctx.biases[k])
                dxi, = torch.autograd.grad(a_, xin, g[:, :dims[-1]])
            dx = torch.zeros_like(x)
            dx[:, :dims[0]] = dxi
        return (None, dx, None, None, *grads)


class _OdePhysicsFunction(torch.autograd.Function):
    """Physics part of ``ODE_parallel`` with gradients into its inputs (cosserat_ode_torch.py:264-306 lets autograd
    flow there; no caller of the reference uses it).  Forward: ``kr_ode_batch``.  Backward: ``kr_ode_vjp_batch`` - the
    vector-Jacobian product of the point map by fp64 forward-mode differentiation on the device (uncut: ODE_parallel
    builds R and the quaternion-rate matrix with torch.stack, so its graph is the complete derivative)."""

    @staticmethod
    def forward(ctx, rod, ys, yhs, zhs, tf):
        h = rod._native()
        c = lambda t: t.detach().float().contiguous()
        dys, z = h.ode_batch(c(ys), c(yhs), c(zhs), c(tf), use_nn=False)
        ctx.rod = rod
        ctx.save_for_backward(c(ys), c(yhs), c(zhs), c(tf))
        return dys, z

    @staticmethod
    def backward(ctx, g_dys, g_z):
        h = ctx.rod._native()
        ys, yhs, zhs, tf = ctx.saved_tensors
        zero = lambda g, n: torch.zeros((ys.shape[0], n), dtype=torch.float32, device=ys.device) if g is None else g.float().contiguous()
        need = tuple(bool(n) for n in ctx.needs_input_grad[1:])
        grads = h.ode_vjp(ys, yhs, zhs, tf, zero(g_dys, 19), zero(g_z, 6), cut=False, need=need)
        return (None, *grads)


class _SweepFunction(torch.autograd.Function):
    """``CosseratRodTorch.getResidualEuler`` with autograd (cosserat_ode_torch.py:325-367 builds the graph op by op).

    Forward: the shooting-residual kernel (``kr_residual_batch``).  Backward: the discrete adjoint of the Euler sweep,
        lam_j = dL/dy_j = g_full[:19, j] + lam_{j+1} + J_j^T [ds lam_{j+1}; g_full[19:, j+1]],
    where J_j (25 x 19) is the Jacobian of one grid point's map y_j -> (y_s, z) including the network correction:
    the physics part from ``kr_ode_jacobian_batch`` (all N - 1 points at once), the network's input Jacobian from forward
    differences of the device MLP (``kr_mlp_eval_batch``, fp64).  dL/dG = lam_0[7:13]; the parameter gradients are
    ``kr_mlp_backward`` on the rows x_j = [y_j, z_j before the correction, tendon force] with output gradients
    [ds lam_{j+1}; g_full[19:, j+1]].  Inputs other than G and the parameters (history, tensions) get no gradient.
    By default the adjoint follows the reference's GRAPH, which differs from the function it computes in two places
    (include/knode_rod.h, kr_ode_vjp_batch: ``cut``); ``rod.exact_sweep_gradient = True`` gives the true gradient instead."""

    @staticmethod
    def forward(ctx, rod, G, *params):
        total, full, r = rod._sweep_forward(G)
        ctx.rod = rod
        ctx.n_params = len(params)
        ctx.save_for_backward(rod.y.detach().clone(), r.detach().clone(), *[p.detach() for p in params])
        ctx.hist = (rod.residualArgs["yh"].detach().clone(), rod.residualArgs["zh"].detach().clone(),
                    rod.tendon_tensions.detach().clone())
        return total, full

    @staticmethod
    def backward(ctx, g_total, g_full):
        rod = ctx.rod
        y, r, *params = ctx.saved_tensors
        yh, zh, tens = ctx.hist
        h = rod._native()
        dev = y.device
        N = int(rod.N)
        ds = float(h.derived().ds)
        f64 = torch.float64
        use_nn = bool(rod.use_nn)
        if use_nn:
            struct = mlp_structure(rod.nn_models)
            h.set_mlp([p.cpu().numpy() for p in params[0::2]], [p.cpu().numpy() for p in params[1::2]],
                      [a for _, a in struct])
        if g_full is None:
            g_full = torch.zeros((25, N), dtype=torch.float32, device=dev)
        if g_total is None:
            g_total = torch.zeros((), dtype=torch.float32, device=dev)
        gf = g_full.to(f64)
        # ---- Jacobians of the N - 1 grid-point maps y_j -> (y_s, z), [Q, 25, 19] ----
        Q = N - 1
        yq = y[:, :Q].t().to(f64).contiguous()   # [Q, 19]
        yhq = yh.float().to(dev)[:, :Q].t().to(f64).contiguous()
        zhq = zh.float().to(dev)[:, :Q].t().to(f64).contiguous()
        tdirs = torch.as_tensor(np.asarray(rod.tendon_dirs.detach().cpu() if torch.is_tensor(rod.tendon_dirs)
                                           else rod.tendon_dirs), dtype=f64, device=dev).reshape(4, 3)
        tf = (tens.to(dev).to(f64).reshape(1, 4) @ tdirs).expand(Q, 3).contiguous()
        yq = yq.detach()
        # physics Jacobian of every grid point on the device (kr_ode_jacobian_batch, fp64 forward mode); z of the physics alone
        J = h.ode_jacobian(yq, yhq, zhq, tf, cut=not getattr(rod, "exact_sweep_gradient", False))
        _, z_p = h.ode_batch(yq, yhq, zhq, tf, use_nn=False)
        if use_nn:
            # correction added after the physics (cosserat_ode_torch.py:192-213): out = MLP([y, z, f_tendon]).  Its
            # input Jacobian from forward differences of the device MLP (fp64), chained with dz/dy of the physics
            hist_in = params[0].shape[1] != 28
            x0 = torch.cat([yq, yhq, z_p, zhq, tf], dim=1) if hist_in else torch.cat([yq, z_p, tf], dim=1)
            n_in = x0.shape[1]
            stepx = 1e-7 * torch.clamp(x0.abs(), min=1.0)
            xp = x0.unsqueeze(1).repeat(1, n_in + 1, 1)
            ii = torch.arange(n_in, device=dev)
            xp[:, 1 + ii, ii] += stepx
            o = h.mlp_eval(xp.reshape(Q * (n_in + 1), n_in).contiguous()).reshape(Q, n_in + 1, 25)
            Jx = (o[:, 1:, :] - o[:, :1, :]) / stepx.unsqueeze(2)           # [Q, in, 25]
            dxdy = torch.zeros((Q, n_in, 19), dtype=f64, device=dev)
            dxdy[:, :19, :] = torch.eye(19, dtype=f64, device=dev)
            z0 = 38 if hist_in else 19
            dxdy[:, z0:z0 + 6, :] = J[:, 19:25, :]
            J = J + torch.einsum("qio,qid->qod", Jx, dxdy)
        # ---- adjoint recursion ----
        lam = gf[:19, N - 1].clone()
        lam[7:13] += -2.0 * g_total.to(f64) * r.to(f64)         # total = sum((F_tip - n_L)^2 + (M_tip - m_L)^2)
        gout = torch.zeros((Q, 25), dtype=f64, device=dev)
        for j in range(N - 2, -1, -1):
            gout[j, :19] = ds * lam
            gout[j, 19:] = gf[19:, j + 1]
            lam = gf[:19, j] + lam + J[j].t() @ gout[j]
        dG = lam[7:13].float()
        grads = [None] * ctx.n_params
        if use_nn and ctx.n_params:
            # rows of the network: [y_j, z_j of the physics alone, tendon force]
            in_dim = n_in
            x = torch.zeros((Q, (in_dim + 31) // 32 * 32), dtype=torch.float32, device=dev)
            x[:, :in_dim] = x0.float()
            dims = tuple([in_dim] + [p.shape[0] for p in params[0::2]])
            acts = tuple(a for _, a in struct)
            leaves = [p.clone().requires_grad_(True) for p in params]
            with torch.enable_grad():
                out = _MlpFunction.apply(h, x, dims, acts, *leaves)
            g32 = torch.zeros_like(out)
            g32[:, :25] = gout.float()
            grads = list(torch.autograd.grad(out, leaves, grad_outputs=g32))
        return (None, dG, *grads)


class CosseratRodTorch:
    def __init__(self, device, n_layers, nn_input_history=False):
        self.device = device
        self.use_nn = True
        self.nn_input_history = nn_input_history
        self.verbose = False
        self.y = None
        self.z = None
        # independent parameters, reference defaults (cosserat_ode_torch.py:14-45)
        self.L = 0.4
        self.N = 10
        self.E = 109e9
        self.r = 0.0012
        self.rho = 8000.
        dev = self.device
        self.vstar = torch.tensor([0., 0., 1.], device=dev)
        self.g = torch.tensor([0, 0, -9.81], device=dev)
        self.Bse = torch.zeros((3, 3), device=dev)
        self.Bbt = torch.diag(torch.tensor([3e-2, 3e-2, 3e-2], device=dev))
        self.C = torch.tensor([1e-4, 1e-4, 1e-4], device=dev)
        self.del_t = 0.005
        self.F_tip = torch.zeros(3, device=dev)
        self.M_tip = torch.zeros(3, device=dev)
        self.T0 = 5
        self.n_tendons = 4
        self.tendon_tensions = None
        self.tendon_offset = 0.02
        th = np.pi / self.n_tendons
        self.tendon_dirs = torch.tensor(
            [[np.cos(th + k * np.pi / 2), np.sin(th + k * np.pi / 2), 0.0] for k in range(4)],
            dtype=torch.float32).to(dev)
        self.p0 = torch.zeros(3, device=dev)
        self.h0 = torch.tensor([1., 0., 0., 0.], device=dev)
        self.q0 = torch.zeros(3, device=dev)
        self.w0 = torch.zeros(3, device=dev)
        self._handle = None
        self.compute_intermediate_terms()
        self.residualArgs = {"yh": None, "zh": None, "tendon_forces": None}

        # residual MLP, cosserat_ode_torch.py:60-88: Linear -> ELU -> Linear, weights |N(0.01, 0.01)|,
        # biases N(0, 0.01)
        width = int(n_layers)
        self.layers = [nn.Linear(53 if self.nn_input_history else 28, width), nn.ELU(), nn.Linear(width, 25)]
        for layer in self.layers:
            if isinstance(layer, nn.Linear):
                self.non_negative_normal_init(layer, mean=0.01, std=0.01)
                nn.init.normal_(layer.bias, mean=0.0, std=0.01)
        self.nn_models = nn.ModuleList(self.layers).to(self.device)

    def non_negative_normal_init(self, m, mean, std):
        """cosserat_ode_torch.py:90-105"""
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            assert mean >= 0, "Mean must be non-negative"
            with torch.no_grad():
                m.weight.data.normal_(mean, std).abs_()

    # ------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------
    def _np(self, a):
        return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)

    def _params(self) -> kn.KrParams:
        f = self._np
        return kn.params_from_dict(dict(
            L=float(self.L), N=int(self.N), nn_input_history=int(bool(self.nn_input_history)), E=float(self.E),
            r=float(self.r), rho=float(self.rho), vstar=f(self.vstar), g=f(self.g), Bse=f(self.Bse), Bbt=f(self.Bbt),
            C=f(self.C), del_t=float(self.del_t), F_tip=f(self.F_tip), M_tip=f(self.M_tip),
            tendon_dirs=f(self.tendon_dirs), p0=f(self.p0), h0=f(self.h0), q0=f(self.q0), w0=f(self.w0)))

    def compute_intermediate_terms(self):
        """cosserat_ode_torch.py:108-129; the dependent terms come from kr_derive."""
        d = kn.derive(self._params())
        dev = self.device
        m3 = lambda a: torch.tensor(np.array(a, dtype=np.float64).reshape(3, 3), dtype=torch.float32, device=dev)
        self.A, self.G, self.ds = d.A, d.G, d.ds
        self.J, self.Kse, self.Kbt = m3(d.J), m3(d.Kse), m3(d.Kbt)
        self.c0, self.c1, self.c2 = d.c0, d.c1, d.c2
        self.Kse_plus_c0_Bse_inv = m3(d.Kse_plus_c0_Bse_inv)
        self.Kbt_plus_c0_Bbt_inv = m3(d.Kbt_plus_c0_Bbt_inv)
        self.Kse_vstar = torch.tensor(list(d.Kse_vstar), dtype=torch.float32, device=dev)
        self.rhoA = d.rhoA
        self.rhoAg = torch.tensor(list(d.rhoAg), dtype=torch.float32, device=dev)
        self.rhoJ = m3(d.rhoJ)

    def _device_index(self) -> int:
        d = torch.device(self.device)
        if d.type != "cuda":
            raise kn.KrError(f"CosseratRodTorch computes on the MI355X only; device={self.device!r} has no kernels")
        return d.index if d.index is not None else torch.cuda.current_device()

    def _native(self) -> kn.Handle:
        p = self._params()
        if self._handle is None:
            self._handle = kn.Handle(p, self._device_index())
        else:
            self._handle.set_params(p)
        return self._handle

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_handle"] = None
        return st

    def __setstate__(self, st):
        """Also adopts the attribute dictionary of a robot pickled by the reference's class of the same
        name (physics_train.py:284-288): same attribute names, no native handle."""
        self.__dict__.update(st)
        self._handle = None
        self.__dict__.setdefault("nn_input_history", False)
        self.__dict__.setdefault("verbose", False)
        self.__dict__.setdefault("residualArgs", {"yh": None, "zh": None, "tendon_forces": None})

    def to(self, device):
        """Moves every tensor attribute and the MLP to `device` (a checkpoint written on another machine)."""
        self.device = device
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        self.nn_models = self.nn_models.to(device)
        if isinstance(self.residualArgs, dict):
            self.residualArgs = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.residualArgs.items()}
        self._handle = None
        return self

    # ------------------------------------------------------------------
    # MLP
    # ------------------------------------------------------------------
    def _mlp(self, x_padded):
        """out[Q, 25] for padded input rows x[Q, in_pad] (HIP GEMMs, differentiable w.r.t. the parameters)."""
        struct = mlp_structure(self.nn_models)
        dims = [struct[0][0].in_features] + [l.out_features for l, _ in struct]
        acts = [a for _, a in struct]
        params = []
        for l, _ in struct:
            if l.bias is None:
                raise kn.KrError("Linear layers of the residual MLP need a bias")
            params += [l.weight, l.bias]
        out = _MlpFunction.apply(self._native(), x_padded, tuple(dims), tuple(acts), *params)
        return out[:, :dims[-1]]

    def forward(self, x):
        """cosserat_ode_torch.py:131-134; accepts [in] or [Q, in]."""
        single = x.dim() == 1
        xx = x.reshape(1, -1) if single else x
        in_dim = xx.shape[1]
        pad = (in_dim + 31) // 32 * 32
        xp = torch.zeros((xx.shape[0], pad), dtype=torch.float32, device=xx.device)
        xp[:, :in_dim] = xx.float()
        out = self._mlp(xp)
        return out[0] if single else out

    # ------------------------------------------------------------------
    # physics
    # ------------------------------------------------------------------
    def ODE_parallel(self, ys, yhs, zhs, tendon_forcess):
        """Batched arc-length derivative, cosserat_ode_torch.py:217-322:
        [Q,19],[Q,19],[Q,6],[Q,3] -> (dys[Q,19], z[Q,6])."""
        h = self._native()
        if torch.is_grad_enabled() and any(t.requires_grad for t in (ys, yhs, zhs, tendon_forcess)):
            # gradients into the inputs requested (no reference caller does): same kernels forward, see the Functions
            y_, yh_, zh_, tf_ = (t.float() for t in (ys, yhs, zhs, tendon_forcess))
            dys, z = _OdePhysicsFunction.apply(self, y_, yh_, zh_, tf_)
        else:
            c = lambda t: t.detach().float().contiguous()
            y_, yh_, zh_, tf_ = c(ys), c(yhs), c(zhs), c(tendon_forcess)
            dys, z = h.ode_batch(y_, yh_, zh_, tf_, use_nn=False)
        if self.use_nn:
            parts = [y_, yh_, z, zh_, tf_] if self.nn_input_history else [y_, z, tf_]
            x = torch.cat(parts, dim=1)
            out = self.forward(x)
            dys = dys + out[:, :19]
            z = z + out[:, 19:]
        return dys, z

    def ODE(self, y, yh, zh, tendon_forces):
        """Single grid point, cosserat_ode_torch.py:137-214.  (Values as the reference; gradients into the inputs, when
        asked for, are those of ODE_parallel - the reference's single-point graph is cut at R(h) and at the
        quaternion-rate matrix, which only ``getResidualEuler`` reproduces: see kr_ode_vjp_batch (``cut``) in include/knode_rod.h.)"""
        dys, z = self.ODE_parallel(y.reshape(1, 19), yh.reshape(1, 19), zh.reshape(1, 6), tendon_forces.reshape(1, 3))
        return dys[0], z[0]

    def _next_segment(self, Gs, yhs, zhs, tensions, idx):
        """pred[S, K, 25] for key columns idx (1-based columns of the output, see
        cosserat_ode_torch.py:412)."""
        h = self._native()
        S, K = Gs.shape[0], len(idx)
        dev = Gs.device
        c = lambda t: t.detach().float().contiguous()
        idx_t = torch.as_tensor(np.asarray(idx, dtype=np.int32), device=dev)
        in_dim = 53 if self.nn_input_history else 28
        pad = (in_dim + 31) // 32 * 32
        x = torch.empty((S * K, pad), dtype=torch.float32, device=dev)
        base = torch.empty((S * K, 25), dtype=torch.float32, device=dev)
        kn.check(h.lib.kr_next_segment_physics(h._h, S, K, kn._ptr(c(Gs)), kn._ptr(c(yhs)), kn._ptr(c(zhs)),
                                               kn._ptr(c(tensions)), kn._ptr(idx_t), kn._ptr(x), pad, kn._ptr(base),
                                               kn.KR_F32, kn._stream()))
        if not self.use_nn:
            return base.reshape(S, K, 25)
        out = self._mlp(x)
        pred = base + torch.cat([self.ds * out[:, :19], out[:, 19:]], dim=1)
        return pred.reshape(S, K, 25)

    def parallelGetNextSegmentEuler(self, Gs, segment_idxs, args):
        """cosserat_ode_torch.py:401-437: Gs[S,25,N], key columns -> [S,25,K]."""
        pred = self._next_segment(Gs, args["yh"], args["zh"], args["tendon_tensions"], np.asarray(segment_idxs))
        return pred.transpose(1, 2)

    def getNextSegmentEuler(self, G):
        """cosserat_ode_torch.py:370-399: teacher-forced one-step predictor for every segment,
        G[25,N] -> full_rod[25,N] (column 0 is the guess itself)."""
        N = int(self.N)
        yh, zh = self.residualArgs["yh"], self.residualArgs["zh"]
        pred = self._next_segment(G.reshape(1, 25, N), yh.reshape(1, 19, N), zh.reshape(1, 6, N),
                                  self.tendon_tensions.reshape(1, 4), np.arange(1, N))
        first = G.detach().float()[:, :1]
        return torch.cat([first, pred[0].transpose(0, 1)], dim=1)

    def getResidualEuler(self, G):
        """cosserat_ode_torch.py:325-367: one full shooting sweep from the guessed base wrench
        G[6] -> (sum of squared tip residuals, full_rod[25,N]).  Differentiable with respect to G and the MLP
        parameters (what the reference's autograd graph reaches from a caller's point of view): forward on the
        shooting kernel, backward as an adjoint sweep (``_SweepFunction``)."""
        params = []
        if self.use_nn:
            for l, _ in mlp_structure(self.nn_models):
                params += [l.weight, l.bias]
        total, full = _SweepFunction.apply(self, G, *params)
        return total, full

    def _sweep_forward(self, G):
        h = self._native()
        N = int(self.N)
        dev = G.device
        c = lambda t: t.detach().float().contiguous()
        hist = h.pack(c(self.residualArgs["yh"]).reshape(1, 19, N), c(self.residualArgs["zh"]).reshape(1, 6, N))
        nxt = h.new_state(1, torch.float32)
        tens = c(self.tendon_tensions).reshape(1, 4)
        use_nn = bool(self.use_nn)
        if use_nn:
            struct = mlp_structure(self.nn_models)
            h.set_mlp([l.weight.detach().cpu().numpy() for l, _ in struct],
                      [l.bias.detach().cpu().numpy() for l, _ in struct], [a for _, a in struct])
        r = h.residual(c(G).reshape(1, 6), None, hist, nxt, tens, use_nn=use_nn, hist_is_explicit=True)
        y_new, z_new = h.unpack(nxt)
        # column layout of the reference's full_rod: column 0 = [y0; z[:,0] of the caller], column j+1 = [y_{j+1}; z_j]
        z_in = c(self.z) if self.z is not None else torch.zeros((6, N), device=dev)
        full = torch.cat([torch.cat([y_new[0, :, :1], z_in[:, :1]], dim=0),
                          torch.cat([y_new[0, :, 1:], z_new[0, :, : N - 1]], dim=0)], dim=1)
        self.y = y_new[0]
        return torch.sum(r[0] ** 2), full, r[0]
