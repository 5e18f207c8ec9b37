"""Fused KNODE training step and its data-parallel form.

Restates the epoch body of ``physics_train.py`` (slow loop :209-304, ``--fast``
loop :306-408) on top of the HIP kernels:

    for every trajectory and every window step t (29 of them):
        pred = one-step-ahead predictor at the key points      (a12 / a13)
        loss += MSE(p) + MSE(n,m,q,w) + MSE(euler(h)) + MSE(z vs column key-1)
    loss /= 29;  backward;  Adam(lr=1e-2);  ReduceLROnPlateau;  clamp weights >= 0

Two facts of the reference make this cheap on a GPU:
  * every (trajectory, t, key point) row is independent - one batched launch;
  * the physics part of the predictor does not depend on the MLP parameters,
    so the MLP input rows ``x`` and the parameter-free part of the prediction
    ``base`` are computed once per data set, not once per epoch.

Per epoch the device work is: MLP forward (MFMA GEMMs) -> fused prediction +
loss + d loss/d out -> MLP backward (MFMA GEMMs) -> one all-reduce of the flat
gradient buffer (RCCL over xGMI, data parallel only) -> Adam + clamp.

Both reference loops reduce to this with different key points: the slow loop
evaluates all segments but scores columns [2, 6, 9] (``batch_idx`` at :220 is
just ``stp_idx``), the fast loop evaluates and scores [3, 5, 7, 9].
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

import krod_native as kn
from cosserat_ode_torch import mlp_structure


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced split of n items: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatBucket:
    """One flat fp32 buffer holding every parameter gradient plus one trailing
    slot for the loss, so a training step needs exactly one all-reduce."""

    def __init__(self, shapes, device):
        self.shapes = [tuple(s) for s in shapes]
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.flat = torch.zeros(sum(self.sizes) + 1, dtype=torch.float32, device=device)
        self.views, off = [], 0
        for s, n in zip(self.shapes, self.sizes):
            self.views.append(self.flat[off:off + n].view(s))
            off += n
        self.loss = self.flat[off:off + 1]

    def all_reduce(self, group=None):
        """group=None: the default process group; group=False: never reduce (a single-rank run inside a
        multi-rank job)."""
        import torch.distributed as dist
        if group is False:
            return False
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            return True
        return False


class DevicePlateau:
    """``torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'min', patience, factor)`` whose state lives in device
    memory and is advanced by ``kr_adam_plateau_step`` inside the optimizer launch (physics_train.py:206,297).  The
    methods the training drivers use are kept: ``get_last_lr()`` (a device read: call it when printing, not every
    epoch), ``state_dict()`` / ``load_state_dict()``."""

    def __init__(self, lr, patience, factor, device, threshold=1e-4, min_lr=0.0):
        self.patience, self.factor, self.threshold, self.min_lr = int(patience), float(factor), float(threshold), float(min_lr)
        self.buf = torch.tensor([lr, lr, float("inf"), 0.0, 0.0, 0.0], dtype=torch.float64, device=device)
        self.steps = 0

    def _lr_index(self):
        return self.steps & 1  # the rate the NEXT step will use (= what torch reports after scheduler.step())

    def get_last_lr(self):
        return [float(self.buf[self._lr_index()].item())]

    def set_lr(self, lr):
        self.buf[0:2] = float(lr)

    def state_dict(self):
        b = self.buf.cpu().tolist()
        return {"lr": b[self._lr_index()], "best": b[2], "num_bad_epochs": int(b[3]), "last_loss": b[4],
                "reductions": int(b[5]), "patience": self.patience, "factor": self.factor, "threshold": self.threshold,
                "min_lr": self.min_lr}

    def load_state_dict(self, sd):
        self.buf[0:2] = float(sd["lr"])
        self.buf[2] = float(sd.get("best", float("inf")))
        self.buf[3] = float(sd.get("num_bad_epochs", 0))


class KnodeTrainer:
    """One-step-ahead KNODE training on a fixed set of trajectories.

    trajs    float32 [M, T, 25, N]  reference trajectories (``simulate(...)[:, :25]``)
    controls float32 [M, T, 4]
    Each rank passes its own shard of the M trajectories (see ``shard_range``);
    gradients and the loss are summed over ranks, which reproduces the
    single-process result of the reference exactly (its loss is a sum over
    trajectories, physics_train.py:215-267).
    """

    def __init__(self, robot, trajs, controls, key_pt_idx, lr=1e-2, weight_decay=0.0, clamp_weights=True,
                 patience=80, factor=0.5, group=None, keep_pred=False, native_adam=True, device_plateau=None):
        self.robot = robot
        self.native_adam = native_adam  # Adam + clamp + gradient zeroing as ONE kernel (kr_adam_step)
        # learning-rate schedule (ReduceLROnPlateau) advanced on the device inside the optimizer launch: an epoch
        # then needs no host round trip at all (step(sync_loss=False)); default with the fused optimizer
        self.device_plateau = native_adam if device_plateau is None else (bool(device_plateau) and native_adam)
        # keep_pred: write the predictions of every epoch (two kernels: forward, loss); otherwise the loss runs in the
        # epilogue of the forward kernel and predictions() evaluates them on demand
        self.keep_pred = keep_pred
        self.group = group
        self.fused_epoch = True   # kr_train_epoch (one call per epoch) where the network is one the fused kernels serve
        # the first epoch always packs: the handle judges its fragment copies by pointer identity, and torch's caching
        # allocator hands a new trainer the addresses of a freed one
        self._repack = True
        self._param_versions = None
        self.time_allreduce = False
        self.allreduce_events = []
        self.clamp_weights = clamp_weights
        h = robot._native()
        self.h = h
        dev = trajs.device
        trajs = trajs.float().contiguous()
        controls = controls.float().contiguous()
        M, T, _, N = trajs.shape
        assert N == int(robot.N)
        self.M, self.T, self.N = M, T, N
        self.steps = T - 1  # batch_len - 1 = 29 in the reference
        self.idx = np.asarray(key_pt_idx, dtype=np.int32)
        self.K = len(self.idx)
        self.idx_t = torch.as_tensor(self.idx, device=dev)
        S = M * (T - 1)
        self.S = S
        # physics_train.py:318-333: states t = 0..T-2, previous state (first one repeated), next state as guess
        ys, zs = trajs[:, : T - 1, :19], trajs[:, : T - 1, 19:]
        y_prev = torch.cat([ys[:, :1], ys[:, :-1]], dim=1)
        z_prev = torch.cat([zs[:, :1], zs[:, :-1]], dim=1)
        yh = (robot.c1 * ys + robot.c2 * y_prev).reshape(S, 19, N).contiguous()
        zh = (robot.c1 * zs + robot.c2 * z_prev).reshape(S, 6, N).contiguous()
        self.target = trajs[:, 1:T].reshape(S, 25, N).contiguous()
        tens = controls[:, : T - 1].reshape(S, 4).contiguous()
        in_dim = 53 if robot.nn_input_history else 28
        self.in_pad = (in_dim + 31) // 32 * 32
        Q = S * self.K
        self.Q = Q
        self.x = torch.empty((Q, self.in_pad), dtype=torch.float32, device=dev)
        self.base = torch.empty((Q, 25), dtype=torch.float32, device=dev)
        # target values of every scored row, gathered once (they do not change between epochs)
        self.target_rows = torch.empty((max(Q, 1), 25), dtype=torch.float32, device=dev)
        if Q:
            kn.check(h.lib.kr_gather_targets(h._h, S, self.K, kn._ptr(self.target), kn._ptr(self.idx_t),
                                             kn._ptr(self.target_rows), kn._stream()))
        if Q:
            kn.check(h.lib.kr_next_segment_physics(h._h, S, self.K, kn._ptr(self.target), kn._ptr(yh), kn._ptr(zh),
                                                   kn._ptr(tens), kn._ptr(self.idx_t), kn._ptr(self.x), self.in_pad,
                                                   kn._ptr(self.base), kn.KR_F32, kn._stream()))
        # MLP description
        self.struct = mlp_structure(robot.nn_models)
        self.n = len(self.struct)
        dims = [self.struct[0][0].in_features] + [l.out_features for l, _ in self.struct]
        self.dims_c = (C.c_int32 * (self.n + 1))(*dims)
        self.acts_c = (C.c_int32 * self.n)(*[a for _, a in self.struct])
        self.params = []
        for l, _ in self.struct:
            self.params += [l.weight, l.bias]
        self.bucket = FlatBucket([p.shape for p in self.params], dev)
        for p, v in zip(self.params, self.bucket.views):
            p.grad = v  # Adam reads the all-reduced gradients straight from the flat buffer
        ws_bytes = h.lib.kr_mlp_ws_bytes(self.n, self.dims_c, max(Q, 1))
        self.ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        self.out = torch.zeros((max(Q, 1), 32), dtype=torch.float32, device=dev)
        self.dout = torch.zeros((max(Q, 1), 32), dtype=torch.float32, device=dev)
        self.pred = torch.zeros((max(Q, 1), 25), dtype=torch.float32, device=dev)
        self.weight_decay = float(weight_decay)
        if native_adam:
            # parameters become views of one flat buffer (same layout as the gradient bucket), Adam's moments
            # live beside it; torch keeps only the learning-rate schedule (ReduceLROnPlateau on a stand-in group)
            n = sum(self.bucket.sizes)
            self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
            self.lower = torch.full((n,), float("-inf"), dtype=torch.float32, device=dev)
            off = 0
            for k, (p, sz) in enumerate(zip(self.params, self.bucket.sizes)):
                view = self.flat_p[off:off + sz].view(p.shape)
                view.copy_(p.data)
                p.data = view
                if clamp_weights and k % 2 == 0:
                    self.lower[off:off + sz] = 0.0
                off += sz
            self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
            self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
            self.adam_step = 0
            self.betas, self.adam_eps = (0.9, 0.999), 1e-8
            self._lr_holder = torch.nn.Parameter(torch.zeros(1, device=dev))
            self.optimizer = torch.optim.SGD([self._lr_holder], lr=lr)  # carries lr for the scheduler only
            h.set_option("mlp_grad_accumulate", 1)
            self.bucket.flat.zero_()
        else:
            self.optimizer = torch.optim.Adam(self.params, lr=lr, weight_decay=weight_decay)
            # the option lives on the robot's (shared) handle: an earlier native-Adam trainer may have left it on,
            # and this branch relies on the library zeroing dW / db / loss itself
            h.set_option("mlp_grad_accumulate", 0)
        if self.device_plateau:
            self.scheduler = DevicePlateau(lr, patience, factor, dev)
            self.loss_log = torch.zeros(4096, dtype=torch.float32, device=dev)  # loss of every epoch (grown on demand)
        else:
            self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, "min", patience=patience,
                                                                        factor=factor)

    def _ptr_arrays(self):
        n = self.n
        Wp = (C.c_void_p * n)(*[self.params[2 * k].data_ptr() for k in range(n)])
        bp = (C.c_void_p * n)(*[self.params[2 * k + 1].data_ptr() for k in range(n)])
        dWp = (C.c_void_p * n)(*[self.bucket.views[2 * k].data_ptr() for k in range(n)])
        dbp = (C.c_void_p * n)(*[self.bucket.views[2 * k + 1].data_ptr() for k in range(n)])
        return Wp, bp, dWp, dbp

    def loss_and_grads(self):
        """Forward + loss + backward + gradient all-reduce.  Leaves the summed
        gradients in ``p.grad`` and returns the (device) loss scalar."""
        h, s = self.h, kn._stream()
        Wp, bp, dWp, dbp = self._ptr_arrays()
        Q = self.Q
        if self.keep_pred:
            kn.check(h.lib.kr_mlp_forward(h._h, Q, self.n, self.dims_c, self.acts_c, Wp, bp, kn._ptr(self.x), self.in_pad,
                                          kn._ptr(self.out), kn._ptr(self.ws), s))
            kn.check(h.lib.kr_loss_rows_fwd_bwd(h._h, self.S, self.K, kn._ptr(self.base), kn._ptr(self.out),
                                                kn._ptr(self.target_rows), float(self.steps), kn._ptr(self.pred),
                                                kn._ptr(self.bucket.loss), kn._ptr(self.dout), s))
        else:  # the loss runs in the epilogue of the forward kernel where the fused kernels serve the network
            kn.check(h.lib.kr_mlp_forward_loss(h._h, self.S, self.K, self.n, self.dims_c, self.acts_c, Wp, bp,
                                               kn._ptr(self.x), self.in_pad, kn._ptr(self.base), kn._ptr(self.target_rows),
                                               float(self.steps), kn._ptr(self.out), kn._ptr(self.bucket.loss),
                                               kn._ptr(self.dout), kn._ptr(self.ws), s))
        kn.check(h.lib.kr_mlp_backward(h._h, Q, self.n, self.dims_c, self.acts_c, Wp, kn._ptr(self.x), self.in_pad,
                                       kn._ptr(self.dout), kn._ptr(self.ws), dWp, dbp, s))
        if self.time_allreduce:  # (bench.py's data-parallel leg: HIP events around the collective, on the compute stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if self.bucket.all_reduce(self.group):
                e1.record()
                self.allreduce_events.append((e0, e1))
        else:
            self.bucket.all_reduce(self.group)
        return self.bucket.loss

    def apply_update(self):
        """Adam + clamp on the gradients loss_and_grads() left in the bucket (physics_train.py:289-304)."""
        if self.native_adam and self.device_plateau:
            h = self.h
            self.adam_step += 1
            n = self.flat_p.numel()
            e = self.scheduler.steps
            if e >= self.loss_log.numel():
                self.loss_log = torch.cat([self.loss_log, torch.zeros_like(self.loss_log)])
            sc = self.scheduler
            kn.check(h.lib.kr_adam_plateau_step(
                h._h, n, kn._ptr(self.flat_p), kn._ptr(self.bucket.flat), kn._ptr(self.exp_avg), kn._ptr(self.exp_avg_sq),
                kn._ptr(self.lower) if self.clamp_weights else None, kn._ptr(sc.buf), self.betas[0], self.betas[1],
                self.adam_eps, self.weight_decay, self.adam_step, n + 1, n, sc.factor, sc.patience, sc.threshold,
                sc.min_lr, self.loss_log.data_ptr() + 4 * e, kn._stream()))
            # the kernel reads sched[(adam_step - 1) & 1]; keep the scheduler's parity in step with Adam's
            sc.steps = self.adam_step
        elif self.native_adam:
            h = self.h
            self.adam_step += 1
            n = self.flat_p.numel()
            kn.check(h.lib.kr_adam_step(h._h, n, kn._ptr(self.flat_p), kn._ptr(self.bucket.flat), kn._ptr(self.exp_avg),
                                        kn._ptr(self.exp_avg_sq), kn._ptr(self.lower) if self.clamp_weights else None,
                                        float(self.optimizer.param_groups[0]["lr"]), self.betas[0], self.betas[1],
                                        self.adam_eps, self.weight_decay, self.adam_step, n + 1, kn._stream()))
        else:
            self.optimizer.step()
            if self.clamp_weights:  # physics_train.py:299-304 - hits every weight matrix
                with torch.no_grad():
                    for k in range(self.n):
                        self.params[2 * k].clamp_(min=0)

    def _epoch_call(self, phase):
        """kr_train_epoch: the epoch (phase 0) or its two halves around the all-reduce (1, 2) as 3 kernel launches."""
        h, sc = self.h, self.scheduler
        e = sc.steps
        if e >= self.loss_log.numel():
            self.loss_log = torch.cat([self.loss_log, torch.zeros_like(self.loss_log)])
        kn.check(h.lib.kr_train_epoch(
            h._h, self.S, self.K, self.n, self.dims_c, self.acts_c, kn._ptr(self.flat_p), kn._ptr(self.bucket.flat),
            kn._ptr(self.exp_avg), kn._ptr(self.exp_avg_sq), kn._ptr(self.lower) if self.clamp_weights else None,
            kn._ptr(sc.buf), kn._ptr(self.x), self.in_pad, kn._ptr(self.base), kn._ptr(self.target_rows),
            float(self.steps), kn._ptr(self.dout), kn._ptr(self.ws), self.betas[0], self.betas[1], self.adam_eps,
            self.weight_decay, self.adam_step + 1, sc.factor, sc.patience, sc.threshold, sc.min_lr,
            self.loss_log.data_ptr() + 4 * e, phase, 1 if self._repack else 0, kn._stream()))
        if phase != 2:
            self._repack = False
        if phase != 1:
            self.adam_step += 1
            sc.steps = self.adam_step

    def weights_changed(self):
        """Tell the trainer that the parameters were written from outside (a loaded checkpoint): the next epoch packs the
        kernels' weight fragments afresh instead of relying on the copies its own updates maintain."""
        self._repack = True

    def _fused_epoch(self):
        """The single-call epoch where it applies; False = not served (the caller takes the separate calls)."""
        if not (self.fused_epoch and self.native_adam and self.device_plateau and not self.keep_pred and self.Q > 0):
            return False
        import torch.distributed as dist
        dp = (self.group is not False and dist.is_available() and dist.is_initialized()
              and dist.get_world_size(self.group) > 1)
        # in-place writes to the parameters from outside (load_state_dict, copy_, clamp_) bump torch's version counters;
        # the kernels' own updates do not
        ver = tuple(p._version for p in self.params) + (self.flat_p._version,)
        if ver != self._param_versions:
            self._repack = self._repack or self._param_versions is not None
            self._param_versions = ver
        try:
            self._epoch_call(1 if dp else 0)
        except kn.KrError as err:
            if err.code != kn.KR_E_UNSUPPORTED:
                raise
            self.fused_epoch = False  # a network the fused kernels do not serve
            return False
        if dp:
            if self.time_allreduce:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.bucket.all_reduce(self.group)
                e1.record()
                self.allreduce_events.append((e0, e1))
            else:
                self.bucket.all_reduce(self.group)
            self._epoch_call(2)
        return True

    def step(self, sync_loss=True):
        """One epoch of physics_train.py (:313-401): forward, loss, backward, all-reduce, Adam, plateau schedule, clamp.
        With the device-side schedule nothing in here waits for the GPU unless ``sync_loss`` asks for the loss as a
        float (``losses()`` returns the whole history later); otherwise the schedule is torch's and needs the value."""
        if self._fused_epoch():
            if sync_loss:
                return float(self.loss_log[self.scheduler.steps - 1].item())
            return None
        loss = self.loss_and_grads()
        if self.device_plateau:
            self.apply_update()  # (reads the loss slot on the device, logs it, steps the schedule, clears the slot)
            if sync_loss:
                return float(self.loss_log[self.scheduler.steps - 1].item())
            return None
        val = float(loss.item())  # before the update: the native Adam kernel also clears the loss slot
        self.apply_update()
        self.scheduler.step(val)
        return val

    def run(self, n_epochs):
        """n_epochs epochs queued by ONE library call (kr_train_epochs) where the single-call epoch applies and no
        all-reduce sits between its halves; otherwise n_epochs x step().  Nothing waits for the GPU; losses() has the
        curve.  The per-epoch host work of step() (Python + ctypes: ~135 us) is longer than the epoch on the GPU."""
        import torch.distributed as dist
        dp = (self.group is not False and dist.is_available() and dist.is_initialized()
              and dist.get_world_size(self.group) > 1)
        ok = (self.fused_epoch and self.native_adam and self.device_plateau and not self.keep_pred and self.Q > 0 and not dp)
        if not ok or n_epochs < 2:
            for _ in range(n_epochs):
                self.step(sync_loss=False)
            return
        if not self._fused_epoch():   # first epoch through step(): settles repack / version bookkeeping, may say "unsupported"
            self.step(sync_loss=False)
            for _ in range(n_epochs - 1):
                self.step(sync_loss=False)
            return
        n = n_epochs - 1
        h, sc = self.h, self.scheduler
        e = sc.steps
        while e + n > self.loss_log.numel():
            self.loss_log = torch.cat([self.loss_log, torch.zeros_like(self.loss_log)])
        kn.check(h.lib.kr_train_epochs(
            h._h, n, self.S, self.K, self.n, self.dims_c, self.acts_c, kn._ptr(self.flat_p), kn._ptr(self.bucket.flat),
            kn._ptr(self.exp_avg), kn._ptr(self.exp_avg_sq), kn._ptr(self.lower) if self.clamp_weights else None,
            kn._ptr(sc.buf), kn._ptr(self.x), self.in_pad, kn._ptr(self.base), kn._ptr(self.target_rows),
            float(self.steps), kn._ptr(self.dout), kn._ptr(self.ws), self.betas[0], self.betas[1], self.adam_eps,
            self.weight_decay, self.adam_step + 1, sc.factor, sc.patience, sc.threshold, sc.min_lr,
            self.loss_log.data_ptr() + 4 * e, 0, kn._stream()))
        self.adam_step += n
        sc.steps = self.adam_step

    def losses(self):
        """Loss of every epoch taken so far (one device read)."""
        if not self.device_plateau:
            raise kn.KrError("losses(): only kept with the device-side schedule")
        return self.loss_log[: self.scheduler.steps].cpu().tolist()

    def optimizer_state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (what physics_train.py:284-288 stores under 'optim'): per
        parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` plus one param group.  With the fused optimizer the moments
        are views into its flat buffers, copied out here."""
        if not self.native_adam:
            return self.optimizer.state_dict()
        state, off = {}, 0
        for k, (p, sz) in enumerate(zip(self.params, self.bucket.sizes)):
            if self.adam_step > 0:
                state[k] = {"step": torch.tensor(float(self.adam_step)),
                            "exp_avg": self.exp_avg[off:off + sz].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + sz].view(p.shape).clone()}
            off += sz
        lr_now = self.scheduler.get_last_lr()[0] if self.device_plateau else float(self.optimizer.param_groups[0]["lr"])
        group = {"lr": lr_now, "betas": tuple(self.betas), "eps": self.adam_eps,
                 "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """Resume from an Adam ``state_dict`` (ours or one written by the reference's torch.optim.Adam)."""
        if not self.native_adam:
            self.optimizer.load_state_dict(sd)
            return
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise kn.KrError("optimizer state does not match the network (number of parameters)")
        if group.get("amsgrad") or group.get("maximize"):
            raise kn.KrError("only plain Adam state can be resumed (amsgrad / maximize are not implemented)")
        self.betas, self.adam_eps = tuple(group["betas"]), float(group["eps"])
        self.weight_decay = float(group["weight_decay"])
        self.optimizer.param_groups[0]["lr"] = float(group["lr"])
        if self.device_plateau:
            self.scheduler.set_lr(float(group["lr"]))
        steps, off = set(), 0
        for k, (p, sz) in enumerate(zip(self.params, self.bucket.sizes)):
            st = sd["state"].get(group["params"][k])
            if st is None:
                self.exp_avg[off:off + sz].zero_()
                self.exp_avg_sq[off:off + sz].zero_()
                steps.add(0)
            else:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise kn.KrError(f"optimizer state of parameter {k} has shape {tuple(st['exp_avg'].shape)}")
                self.exp_avg[off:off + sz].copy_(st["exp_avg"].reshape(-1).to(self.exp_avg))
                self.exp_avg_sq[off:off + sz].copy_(st["exp_avg_sq"].reshape(-1).to(self.exp_avg_sq))
                steps.add(int(float(st["step"])))
            off += sz
        if len(steps) != 1:
            raise kn.KrError("per-parameter step counts differ: not a state the fused Adam can continue")
        self.adam_step = steps.pop()
        if self.device_plateau:
            self.scheduler.steps = self.adam_step  # (parity of the rate slot; the loss log restarts at this index)
            if self.adam_step >= self.loss_log.numel():
                self.loss_log = torch.zeros(2 * self.adam_step + 4096, dtype=torch.float32, device=self.loss_log.device)

    def predictions(self):
        """[S, 25, K] predictions with the current weights (reference layout of grow_trajs)."""
        if not self.keep_pred:
            h = self.h
            Wp, bp, _, _ = self._ptr_arrays()
            kn.check(h.lib.kr_mlp_forward(h._h, self.Q, self.n, self.dims_c, self.acts_c, Wp, bp, kn._ptr(self.x),
                                          self.in_pad, kn._ptr(self.out), kn._ptr(self.ws), kn._stream()))
            ds = float(h.derived().ds)
            self.pred[: self.Q] = self.base[: self.Q]
            self.pred[: self.Q, :19] += ds * self.out[: self.Q, :19]   # cosserat_ode_torch.py:386-393: y + ds ys, z as is
            self.pred[: self.Q, 19:] += self.out[: self.Q, 19:25]
        return self.pred[: self.Q].reshape(self.S, self.K, 25).transpose(1, 2)
