"""Shared pieces of the ``-m gpu`` parity tests: robot construction, MLP injection (what
physics_train.py:104-110 does) and the bookkeeping of WHICH step kernel a test exercised.

Three kernels serve one time step: single shooting (8 rods per wavefront, path 0),
multiple shooting (one rod per wavefront, one launch per step, path 1) and its persistent form (all steps of
kr_simulate_batch in one launch, path 2).  Tests that are parametrised over them call ``require_path`` first (it
SKIPS the parametrisation when the named kernel cannot serve the problem - the run would silently duplicate
another mode) and ``assert_path`` after the simulation (the handle reports what actually ran)."""
import numpy as np
import pytest

MODES = ["single", "multi", "persistent", "overlap"]
# "overlap": the persistent launch with the verifying sweep of step t folded into the Jacobian sweep of step t + 1
# (kr_mso_impl.hpp, the default where it applies); "persistent" pins the plain persistent kernel (KR_OVERLAP=0).
# Pseudo path 3 = path 2 with the overlapped kernel.
PATH_OF_MODE = {"single": 0, "multi": 1, "persistent": 2, "overlap": 3}
MS_P = 4            # sub-intervals of the multiple-shooting kernels (kr_ms_impl.hpp)
PERSIST_MAX_N = 128  # the persistent kernel keeps the older history lane-per-grid-point in registers


def set_mode_env(monkeypatch, mode, waves_per_rod=1):
    """kr_create reads KR_MS_MODE / KR_PERSISTENT / KR_WAVES_PER_ROD.  The modes pin one wavefront per rod; the
    several-wavefront form of path 1 (kr_msw_impl.hpp) has its own tests (test_gpu_msw.py)."""
    monkeypatch.setenv("KR_MS_MODE", "0" if mode == "single" else "1")
    monkeypatch.setenv("KR_PERSISTENT", "1" if mode in ("persistent", "overlap") else "0")
    monkeypatch.setenv("KR_OVERLAP", "1" if mode == "overlap" else "0")
    monkeypatch.setenv("KR_WAVES_PER_ROD", str(waves_per_rod))


def make_robot(mod, N, use_fsolve=True):
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    r = CosseratRod(use_fsolve=use_fsolve)
    if mod != "default":
        setup_robot(r, mod)
    r.N = N
    r.compute_intermediate_terms()
    return r


def inject(robot, mlp):
    """What physics_train.py:104-110 does, with plain strings standing in for the torch modules
    (the reference only ever looks at str(layer))."""
    import cosserat_oracle as orc
    names = {orc.ACT_TANH: "Tanh()", orc.ACT_SOFTPLUS: "Softplus(beta=1.0, threshold=20.0)",
             orc.ACT_RELU: "ReLU()", orc.ACT_ELU: "ELU(alpha=1.0)"}
    model, params = [], []
    for W, b, a in zip(mlp.weights, mlp.biases, mlp.acts):
        model.append(f"Linear(in_features={W.shape[1]}, out_features={W.shape[0]}, bias=True)")
        params += [W, b]
        if a != orc.ACT_NONE:
            model.append(names[a])
    robot.nn_model = model
    robot.param_ls = params
    robot.nn_path = "whatever"
    robot.nn_input_history = mlp.history


def mlp_on_matrix_cores(mlp):
    """Networks the in-sweep matrix-core evaluator serves (kr_set_mlp): 28 inputs, one or two hidden layers
    (two: the first at most 64 wide) with one activation, no history inputs."""
    if mlp is None:
        return True
    n = len(mlp.weights)
    if mlp.history or mlp.weights[0].shape[1] != 28 or n not in (2, 3):
        return False
    if n == 3 and (mlp.weights[0].shape[0] > 64 or mlp.acts[0] != mlp.acts[1]):
        return False
    return True


def expected_path(mode, N, mlp=None, scheme="euler"):
    """The kernel kr_simulate_batch uses for this problem in this mode (mirror of ms_eligible /
    launch_sim_persistent in kr_ms_impl.hpp)."""
    ms_ok = (N - 1 >= 2 * MS_P) and mlp_on_matrix_cores(mlp)
    if mode == "single" or not ms_ok:
        return 0
    if mode == "multi":
        return 1
    if N > PERSIST_MAX_N or (mlp is not None and scheme != "euler"):
        return 1
    if mlp is not None and len(mlp.weights) == 3 and mlp.weights[1].shape[0] > 192:
        return 1  # (the persistent kernels carry the base + JVP evaluator only: mlp_jvp.hpp, MJ_ACT_SLOTS)
    if mode == "overlap":  # Euler sweeps, MLP off, diagonal material matrices (every preset)
        return 3 if (mlp is None and scheme == "euler") else 2
    return 2


def require_path(mode, N, mlp=None, scheme="euler"):
    want = expected_path(mode, N, mlp, scheme)
    if want != PATH_OF_MODE[mode]:
        pytest.skip(f"the {mode} kernel does not serve N={N}, scheme={scheme}, this MLP (would run path {want}: "
                    f"covered by another parametrisation)")
    return want


def assert_path(robot_or_handle, want, waves_per_rod=1):
    h = robot_or_handle if hasattr(robot_or_handle, "get_option") else robot_or_handle._handle
    got = h.get_option("last_sim_path")
    if got == 2 and h.get_option("last_overlap"):
        got = 3
    assert got == want, f"kernel path {got} ran, the test is meant to exercise path {want}"
    if want == 1:
        w = h.get_option("last_waves_per_rod")
        assert w == waves_per_rod, f"{w} wavefronts per rod ran, the test is meant to exercise {waves_per_rod}"
