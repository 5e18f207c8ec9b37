"""CPU-only tier (``-m "not gpu"``): the C-ABI library loads and exports every
symbol ``include/knode_rod.h`` declares, the host-side logic of the shims
(parameters, presets, derived terms, controls, MLP parsing, pickling) agrees
with the oracle / golden vectors, the product path fails loudly without a GPU,
and the data-parallel plumbing works under gloo with world_size 2.
No compute kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT, load_golden

GOLDEN = os.path.join(ROOT, "tests", "golden")

import cosserat_oracle as orc

HEADER = os.path.join(ROOT, "include", "knode_rod.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import krod_native as kn
    if not os.path.exists(kn.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.build()
    return kn.load()


def test_library_exports_every_declared_symbol(lib):
    import krod_native as kn
    declared = _declared_symbols()
    assert len(declared) >= 20
    raw = ctypes.CDLL(kn.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/knode_rod.h but not exported"
    # and the Python binding covers the same set
    assert sorted(kn.EXPORTED_SYMBOLS) == declared
    assert lib.kr_version() >= 100


def test_struct_layouts_match_header():
    """ctypes mirrors of kr_params / kr_derived have the C layout (checked by compiling a probe)."""
    import krod_native as kn
    probe = os.path.join(ROOT, "gpurun_out", "_layout_probe")
    os.makedirs(os.path.dirname(probe), exist_ok=True)
    with open(probe + ".c", "w") as f:
        f.write('#include <stdio.h>\n#include <stddef.h>\n#include "knode_rod.h"\n'
                'int main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(kr_params), sizeof(kr_derived),'
                'offsetof(kr_params,E), offsetof(kr_params,del_t), offsetof(kr_params,w0), offsetof(kr_derived,rhoJ));return 0;}\n')
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), probe + ".c", "-o", probe], check=True)
    vals = [int(v) for v in subprocess.run([probe], capture_output=True, text=True, check=True).stdout.split()]
    assert vals == [ctypes.sizeof(kn.KrParams), ctypes.sizeof(kn.KrDerived), kn.KrParams.E.offset,
                    kn.KrParams.del_t.offset, kn.KrParams.w0.offset, kn.KrDerived.rhoJ.offset]


MODS = ["default", None, "noair", "nsw", "short", "damping", "dampstiff", "lengthstiff", "youngs"]


@pytest.mark.parametrize("mod", MODS)
def test_derived_terms_and_presets(lib, mod):
    """CosseratRod.__init__ / compute_intermediate_terms / knode.setup_robot against the oracle."""
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    r = CosseratRod(use_fsolve=True)
    if mod != "default":
        setup_robot(r, mod)
    r.N = 37
    r.compute_intermediate_terms()
    D = orc.params_for(mod, 37).derived()
    for name in ("A", "G", "ds", "c0", "c1", "c2", "rhoA"):
        assert getattr(r, name) == pytest.approx(getattr(D, name), rel=1e-14), name
    for a, b in ((r.J, D.J), (r.Kse, D.Kse), (r.Kbt, D.Kbt), (r.Kse_plus_c0_Bse_inv, D.Kse_inv),
                 (r.Kbt_plus_c0_Bbt_inv, D.Kbt_inv), (r.Kse_vstar, D.Kse_vstar), (r.rhoAg, D.rhoAg), (r.rhoJ, D.rhoJ)):
        assert np.allclose(a, b, rtol=1e-13, atol=0)
    assert np.allclose(r.tendon_dirs, D.P.tendon_dirs, atol=1e-16)
    assert r.tendon_offset == (0.02 if mod == "default" else 0.04445)


def test_preset_errors(lib):
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    r = CosseratRod()
    with pytest.raises(Exception, match="Unknown mod"):
        setup_robot(r, "bogus")
    with pytest.raises(Exception, match="no longer supported"):
        setup_robot(r, None, original=True)
    r.N = 1
    with pytest.raises(Exception):
        r.compute_intermediate_terms()


def test_torch_twin_host_side(lib):
    import io
    import torch
    from cosserat_ode_torch import CosseratRodTorch, mlp_structure
    from knode import setup_robot
    import krod_native as kn
    rob = CosseratRodTorch("cpu", 64)
    setup_robot(rob, "dampstiff")
    D = orc.params_for("dampstiff", 10).derived()
    assert rob.ds == pytest.approx(D.ds) and rob.c0 == pytest.approx(D.c0)
    assert np.allclose(rob.Kbt_plus_c0_Bbt_inv.numpy(), D.Kbt_inv, rtol=1e-6)
    # reference initialisation: non-negative weights (cosserat_ode_torch.py:90-105), state-dict key names
    assert list(rob.nn_models.state_dict().keys()) == ["0.weight", "0.bias", "2.weight", "2.bias"]
    assert float(rob.nn_models[0].weight.min()) >= 0 and rob.nn_models[0].weight.shape == (64, 28)
    assert CosseratRodTorch("cpu", 16, nn_input_history=True).nn_models[0].weight.shape == (16, 53)
    st = mlp_structure(rob.nn_models)
    assert [a for _, a in st] == [kn.ACT_ELU, kn.ACT_NONE]
    buf = io.BytesIO()
    torch.save({"robot": rob}, buf)
    buf.seek(0)
    rob2 = torch.load(buf, weights_only=False)["robot"]
    assert torch.equal(rob2.nn_models[2].weight, rob.nn_models[2].weight) and rob2._handle is None
    # no CPU fallback: compute entry points raise
    with pytest.raises(kn.KrError):
        rob.ODE_parallel(torch.zeros(2, 19), torch.zeros(2, 19), torch.zeros(2, 6), torch.zeros(2, 3))


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import krod_native as kn
    from cosserat_ode import CosseratRod
    from knode import simulate
    r = CosseratRod(use_fsolve=True)
    with pytest.raises(kn.KrError, match="no CPU fallback"):
        r.ODE(np.zeros(19), np.zeros(19), np.zeros(6), np.zeros(3))
    with pytest.raises(kn.KrError):
        simulate(r, [[6, 5, 5, 6]] * 3)
    # the product package never imports the oracle
    pkg = os.path.join(ROOT, "knode-cosserat_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "cosserat_oracle" not in open(os.path.join(pkg, fn)).read(), fn


def test_controls_match_reference(lib):
    from physics_controls import calc_controls
    g = load_golden("small")
    for key in g.files:
        if key.startswith("ctl_"):
            _, kind, arg, dt, T = key.split("_")
            got = np.array(calc_controls(kind, float(arg), float(dt), int(T)))
            assert np.array_equal(got, g[key]), key
    with pytest.raises(Exception, match="Unknown control type"):
        calc_controls("zigzag", 1.0, 0.05, 3)
    with pytest.raises(Exception):
        calc_controls("ramp", 1.0, 0.05, 3)


def test_quaternion_to_euler_matches_reference(lib):
    import torch
    from Utils.transformations import quaternion_to_euler
    g = load_golden("small")
    e = quaternion_to_euler(torch.tensor(g["quat"])).numpy()
    far = np.r_[0:40, 80:400]
    assert np.allclose(e[:, far], g["euler"][:, far], atol=2e-6)


def test_mlp_layer_string_parsing(lib):
    from cosserat_ode import mlp_from_layer_strings
    import krod_native as kn
    W0, b0, W1, b1 = np.ones((4, 28)), np.zeros(4), np.ones((25, 4)), np.zeros(25)
    model = ["Linear(in_features=28, out_features=4, bias=True)", "Dropout(p=0.5, inplace=False)", "Tanh()",
             "Linear(in_features=4, out_features=25, bias=True)"]
    w, b, a = mlp_from_layer_strings(model, [W0, b0, W1, b1])
    assert a == [kn.ACT_TANH, kn.ACT_NONE] and w[1].shape == (25, 4) and w[0].dtype == np.float32
    with pytest.raises(kn.KrError):
        mlp_from_layer_strings(["ReLU()", "Linear(...)"], [W0, b0])


def test_shard_range_and_bucket():
    from krod_train import shard_range
    for n in (0, 1, 7, 4096, 4099):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_DIST_WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "knode-cosserat_amd"))
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch, torch.distributed as dist
from krod_train import FlatBucket, shard_range
world, n_traj = int(sys.argv[4]), int(sys.argv[5])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=world)
rank = dist.get_rank()
shapes = [(64, 28), (64,), (25, 64), (25,)]
b = FlatBucket(shapes, "cpu")
assert b.flat.numel() == 64 * 28 + 64 + 25 * 64 + 25 + 1
# every rank contributes the gradients of its own trajectory shard (value = 1 + global trajectory index); a rank whose
# shard is empty (more ranks than trajectories) contributes zeros - train_knode.py hands KnodeTrainer the empty shard
lo, hi = shard_range(n_traj, rank, world)
for v in b.views:
    v.fill_(float(sum(t + 1 for t in range(lo, hi))))
b.loss.fill_(float(hi - lo))
b.all_reduce()
total = float(sum(t + 1 for t in range(n_traj)))
assert all(float(v.min()) == float(v.max()) == total for v in b.views), "sum over ranks"
assert float(b.loss) == float(n_traj)
# parameters that start identical and see the same reduced gradient stay identical
p = torch.nn.Parameter(torch.ones(64, 28)); p.grad = b.views[0]
opt = torch.optim.Adam([p], lr=1e-2); opt.step()
ref = [torch.zeros_like(p) for _ in range(world)]
dist.all_gather(ref, p.detach())
assert all(torch.equal(ref[0], r) for r in ref)
# bench.py shards the global batch of rods: rank r simulates rods [r B, (r + 1) B) of the world x B draw
import bench
B, steps = 5, 7
mine = torch.as_tensor(bench.rank_controls(B, world, rank, steps, 0.05))
parts = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(parts, mine)
whole = torch.as_tensor(bench.rank_controls(B * world, 1, 0, steps, 0.05))
assert torch.equal(torch.cat(parts), whole), "per-rank slices tile the global batch"
assert mine.shape == (B, steps, 4) and not torch.equal(parts[0], parts[-1])
dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.parametrize("world,n_traj", [(2, 7), (3, 2)])
def test_data_parallel_allreduce_gloo(tmp_path, world, n_traj):
    """gloo on CPU, world_size 2 (7 trajectories) and 3 (2 trajectories: one rank holds an empty shard and must add
    zeros): one flat all-reduce carries every gradient and the loss; bench.py's per-rank rod slices tile the
    global batch."""
    script = tmp_path / "w.py"
    script.write_text(_DIST_WORKER)
    port = str(29500 + (os.getpid() * 3 + world) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(world), str(n_traj)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"ok {r}" in o


def test_train_driver_shards_without_duplicates():
    """train_knode.py hands every rank shard_range(...) as is: no rank re-trains a trajectory another rank owns
    (round 1 gave idle ranks trajectory 0 again, which the SUM all-reduce then counted several times)."""
    src = open(os.path.join(ROOT, "knode-cosserat_amd", "train_knode.py")).read()
    assert "lo, hi = 0, 1" not in src
    from krod_train import shard_range
    for world in (3, 8):
        owned = [t for r in range(world) for t in range(*shard_range(2, r, world))]
        assert sorted(owned) == [0, 1]


def test_reference_checkpoint_loads_without_reference_code(tmp_path):
    """tests/golden/ref_checkpoint.pth was written by the REFERENCE classes exactly as physics_train.py:
    284-288 does.  With this package on sys.path it unpickles into our CosseratRodTorch (no reference code
    anywhere), with the same attributes and weights; the weights-only format round-trips."""
    import torch
    import krod_checkpoint as kc
    from cosserat_ode_torch import CosseratRodTorch
    g = load_golden("checkpoint")
    ckpt = kc.load_checkpoint(os.path.join(GOLDEN, "ref_checkpoint.pth"), "cpu")
    rob = ckpt["robot"]
    assert type(rob) is CosseratRodTorch and type(rob).__module__ == "cosserat_ode_torch"
    assert "/root/reference" not in (sys.modules["cosserat_ode_torch"].__file__ or "")
    assert ckpt["dtw"] == [[1.5]] and ckpt["loss"] == [0.25] and "state" in ckpt["optim"]
    assert rob.L == float(g["L"]) and rob.del_t == float(g["del_t"]) and rob.E == float(g["E"])
    assert np.array_equal(rob.Bbt.numpy(), g["Bbt"])
    assert [str(l) for l in rob.nn_models] == [str(s) for s in g["layer_strings"]]
    sd = rob.nn_models.state_dict()
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g["p_" + k])
    # the NumPy class's reader (cosserat_ode.py:81-88)
    nn_model = torch.load(os.path.join(GOLDEN, "ref_checkpoint.pth"), map_location="cpu", weights_only=False)["robot"].nn_models
    assert len(nn_model) == 3
    # weights-only round trip
    kc.save_weights(tmp_path / "w.npz", rob.nn_models)
    ml, param_ls = kc.load_weights(tmp_path / "w.npz")
    assert [str(l) for l in ml] == [str(l) for l in rob.nn_models]
    for a, (k, v) in zip(param_ls, sd.items()):
        assert np.array_equal(a, v.numpy())
    # and our own checkpoint goes through the same door
    kc.save_checkpoint(tmp_path / "ours.pth", rob, dtw=[[2.0]], loss=[0.5])
    again = kc.load_checkpoint(tmp_path / "ours.pth")["robot"]
    assert torch.equal(again.nn_models[0].weight, rob.nn_models[0].weight)


def test_dtw_and_eval_metrics(tmp_path):
    """krod_eval: exact DTW against the textbook recursion, the pos+Euler MSE against a direct evaluation,
    and the reference's trajectory file layout."""
    import krod_eval as ke
    rng = np.random.default_rng(0)
    for (ta, tb, d) in ((1, 1, 3), (5, 9, 3), (40, 33, 3), (17, 17, 1)):
        a, b = rng.standard_normal((ta, d)), rng.standard_normal((tb, d))
        D = np.full((ta + 1, tb + 1), np.inf)
        D[0, 0] = 0
        for i in range(1, ta + 1):
            for j in range(1, tb + 1):
                D[i, j] = np.abs(a[i - 1] - b[j - 1]).sum() + min(D[i - 1, j], D[i, j - 1], D[i - 1, j - 1])
        assert abs(ke.dtw_distance(a, b) - D[ta, tb]) < 1e-12 * max(1.0, D[ta, tb])
    x = rng.standard_normal(30)
    assert ke.dtw_distance(x, x) == 0.0
    assert ke.dtw_distance(np.repeat(x, 2), x) == 0.0  # warping absorbs a uniform slow-down
    # FastDTW (restated from Salvador & Chan 2007; what the reference's evaluate calls with radius 1): never below the
    # exact DTW, equal to it once the radius covers the table, on short series (below radius + 2 samples the algorithm IS
    # the exact one) and on the kind of data it is used on (two close, smooth tip paths); hand-checked small case
    for (ta, tb, d) in ((1, 1, 3), (2, 5, 3), (40, 33, 3), (64, 64, 1), (101, 57, 3)):
        a, b = rng.standard_normal((ta, d)), rng.standard_normal((tb, d))
        ex = ke.dtw_distance(a, b)
        f1 = ke.fastdtw_distance(a, b)
        assert f1 >= ex - 1e-12
        assert abs(ke.fastdtw_distance(a, b, radius=max(ta, tb)) - ex) < 1e-10 * max(1.0, ex)
        if min(ta, tb) < 3:
            assert abs(f1 - ex) < 1e-12
    t = np.linspace(0, 4, 200)
    path = np.stack([0.1 * np.sin(3 * t), 0.1 * np.cos(2 * t), 0.4 + 0.01 * t], axis=1)
    other = path + 1e-3 * np.stack([np.sin(5 * t), np.cos(7 * t), np.sin(t)], axis=1)
    assert abs(ke.fastdtw_distance(path, other) - ke.dtw_distance(path, other)) < 0.05 * ke.dtw_distance(path, other)
    assert ke.fastdtw_distance(x, x) == 0.0
    # x = [0 1 2 3], y = [0 0 1 2 3]: halves (0.5, 2.5) / (0, 1.5), coarse path (0,0) (1,1); refined window holds the
    # optimal alignment of cost 0
    assert ke.fastdtw_distance(np.array([0.0, 1, 2, 3]), np.array([0.0, 0, 1, 2, 3])) == 0.0
    # pos + euler MSE
    T, N = 6, 5
    tr = rng.standard_normal((T, 25, N)); rf = tr + 0.01 * rng.standard_normal((T, 25, N))
    tr[:, 3] += 3; rf[:, 3] += 3
    m = ke.pos_euler_mse(tr, rf)
    eul = lambda q: np.array([orc.quaternion_to_euler(qq / np.linalg.norm(qq)) for qq in q])
    q1 = tr[:, 3:7].transpose(0, 2, 1).reshape(-1, 4); q2 = rf[:, 3:7].transpose(0, 2, 1).reshape(-1, 4)
    assert m > 0 and np.isfinite(m)
    se_pos = ((tr[:, :3] - rf[:, :3]).reshape(-1, 3)) ** 2
    from scipy.spatial.transform import Rotation
    e1 = Rotation.from_quat(q1[:, [1, 2, 3, 0]]).as_euler("zyx"); e2 = Rotation.from_quat(q2[:, [1, 2, 3, 0]]).as_euler("zyx")
    assert abs(m - np.mean(np.concatenate([(e1 - e2) ** 2, se_pos])) * 1000) < 1e-12
    # file layout of simulate.py:97-100
    ke.save_trajectory(tmp_path / "t.npy", np.zeros((4, 50, 3)), np.ones((3, 4)))
    d = np.load(tmp_path / "t.npy", allow_pickle=True).item()
    assert d["traj"].shape == (4, 50, 3) and d["controls"].shape == (3, 4)
    t2, c2 = ke.load_trajectory(tmp_path / "t.npy")
    assert t2.dtype == np.float64 and np.array_equal(c2, np.ones((3, 4)))


def test_legacy_preset_table():
    """kr_apply_preset_original restates knode_cosserat_realworld/prepare.py:35-73 (host only)."""
    import krod_native as kn
    lib = kn.load()
    want = {  # mod: (L, E, r, Bbt, g_z)
        None: (0.4, 209e9, 0.0012, 5e-4, -9.81), "nsw": (0.4, 209e9, 0.0012, 5e-4, 0.0),
        "short": (0.3, 209e9, 0.0012, 5e-4, -9.81), "damping": (0.4, 209e9, 0.0012, 9e-4, -9.81),
        "diameter": (0.4, 209e9, 0.002, 5e-4, -9.81), "youngs": (0.4, 109e9, 0.0012, 5e-4, -9.81),
        "dampstiff": (0.4, 109e9, 0.0012, 3e-2, -9.81), "lengthstiff": (0.3, 109e9, 0.0012, 5e-4, -9.81),
    }
    for mod, (L, E, r, bbt, gz) in want.items():
        p = kn.KrParams()
        kn.check(lib.kr_default_params(p))
        assert lib.kr_apply_preset_original(p, None if mod is None else mod.encode()) == 0
        assert (p.del_t, p.L, p.E, p.r, p.rho) == (0.005, L, E, r, 8000.0)
        assert [p.Bbt[i] for i in range(9)] == [bbt, 0, 0, 0, bbt, 0, 0, 0, bbt] and p.g[2] == gz
    p = kn.KrParams()
    assert lib.kr_apply_preset_original(p, b"noair") != 0  # not a modifier of the legacy set


@pytest.mark.parametrize("N", [10, 12])
def test_estimate_state_oracle_matches_reference(N):
    """oracle/estimate_oracle.py (the checker of the device implementation) against
    knode_cosserat_realworld/estimate_state.py:158-242 run by the reference (fixture estimate_state.npz)."""
    import estimate_oracle as kest
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    g = load_golden("estimate_state")
    r = CosseratRod(use_fsolve=True)
    setup_robot(r)
    r.N = N
    r.compute_intermediate_terms()
    est = kest.estimate_state(g[f"N{N}_data"], g[f"N{N}_ctl"], r)
    want = g[f"N{N}_est"]
    assert est.shape == want.shape
    for rows, name in ((slice(0, 7), "p,h"), (slice(13, 19), "q,w"), (slice(7, 13), "n,m"), (slice(19, 25), "v,u")):
        err = np.linalg.norm(est[:, rows] - want[:, rows]) / np.linalg.norm(want[:, rows])
        assert err < 1e-9, (name, err)
    assert np.allclose(np.asarray(r.vstar, dtype=np.float64), g[f"N{N}_vstar_after"], rtol=0, atol=1e-12)


def test_wwm_spill_scanner(tmp_path):
    """tools/wwm_spill_scan.py: an ordinary VGPR spill inside a whole-wave-mode bracket is
    flagged, the save of an SGPR-spill register and a spill outside a bracket are not."""
    import subprocess
    import sys
    asm = tmp_path / "k.s"
    asm.write_text("""
_ZN2kr6kernelEv:
	v_writelane_b32 v255, s4, 3
	scratch_store_dwordx4 off, v[2:5], off offset:608 ; 16-byte Folded Spill
	s_or_saveexec_b64 s[100:101], -1
	scratch_store_dword off, v255, off offset:16 ; 4-byte Folded Spill
	s_mov_b64 exec, s[100:101]
	s_or_saveexec_b64 s[100:101], -1
	v_mov_b32_e32 v255, v250
	scratch_store_dwordx4 off, v[2:5], off offset:608 ; 16-byte Folded Spill
	s_mov_b64 exec, s[100:101]
	s_endpgm
""")
    tool = os.path.join(ROOT, "tools", "wwm_spill_scan.py")
    r = subprocess.run([sys.executable, tool, str(asm)], capture_output=True, text=True)
    assert r.returncode == 1 and "1 spill(s)" in r.stdout and "offset:608" in r.stdout, r.stdout
    clean = tmp_path / "c.s"
    clean.write_text(asm.read_text().replace("	v_mov_b32_e32 v255, v250\n	scratch_store_dwordx4 off, v[2:5], off offset:608 ; 16-byte Folded Spill\n", "	v_mov_b32_e32 v255, v250\n"))
    r = subprocess.run([sys.executable, tool, str(clean)], capture_output=True, text=True)
    assert r.returncode == 0 and "0 spill(s)" in r.stdout, r.stdout


def test_shipped_assembly_has_no_spill_in_whole_wave_brackets():
    """The build keeps the gfx950 assembly of every translation unit (csrc/Makefile, -save-temps of the same compile
    that made the object) and runs the scanner as a build gate; here it runs again over what the build left, and every
    unit of the Makefile's SRCS list must be there (the gate cannot be skipped by not producing the assembly)."""
    import re
    import subprocess
    import sys
    mk = open(os.path.join(PKG, "csrc", "Makefile")).read()
    srcs = re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split()
    asm_dir = os.path.join(PKG, "lib", "asm")
    if not os.path.isdir(asm_dir):
        pytest.skip("no build in this tree (the assembly is written next to the objects; the GPU box only gets the .so)")
    files = [os.path.join(asm_dir, s.replace(".hip", ".s")) for s in srcs]
    missing = [f for f in files if not os.path.exists(f)]
    assert not missing, missing
    assert "kr_msn_f64.hip" in srcs and "-amdgpu-spill-sgpr-to-vgpr=0" in mk
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wwm_spill_scan.py")] + files,
                       capture_output=True, text=True)
    assert r.returncode == 0 and "0 spill(s)" in r.stdout, r.stdout[-2000:]


# ---------------------------------------------------------------------------
# the reference-side ctypes stub is checked against the header, so it cannot rot silently
# ---------------------------------------------------------------------------
def integration_stub_source():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## B. Binding the C ABI directly"):]
    code = sec[sec.index("```python") + len("```python"):]
    return code[:code.index("```")]


def header_prototypes():
    """{name: number of parameters} of every function include/knode_rod.h declares, and the fields of struct kr_params."""
    h = open(HEADER).read()
    h_nc = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char\s*\*|size_t)\s+(kr_\w+)\s*\(([^;{]*?)\)\s*;", h_nc, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    st = re.search(r"typedef struct kr_params\s*\{(.*?)\}\s*kr_params\s*;", h_nc, flags=re.S)
    fields = []
    for decl in st.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype = decl.split()[0]
        for item in decl[len(ctype):].split(","):
            mm = re.match(r"\s*(\w+)\s*(?:\[(\d+)\])?\s*$", item)
            fields.append((mm.group(1), ctype, int(mm.group(2)) if mm.group(2) else 0))
    return protos, fields


def test_integration_stub_matches_the_header():
    import ast
    src = integration_stub_source()
    tree = ast.parse(src)
    protos, fields = header_prototypes()
    assert len(protos) >= 40 and "kr_simulate_batch" in protos and "kr_ode_vjp_batch" in protos
    # (1) every _lib.kr_* call passes as many positional arguments as the prototype has parameters
    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)
             and isinstance(n.func.value, ast.Name) and n.func.value.id == "_lib" and n.func.attr.startswith("kr_")]
    seen = set()
    for c in calls:
        name = c.func.attr
        assert name in protos, f"INTEGRATION.md section B calls {name}, which include/knode_rod.h does not declare"
        assert not c.keywords
        assert len(c.args) == protos[name], (name, len(c.args), protos[name])
        seen.add(name)
    assert {"kr_default_params", "kr_create", "kr_state_init_straight", "kr_simulate_batch", "kr_state_unpack",
            "kr_destroy"} <= seen
    # (2) the Structure mirrors struct kr_params: same names, order, element types and array lengths
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "kr_params")
    assign = next(n for n in cls.body if isinstance(n, ast.Assign) and n.targets[0].id == "_fields_")
    got = []
    for elt in assign.value.elts:
        fname = elt.elts[0].value
        t = elt.elts[1]
        if isinstance(t, ast.BinOp):      # C.c_double * 9
            got.append((fname, t.left.attr, t.right.value))
        else:
            got.append((fname, t.attr, 0))
    ctype_of = {"double": "c_double", "int32_t": "c_int32", "int": "c_int32"}
    want = [(n, ctype_of[t], k) for n, t, k in fields]
    assert got == want, (got, want)


@pytest.mark.gpu
def test_integration_stub_runs():
    """The stub itself, executed: its simulate_gpu() on a reference-shaped robot object reproduces the reference's cfg1 run."""
    import types
    import numpy as np
    from conftest import rel_l2
    import krod_native as kn
    src = integration_stub_source().replace('C.CDLL("libknode_rod.so")', f'C.CDLL({kn.LIB_PATH!r})')
    mod = types.ModuleType("_rod_backend")
    exec(compile(src, "INTEGRATION.md:B", "exec"), mod.__dict__)
    g = load_golden("sim_cfg1")
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    robot = CosseratRod(use_fsolve=True)
    setup_robot(robot)
    robot.N = 20
    robot.compute_intermediate_terms()
    T = 40
    out = mod.simulate_gpu(robot, g["ctl"][:T])
    assert out.shape == (T + 1, 25, 20)
    assert rel_l2(out[:T, :3, -1], g["tip"][:T]) < 1e-8


# ---- tools/profile_summary.py: a counter file of another workload must be refused -------------
def _fake_profile_tree(tmp, tag, kernel_name, dur_ns, value_kb, own_ms_per_step=0.0285):
    """gpurun_out-like tree: the bench line, and for the FETCH / WRITE passes the profiled process's own bench line plus a
    counter_collection.csv with five dispatches of `kernel_name` lasting dur_ns each (and one unrelated torch kernel)."""
    import json
    G = os.path.join(tmp, "gpurun_out")
    os.makedirs(G)
    line = {"metric": "rod-steps/sec (N=100 segments, batch=1024)", "value": 35.9e6, "steps": 1000, "ms_per_step": 0.0285, "dtype": "f64",
            "config": {"rods_per_gpu": 1024, "N": 100},
            "roofline": {"kernel": "kr::mso_sim_kernel (persistent, overlapped steps)", "kernel_ms": 28.5, "launches": 1,
                         "hbm": {"algorithmic_bytes_per_launch": 57344000}}}
    open(os.path.join(G, f"{tag}_bench.json"), "w").write(json.dumps(line) + "\n")
    own = dict(line, ms_per_step=own_ms_per_step)
    hdr = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
           '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
           '"Start_Timestamp","End_Timestamp"\n')
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        p = f"{tag}_bench_{'fetch' if cname == 'FETCH_SIZE' else 'write'}"
        os.makedirs(os.path.join(G, p, "box"))
        open(os.path.join(G, p + ".out"), "w").write("[bench] noise\n" + json.dumps(own) + "\n")
        rows = [f'1,1,"Agent 2",1,7,7,65536,67,"void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float> >(int)",'
                f'128,0,0,8,0,64,"{cname}",1024.0,1000,9000\n']
        t = 100000
        for d in range(2, 7):
            rows.append(f'{d},{d},"Agent 2",1,7,7,262144,9,"{kernel_name}",256,0,0,128,0,96,"{cname}",{value_kb},{t},{t + dur_ns}\n')
            t += dur_ns + 5000
        open(os.path.join(G, p, "box", "7_counter_collection.csv"), "w").write(hdr + "".join(rows))
    return G


def _load_profile_summary():
    import importlib.util
    spec = importlib.util.spec_from_file_location("profile_summary", os.path.join(ROOT, "tools", "profile_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_profile_summary_accepts_the_right_kernel(tmp_path):
    import json
    ps = _load_profile_summary()
    G = _fake_profile_tree(str(tmp_path), "t0", "void kr::mso_sim_kernel<double, true, 18, 1>(kr::RodConst<double>, kr::SimArgs<double>)",
                           28_540_000, 5_306_000.0)
    P = os.path.join(str(tmp_path), "profiles")
    ps.summarise("t0", G, P, log=lambda m: None)
    hbm = json.load(open(os.path.join(P, "t0_pmc_hbm.json")))
    # write 5.306 GB + 2 x fetch 5.306 GB over 1.024 M rod-steps
    assert abs(hbm["hbm_bytes_per_rod_step"] - 3 * 5_306_000.0 * 1024 / (1024 * 1000)) < 1.0
    assert hbm["FETCH_SIZE_dispatches"] == 5 and "<double" in hbm["WRITE_SIZE_kernel_name"]


@pytest.mark.parametrize("case", ["other_dtype", "other_duration", "other_run"])
def test_profile_summary_refuses_a_mislabelled_counter_file(tmp_path, case):
    """The round-4 accident restated: the raw directory holds `mso_sim_kernel<float, ...>` launches of 2.2 ms written by the
    training script (or launches of the right type but the wrong length, or a process whose own timing disagrees with the
    bench line) - nothing may be summarised under the headline's name."""
    ps = _load_profile_summary()
    name, dur, own = {
        "other_dtype": ("void kr::mso_sim_kernel<float, true, 20, 1>(kr::RodConst<float>, kr::SimArgs<float>)", 2_200_000, 0.0285),
        "other_duration": ("void kr::mso_sim_kernel<double, true, 18, 1>(kr::RodConst<double>, kr::SimArgs<double>)", 2_200_000, 0.0285),
        "other_run": ("void kr::mso_sim_kernel<double, true, 18, 1>(kr::RodConst<double>, kr::SimArgs<double>)", 28_540_000, 0.0400),
    }[case]
    G = _fake_profile_tree(str(tmp_path), "t1", name, dur, 726_452.0, own_ms_per_step=own)
    P = os.path.join(str(tmp_path), "profiles")
    problems = ps.summarise("t1", G, P, log=lambda m: None)
    assert problems and any("bench_fetch" in p for p in problems), problems
    assert not os.path.exists(os.path.join(P, "t1_pmc_hbm.json"))
    rows = ps.read_rows(os.path.join(G, "t1_bench_fetch", "box", "7_counter_collection.csv"))
    if case != "other_run":
        with pytest.raises(ps.ProfileMismatch):
            ps.timed_dispatches(rows, "mso_sim_kernel", "f64", 28_500.0, "unit")
