"""-m gpu: the symmetric tile form of the normal equations (csrc/pgo_sym.h: every interior off-diagonal block stored and read
once by the CG products) against the incidence-slot kernels it stands in for — same matrix, same CG, so the same solution to
rounding; bitwise reproducible against itself.  PGO_SYM=1 forces the form on graphs below the size where it is the default."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Sym:
    def __init__(self, on, rows=None):
        self.new = {"PGO_SYM": "1" if on else "0"}
        self.rows = rows          # (rows per tile: the knob sym_rows, csrc/pgo_tuning.h)

    def __enter__(self):
        import pgo_loader
        self.old = {k: os.environ.get(k) for k in ("PGO_SYM", "PGO_NO_PIPELINE")}
        for k in self.old:
            os.environ.pop(k, None)
        os.environ.update(self.new)
        os.environ["PGO_NO_PIPELINE"] = "1"      # the host-driven CG is the one that reads the symmetric form
        pgo_loader.load().tuning_set("sym_rows", self.rows if self.rows else None)

    def __exit__(self, *a):
        import pgo_loader
        pgo_loader.load().tuning_set("sym_rows", None)
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _exact(gpu, g, d2, b):
    """The same system through the GPU factorisation (pgo_linear_solve, SPARSE_NORMAL_CHOLESKY): the reference both CG forms are held to."""
    with _Sym(False):
        prob, _ = gpu.problem_from_graph(g)
        x, _ = prob.linear_solve(d2, b, gpu.SolverOptions(linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY))
    return x


def _graphs(ds):
    fat = ds.manhattan_se3(600, 3000, seed=5, loop_radius=6.0)          # dense revisits: rows with > 40 incidences
    return {"manhattan": ds.manhattan_se3(3000, 12000, seed=11), "identity": ds.manhattan_se3(1500, 5000, seed=3, identity_information=True),
            "fat_rows": fat, "sphere": ds.sphere_layers(n_spheres=2, rings=20, per_ring=20), "chain": ds.manhattan_se3(1200, 1300, seed=23)}


@pytest.mark.parametrize("name", ["manhattan", "identity", "fat_rows", "sphere", "chain"])
@pytest.mark.parametrize("rows", [None, 16, 256])
def test_linear_solve_same_solution(gpu, ds, name, rows):
    """One damped Gauss-Newton system solved by PCG to a tight Q tolerance, both storage forms: the solutions agree to rounding
    and the CG takes the same number of iterations (+- 1: the products differ in the last bits)."""
    g = _graphs(ds)[name]
    rng = np.random.default_rng(1)
    d2 = rng.uniform(0.1, 1.0, size=6 * g.N)
    b = rng.normal(size=6 * g.N)
    b[:6] = 0.0       # constant pose
    opt = dict(linear_solver_type=gpu.BLOCK_JACOBI_PCG, eta=1e-10, max_linear_solver_iterations=400, pcg_cluster_poses=2)
    out = {}
    for on in (False, True, True):
        with _Sym(on, rows):
            prob, _ = gpu.problem_from_graph(g)
            x, it = prob.linear_solve(d2, b, gpu.SolverOptions(**opt))
        out.setdefault(on, []).append((x, it))
    (x0, it0), = out[False]
    (x1, it1), (x2, it2) = out[True]
    assert np.array_equal(x1, x2) and it1 == it2                   # reproducible bit for bit
    assert abs(it1 - it0) <= max(1, it0 // 100)      # (the products differ in the last bits: a long CG may stop an iteration or two apart)
    # both against the factorisation's solution of the same system
    xs = _exact(gpu, g, d2, b)
    e0, e1 = np.abs(x0 - xs).max(), np.abs(x1 - xs).max()
    assert e0 <= 1e-4 * np.abs(xs).max()
    assert e1 <= max(3.0 * e0, 1e-9 * np.abs(xs).max()), (e0, e1)


def test_general_information_full_blocks(gpu, ds):
    """Non-block-diagonal information: 36-entry slots (the unpacked kernels)."""
    g = ds.manhattan_se3(800, 2400, seed=9)
    rng = np.random.default_rng(2)
    A = rng.normal(size=(g.E, 6, 6)) * 0.3
    info = A @ np.transpose(A, (0, 2, 1)) + np.diag([4, 4, 4, 25, 25, 25.0])
    g.sqrt_info = np.linalg.cholesky(info).reshape(g.E, 36)
    d2 = rng.uniform(0.1, 1.0, size=6 * g.N)
    b = rng.normal(size=6 * g.N)
    b[:6] = 0.0
    opt = dict(linear_solver_type=gpu.BLOCK_JACOBI_PCG, eta=1e-10, max_linear_solver_iterations=400)
    res = []
    for on in (False, True):
        with _Sym(on, 32):
            prob, _ = gpu.problem_from_graph(g)
            res.append(prob.linear_solve(d2, b, gpu.SolverOptions(**opt)))
    assert abs(res[0][1] - res[1][1]) <= 1
    xs = _exact(gpu, g, d2, b)
    e0, e1 = np.abs(res[0][0] - xs).max(), np.abs(res[1][0] - xs).max()
    assert e1 <= max(3.0 * e0, 1e-9 * np.abs(xs).max()), (e0, e1)


@pytest.mark.parametrize("repack", [False, True])
def test_lm_solve_same_answer(gpu, ds, O, repack):
    """A whole LM solve (truncated PCG, Huber) with the CG products from the symmetric form: same iterations, same costs as the
    incidence-slot kernels, and the oracle's answer."""
    g = ds.manhattan_se3(4000, 16000, seed=21)
    opt = dict(max_num_iterations=12, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2, pcg_form=1)     # (pcg_form 1: Ceres' refreshed CG, k_spmv_sym<0>; the pipelined CG on the form is tests/test_gpu_sym_pipe.py)
    runs = []
    # repack: the incidence-slot linearisation stays and its blocks are copied once per LM iteration (knob sym_repack = 1); otherwise the
    # symmetric form is the session's only storage: the linearisation writes it, damping / cluster preconditioner / tail and refresh
    # products read and write it
    for on in (False, True):
        with _Sym(on), gpu.tuning(sym_repack=1 if (on and repack) else None):
            prob, poses = gpu.problem_from_graph(g)
            runs.append((gpu.solve(gpu.SolverOptions(**opt), prob), poses))
    (a, pa), (b, pb) = runs
    assert len(a.iterations) == len(b.iterations)
    assert list(a.iterations["step_is_successful"]) == list(b.iterations["step_is_successful"])
    assert np.allclose(a.iterations["cost"], b.iterations["cost"], rtol=1e-9)
    assert np.abs(pa - pb).max() < 1e-6
    op, osum, otr = O.solve(O.Graph(g.poses, g.ia, g.ib, g.meas, g.sqrt_info), O.default_options(max_num_iterations=12, linear_solver=1, pcg_cluster=2))
    assert b.final_cost == pytest.approx(osum.final_cost, rel=1e-6)


@pytest.mark.parametrize("lin", ["lean", "rows"])
@pytest.mark.parametrize("name", ["identity", "fat_rows", "sphere"])
def test_lm_solve_symmetric_storage_other_graphs(gpu, ds, name, lin, knobs):
    """Identity information (INFO 0), rows with many incidences (several chunks per tile, runs across wave boundaries), a mesh:
    whole LM solves with the symmetric form as the only storage against the incidence-slot kernels."""
    # lin: which kernel writes the form — the row kernel with the lean per-incidence algebra (k_linearize_lean, the default) or the row
    # kernel with the general body and redirected block stores (k_linearize_symout: information with position / rotation coupling;
    # the knob sym_lin_rows = 1 runs it on every graph)
    knobs(sym_lin_rows=1 if lin == "rows" else None)
    g = _graphs(ds)[name]
    opt = dict(max_num_iterations=10, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=1 if name == "sphere" else 2, pcg_form=1)
    runs = []
    for on in (False, True):
        with _Sym(on, 32):
            prob, poses = gpu.problem_from_graph(g)
            runs.append((gpu.solve(gpu.SolverOptions(**opt), prob), poses))
    (a, pa), (b, pb) = runs
    n = min(len(a.iterations), len(b.iterations))
    assert n >= 5 and list(a.iterations["step_is_successful"][:n]) == list(b.iterations["step_is_successful"][:n])
    assert np.allclose(a.iterations["cost"][:4], b.iterations["cost"][:4], rtol=1e-8)
    # (truncated PCG on an ill-conditioned mesh amplifies the last-bit differences of the products: 1e-9 at iteration 3, 2e-4 at 5)
    assert np.allclose(a.iterations["cost"][:n], b.iterations["cost"][:n], rtol=5e-3)


def _block_diagonal_information(ds, seed=8):
    g = ds.manhattan_se3(2500, 9000, seed=seed)
    rng = np.random.default_rng(4)
    L = np.zeros((len(g.ia), 6, 6))          # W_pp, W_rr full 3 x 3, no position / rotation coupling (INFO 2)
    for blk in (slice(0, 3), slice(3, 6)):
        L[:, blk, blk] = rng.normal(size=(len(g.ia), 3, 3)) * 0.3 + 3.0 * np.eye(3)
    g.sqrt_info = L
    return g


@pytest.mark.parametrize("name", ["manhattan", "identity", "fat_rows", "sphere", "chain", "block_diagonal", "soft_l_one", "no_loss"])
def test_lean_linearisation_equals_the_general_one(gpu, ds, name):
    """k_linearize_lean (csrc/pgo_lean_kernels.hip: hand-reduced algebra, pair sums on the DPP crossbar) against k_linearize_symout
    (the general body) at the same point: every block written into the form, every diagonal block, the gradient — largest
    difference relative to the largest entry of its group below 1e-12; both write exactly the same places of the form."""
    kw = {}
    if name == "block_diagonal":
        g = _block_diagonal_information(ds)
    elif name in ("soft_l_one", "no_loss"):
        g = ds.manhattan_se3(2000, 8000, seed=17)
        kw = dict(loss=gpu.SOFT_L_ONE if name == "soft_l_one" else gpu.TRIVIAL, loss_a=0.7)
    else:
        g = _graphs(ds)[name]
    with _Sym(True, 32):
        prob, _ = gpu.problem_from_graph(g, **kw)
        prob.solver_begin(gpu.SolverOptions(max_num_iterations=50, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2))
        prob.solver_step(2)          # (a point with Huber-active edges and Jacobi scales in place)
        worst = prob.time_kernel("sym_lean_check", 1)
        prob.solver_end()
    assert worst < 1e-12, worst


def test_fuzz_sweep_of_the_symmetric_form():
    """tools/fuzz_sym.py: 120 random graphs (lattice walks, random chords, hubs with hundreds of incidences, duplicate edges, all three
    information kinds, constant blocks, five losses, tile caps from 8 to 256 rows, both storage modes) — the symmetric form against the
    incidence-slot kernels and against itself.  A separate process: the sweep sets driver switches in its environment."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_sym.py"), "120", "100"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
