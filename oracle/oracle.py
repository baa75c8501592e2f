"""ctypes loader for the CPU oracle (oracle/pgo_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (posegraph-ceres_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TRACE_COLS = 9
TERMINATION = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}
REASON = {1: "function_tolerance", 2: "parameter_tolerance", 3: "gradient_tolerance", 4: "min_radius",
          5: "max_iterations", 6: "invalid_steps", 7: "linear_solver_failure"}


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int), ("linear_solver", C.c_int), ("jacobi_scaling", C.c_int),
        ("max_linear_solver_iterations", C.c_int), ("min_linear_solver_iterations", C.c_int),
        ("residual_reset_period", C.c_int), ("max_num_consecutive_invalid_steps", C.c_int),
        ("loss_kind", C.c_int), ("loss_a", C.c_double), ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("eta", C.c_double),
        ("pcg_cluster", C.c_int), ("num_threads", C.c_int), ("pcg_form", C.c_int), ("reserved", C.c_int),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int), ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
        ("num_iterations", C.c_int), ("num_linear_iterations", C.c_int), ("reason", C.c_int),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("total_seconds", C.c_double),
        ("linear_solver_seconds", C.c_double), ("jacobian_seconds", C.c_double), ("cost_eval_seconds", C.c_double),
        ("factor_nnz_blocks", C.c_longlong), ("factor_flops", C.c_double),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libpgo_oracle.so")
    src = os.path.join(_HERE, "pgo_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpgo_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_cost.restype = C.c_double
        _LIB.oracle_evaluate.restype = C.c_double
        _LIB.oracle_normal_equations_dense.restype = C.c_double
        _LIB.oracle_time_jacobian_eval.restype = C.c_double
    return _LIB


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def set_coarse_cuts(cuts=None):
    """Row shards for the two-level PCG (pcg_cluster <= -8): aggregates are formed inside every segment [cuts[r], cuts[r + 1]) — what the
    product's sharded solve does (pgo_row_shard_cuts gives the cuts).  None: one segment.  Process-wide: reset it when done."""
    if cuts is None:
        lib().oracle_set_coarse_cuts(None, C.c_int(0))
    else:
        arr = (C.c_longlong * len(cuts))(*[int(c) for c in cuts])
        lib().oracle_set_coarse_cuts(arr, C.c_int(len(cuts) - 1))


def default_options(**kw):
    o = Options()
    lib().oracle_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def edge_eval(pa, qa, pb, qb, mp, mq, L=None, mode="analytic"):
    fn = lib().oracle_edge_eval_analytic if mode == "analytic" else lib().oracle_edge_eval_autodiff
    r, Ja, Jb = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
    args = [_f64(x) for x in (pa, qa, pb, qb, mp, mq)]
    Lc = _f64(L)
    fn(*[_p(a) for a in args], _p(Lc), _p(r), _p(Ja), _p(Jb))
    return r, Ja, Jb


def quat_plus(q, delta):
    out = np.zeros(4)
    q, delta = _f64(q), _f64(delta)
    lib().oracle_quat_plus(_p(q), _p(delta), _p(out))
    return out


def loss(kind, a, s):
    rho = np.zeros(3)
    lib().oracle_loss(C.c_int(kind), C.c_double(a), C.c_double(s), _p(rho))
    return rho


def chol6(info):
    L = np.zeros((6, 6))
    info = _f64(info)
    rc = lib().oracle_chol6(_p(info), _p(L))
    if rc != 0:
        raise ValueError("information matrix not positive definite")
    return L


class Graph:
    """Plain arrays describing a pose graph (the oracle's and the product's common test input)."""

    def __init__(self, poses, ia, ib, meas, sqrt_info=None, cmask=None):
        self.poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
        self.ia = np.ascontiguousarray(ia, dtype=np.int32)
        self.ib = np.ascontiguousarray(ib, dtype=np.int32)
        self.meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 7)
        self.sqrt_info = None if sqrt_info is None else np.ascontiguousarray(sqrt_info, dtype=np.float64).reshape(-1, 36)
        n = self.poses.shape[0]
        if cmask is None:
            cmask = np.zeros(n, dtype=np.uint8)
            cmask[0] = 3  # finial.cpp:525-527: first pose constant
        self.cmask = np.ascontiguousarray(cmask, dtype=np.uint8)

    @property
    def N(self):
        return self.poses.shape[0]

    @property
    def E(self):
        return self.ia.shape[0]

    def _args(self, poses=None):
        poses = self.poses if poses is None else np.ascontiguousarray(poses, dtype=np.float64)
        return (C.c_int(self.N), C.c_int(self.E), _p(poses), _p(self.cmask, C.c_uint8), _p(self.ia, C.c_int),
                _p(self.ib, C.c_int), _p(self.meas), _p(self.sqrt_info))


def cost(g, poses=None, loss_kind=1, loss_a=1.0):
    return lib().oracle_cost(*g._args(poses), C.c_int(loss_kind), C.c_double(loss_a))


def evaluate(g, poses=None, loss_kind=1, loss_a=1.0):
    r = np.zeros((g.E, 6))
    Ja = np.zeros((g.E, 6, 6))
    Jb = np.zeros((g.E, 6, 6))
    c = lib().oracle_evaluate(*g._args(poses), C.c_int(loss_kind), C.c_double(loss_a), _p(r), _p(Ja), _p(Jb))
    return c, r, Ja, Jb


def normal_equations_dense(g, poses=None, loss_kind=1, loss_a=1.0):
    m = 6 * g.N
    H = np.zeros((m, m))
    grad = np.zeros(m)
    c = lib().oracle_normal_equations_dense(*g._args(poses), C.c_int(loss_kind), C.c_double(loss_a), _p(H), _p(grad))
    return c, H, grad


def linear_solve(g, d2, b, linear_solver=0, q_tol=0.1, max_it=500, poses=None, loss_kind=1, loss_a=1.0):
    x = np.zeros(6 * g.N)
    d2, b = _f64(d2), _f64(b)
    it = lib().oracle_linear_solve(*g._args(poses), C.c_int(loss_kind), C.c_double(loss_a), _p(d2), _p(b),
                                   C.c_int(linear_solver), C.c_double(q_tol), C.c_int(max_it), _p(x))
    return x, it


def solve(g, options=None, trace_capacity=2048):
    """Runs the LM loop on a COPY of g.poses.  Returns (final_poses, Summary, trace ndarray)."""
    o = options or default_options()
    poses = g.poses.copy()
    s = Summary()
    trace = np.zeros((trace_capacity, TRACE_COLS))
    lib().oracle_solve(C.c_int(g.N), C.c_int(g.E), _p(poses), _p(g.cmask, C.c_uint8), _p(g.ia, C.c_int),
                       _p(g.ib, C.c_int), _p(g.meas), _p(g.sqrt_info), C.byref(o), C.byref(s), _p(trace),
                       C.c_int(trace_capacity))
    return poses, s, trace[: min(s.num_iterations, trace_capacity)].copy()


def time_jacobian_eval(g, repeats=1, loss_kind=1, loss_a=1.0):
    chk = C.c_double(0)
    t = lib().oracle_time_jacobian_eval(*g._args(), C.c_int(loss_kind), C.c_double(loss_a), C.c_int(repeats), C.byref(chk))
    return t


# ---- MotionEstimate reprojection problem ----
def reproj_eval(points, obs, intr, q, t):
    """Residuals (n,2) and local Jacobians (n,2,6: columns dtheta | dt) of the reprojection blocks (autodiff chain)."""
    points, obs, intr, q, t = (_f64(a) for a in (points, obs, intr, q, t))
    n = points.shape[0]
    r, J = np.zeros((n, 2)), np.zeros((n, 2, 6))
    lib().oracle_reproj_eval(C.c_int(n), _p(points), _p(obs), _p(intr), _p(q), _p(t), _p(r), _p(J))
    return r, J


def reproj_solve(points, obs, intr, q, t, cmask=2, options=None, trace_capacity=1100):
    """Ceres-style LM on one (q, t) pair; cmask bit0: t constant, bit1: q constant (the reference's setting).
    Returns (q, t, Summary, trace)."""
    o = options or default_options(max_num_iterations=1000)
    points, obs, intr = _f64(points), _f64(obs), _f64(intr)
    q, t = _f64(q).copy(), _f64(t).copy()
    s = Summary()
    trace = np.zeros((trace_capacity, TRACE_COLS))
    lib().oracle_reproj_solve(C.c_int(points.shape[0]), _p(points), _p(obs), _p(intr), _p(q), _p(t), C.c_int(cmask),
                              C.byref(o), C.byref(s), _p(trace), C.c_int(trace_capacity))
    return q, t, s, trace[: min(s.num_iterations, trace_capacity)].copy()
