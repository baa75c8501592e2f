"""posegraph-ceres_amd — Python host-side mirror of the C ABI in include/pgo.h (libpgo_hip.so).

The compute path is hand-written HIP for gfx950 (csrc/); this module only marshals arrays through
ctypes.  It mirrors the reference's use of Ceres (REF = src/POSE_GRAPH_CERES_PLUS):

    problem = Problem()                                   # ceres::Problem            finial.cpp:58
    problem.add_poses(poses)                              # parameter blocks p(3), q(4 xyzw)
    problem.add_se3_between(ia, ib, t_be, sqrt_info)      # PoseGraph3dErrorTerm::Create + AddResidualBlock
    problem.set_loss(HUBER, 1.0)                          # new HuberLoss(1.0)        finial.cpp:495
    problem.set_pose_constant(0)                          # SetParameterBlockConstant finial.cpp:525-527
    summary = solve(SolverOptions(max_num_iterations=1000,
                                  linear_solver_type=SPARSE_NORMAL_CHOLESKY), problem)   # finial.cpp:534-539
    print(summary.full_report()); summary.is_solution_usable()

The directory name contains a hyphen, so import it through `pgo_loader.load()` (repo root).
There is no CPU fallback: without the built library or without a GPU every compute call raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpgo_hip.so")

TRIVIAL, HUBER, SOFT_L_ONE, CAUCHY, ARCTAN, SWITCHABLE = 0, 1, 2, 3, 4, 5
SPARSE_NORMAL_CHOLESKY, BLOCK_JACOBI_PCG = 0, 1
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
TERMINATION_NAMES = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}

OK, ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_NUMERICAL = 0, -1, -2, -3, -4, -5

# every symbol include/pgo.h declares (tests check the built library exports all of them)
C_ABI_SYMBOLS = [
    "pgo_version", "pgo_last_error", "pgo_device_count", "pgo_set_device", "pgo_problem_create",
    "pgo_problem_add_point", "pgo_problem_add_points", "pgo_problem_add_point_observation_batch",
    "pgo_problem_destroy", "pgo_problem_add_pose", "pgo_problem_add_poses", "pgo_problem_add_se3_between",
    "pgo_problem_add_se3_between_batch", "pgo_problem_set_loss", "pgo_problem_set_pose_constant",
    "pgo_problem_set_parameter_block_constant", "pgo_problem_num_poses", "pgo_problem_num_edges",
    "pgo_solver_options_init", "pgo_solve", "pgo_summary_is_solution_usable", "pgo_summary_full_report",
    "pgo_evaluate", "pgo_normal_equations", "pgo_linear_solve", "pgo_plus", "pgo_solver_begin",
    "pgo_solver_step", "pgo_solver_reset", "pgo_solver_end", "pgo_time_kernel", "pgo_solver_trace_start", "pgo_solver_trace_read", "pgo_solver_cg_form", "pgo_solver_exchange_doubles", "pgo_shard_range",
    "pgo_comm_get_unique_id", "pgo_comm_init", "pgo_comm_init_ipc", "pgo_debug_comm_stress", "pgo_debug_lm_decide", "pgo_loopback_create", "pgo_loopback_destroy", "pgo_comm_init_loopback",
    "pgo_generate_candidates", "pgo_reproj_options_init", "pgo_reproj_solve_batch",
    "pgo_row_shard_range", "pgo_row_shard_cuts", "pgo_read_trajectory", "pgo_build_odometry_edges", "pgo_edge_rules_init", "pgo_build_edges",
    "pgo_solve_batch", "pgo_release_device_memory", "pgo_tuning_set", "pgo_tuning_get", "pgo_tuning_describe",
]


class PgoError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("pgo error %d: %s" % (code, message))
        self.code = code


class _COptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int), ("linear_solver_type", C.c_int), ("jacobi_scaling", C.c_int),
        ("max_linear_solver_iterations", C.c_int), ("min_linear_solver_iterations", C.c_int),
        ("max_num_consecutive_invalid_steps", C.c_int), ("cg_batch", C.c_int), ("pcg_cluster_poses", C.c_int),
        ("cg_residual_reset_period", C.c_int), ("pcg_form", C.c_int), ("pcg_coarse_aggregate", C.c_int), ("reserved_options", C.c_int),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("eta", C.c_double),
        ("exact_r_tolerance", C.c_double),
    ]


class _CSummary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int), ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
        ("num_iterations", C.c_int), ("num_linear_solver_iterations", C.c_int), ("num_poses", C.c_int),
        ("num_edges", C.c_int), ("reason", C.c_int), ("linear_solver_used", C.c_int), ("factor_nnz_blocks", C.c_int),
        ("factor_levels", C.c_int), ("num_factorizations", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("total_time_in_seconds", C.c_double), ("setup_time_in_seconds", C.c_double),
        ("linear_solver_time_in_seconds", C.c_double), ("jacobian_evaluation_time_in_seconds", C.c_double),
        ("residual_evaluation_time_in_seconds", C.c_double), ("final_gradient_max_norm", C.c_double),
        ("final_trust_region_radius", C.c_double), ("message", C.c_char * 256),
        ("factor_kind", C.c_int), ("factor_max_front", C.c_int), ("factor_flops", C.c_double),
        ("num_parameter_blocks_reduced", C.c_int), ("num_parameters_reduced", C.c_int), ("num_effective_parameters_reduced", C.c_int),
        ("cg_form", C.c_int), ("cg_exchange", C.c_int), ("sym_form", C.c_int), ("coarse_level", C.c_int),
    ]


class _CRecord(C.Structure):
    _fields_ = [
        ("iteration", C.c_int), ("step_is_successful", C.c_int), ("linear_solver_iterations", C.c_int),
        ("reserved", C.c_int), ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double), ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double),
    ]


RECORD_DTYPE = np.dtype([
    ("iteration", np.int32), ("step_is_successful", np.int32), ("linear_solver_iterations", np.int32),
    ("reserved", np.int32), ("cost", np.float64), ("cost_change", np.float64), ("gradient_max_norm", np.float64),
    ("step_norm", np.float64), ("relative_decrease", np.float64), ("trust_region_radius", np.float64)])

_lib = None


def build(force=False):
    """Compiles csrc/ into libpgo_hip.so with hipcc --offload-arch=gfx950 (works without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src) if os.path.isfile(os.path.join(src, f)) and not f.startswith(".")]
    deps.append(os.path.join(_HERE, "..", "include", "pgo.h"))
    def stale():
        return (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps)
    if force or stale():
        # one make at a time in this directory: the rank processes of a multi-process test / bench all call build() (r06)
        import fcntl
        with open(os.path.join(src, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():          # (somebody else may have built it while this process waited for the lock)
                subprocess.check_call(["make", "-j8", "-C", src] + (["-B"] if force else []))
                # make can answer "up to date" while a source is newer than the library: a header named in the Makefile that no longer
                # exists switches its pattern rules off without a word (r05: the library silently stopped rebuilding).  Rebuild
                # everything once, and refuse to hand out a library older than its sources.
                if stale():
                    subprocess.check_call(["make", "-j8", "-B", "-C", src])
                if stale():
                    raise RuntimeError("%s is older than its sources after make -B: check the HDRS list of csrc/Makefile" % LIB_PATH)
    return LIB_PATH


def kernel_source_sha():
    """sha256 (16 hex digits) of everything the library is built from: every file of csrc/ in name order + the Makefile (its FLAGS).
    bench.py quotes a committed profile only when it was taken on these very sources (tools/rocprof_pmc.py records the same value)."""
    import hashlib
    src = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(src)):
        fp = os.path.join(src, f)
        if os.path.isfile(fp) and not f.startswith(".") and (f.endswith((".hip", ".cpp", ".h", ".inc")) or f == "Makefile"):
            h.update(f.encode())
            h.update(open(fp, "rb").read())
    return h.hexdigest()[:16]


def lib():
    """Loads libpgo_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PgoError(ERR_NO_DEVICE, "%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                                          "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.pgo_last_error.restype = C.c_char_p
        L.pgo_problem_create.restype = C.c_void_p
        L.pgo_problem_destroy.argtypes = [C.c_void_p]
        L.pgo_summary_full_report.restype = C.c_size_t
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise PgoError(rc, lib().pgo_last_error().decode())
    return rc


def device_count():
    return lib().pgo_device_count()


def set_device(i):
    _check(lib().pgo_set_device(C.c_int(i)))


def shard_range(n, rank, world):
    b, e = C.c_longlong(0), C.c_longlong(0)
    _check(lib().pgo_shard_range(C.c_longlong(n), C.c_int(rank), C.c_int(world), C.byref(b), C.byref(e)))
    return b.value, e.value


def row_shard_cuts(n_poses, ia, ib, world):
    """pgo_row_shard_cuts: THE row ownership rule of the sharded solve (r06: shares cut where the incidence slots balance).
    Returns (cut, rows_per): rank r owns the poses [cut[r], cut[r + 1])."""
    import numpy as np
    a = np.ascontiguousarray(ia, dtype=np.int32)
    b = np.ascontiguousarray(ib, dtype=np.int32)
    cut = (C.c_longlong * (world + 1))()
    rp = C.c_int(0)
    _check(lib().pgo_row_shard_cuts(C.c_longlong(n_poses), C.c_longlong(len(a)), a.ctypes.data_as(C.POINTER(C.c_int)),
                                    b.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(world), cut, C.byref(rp)))
    return [int(c) for c in cut], rp.value


def row_shard_range(n_poses, rank, world):
    """pgo_row_shard_range: the row ownership rule of the sharded solve.  Returns (begin, end, rows_per)."""
    b, e, rp = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
    _check(lib().pgo_row_shard_range(C.c_longlong(n_poses), C.c_int(rank), C.c_int(world), C.byref(b), C.byref(e), C.byref(rp)))
    return b.value, e.value, rp.value


def generate_candidates(xyz, search_radius=6.0, gap=100, return_ms=False):
    """GPU loop-closure candidate search (pgo_generate_candidates; role of
    generate_edges_from_trajectory_origion.cpp:58-111).  Returns {frame id: [id-1, matches...]} for id >= 1, the same
    structure as datasets.generate_candidates / read_candidates."""
    import numpy as np
    p = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32).reshape(-1, 3))
    n = p.shape[0]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    fp = p.ctypes.data_as(C.POINTER(C.c_float))
    rp = row_ptr.ctypes.data_as(C.POINTER(C.c_longlong))
    _check(lib().pgo_generate_candidates(fp, C.c_int(n), C.c_float(search_radius), C.c_int(gap), rp, None, C.c_longlong(0), None))
    idx = np.zeros(max(1, int(row_ptr[n])), dtype=np.int32)
    ms = C.c_double(0)
    _check(lib().pgo_generate_candidates(fp, C.c_int(n), C.c_float(search_radius), C.c_int(gap), rp,
                                         idx.ctypes.data_as(C.POINTER(C.c_int)), C.c_longlong(idx.shape[0]), C.byref(ms)))
    out = {k: idx[row_ptr[k]:row_ptr[k + 1]].tolist() for k in range(1, n)}
    return (out, ms.value) if return_ms else out


def read_trajectory(path, fmt):
    """pgo_read_trajectory: GroundTruth::loadPoses1 (fmt 1, with the reference's quaternion scramble) / loadPoses2 (fmt 2, KITTI
    3x4).  Returns Twc (n, 4, 4) float64 holding the float32 values the reference keeps.  Host only (no GPU needed)."""
    import numpy as np
    cnt = C.c_int(0)
    _check(lib().pgo_read_trajectory(str(path).encode(), C.c_int(fmt), None, C.c_int(0), C.byref(cnt)))
    out = np.zeros((max(1, cnt.value), 16))
    _check(lib().pgo_read_trajectory(str(path).encode(), C.c_int(fmt), _dp(out), C.c_int(cnt.value), C.byref(cnt)))
    return out[:cnt.value].reshape(-1, 4, 4)


def build_odometry_edges(Twc, return_ms=False):
    """pgo_build_odometry_edges: t_be of every consecutive frame pair on the GPU, (n-1, 7)."""
    import numpy as np
    T = np.ascontiguousarray(np.asarray(Twc, dtype=np.float64).reshape(-1, 16))
    n = T.shape[0]
    out = np.zeros((max(1, n - 1), 7))
    ms = C.c_double(0)
    _check(lib().pgo_build_odometry_edges(C.c_int(n), _dp(T), _dp(out), C.byref(ms)))
    out = out[:max(0, n - 1)]
    return (out, ms.value) if return_ms else out


class PairObservation(C.Structure):
    """Mirror of pgo_pair_observation."""
    _fields_ = [("nmatches", C.c_int), ("inliers", C.c_int), ("rvec", C.c_double * 3), ("tvec", C.c_double * 3)]


class EdgeRules(C.Structure):
    """Mirror of pgo_edge_rules; defaults = finial.cpp:226, 234, 285."""
    _fields_ = [("match_threshold", C.c_int), ("inlier_threshold", C.c_int), ("norm_threshold", C.c_double),
                ("loop_list_gap", C.c_int), ("reserved", C.c_int)]

    def __init__(self, **kw):
        super().__init__()
        lib().pgo_edge_rules_init(C.byref(self))
        for k, v in kw.items():
            setattr(self, k, v)


def build_edges(Twc, candidates, observations=None, rules=None):
    """pgo_build_edges: checkFrame's rules over recorded front-end results.  candidates: {frame: [ids...]} (file order);
    observations(cur, prev) -> None or dict(nmatches, inliers, rvec, tvec) (as loop_edges.LoopEdgeBuilder takes).
    Returns (id_begin, id_end, t_be (E,7), loop_list (L,2))."""
    import numpy as np
    T = np.ascontiguousarray(np.asarray(Twc, dtype=np.float64).reshape(-1, 16))
    n = T.shape[0]
    ptr = np.zeros(n + 1, dtype=np.int64)
    flat = []
    for f in range(n):
        c = list(candidates.get(f, ()))
        flat.extend(c)
        ptr[f + 1] = len(flat)
    idx = np.asarray(flat if flat else [0], dtype=np.int32)
    obs = (PairObservation * max(1, len(flat)))()
    if observations is not None:
        k = 0
        for f in range(n):
            for prev in candidates.get(f, ()):
                o = observations(f, prev) if f - prev > 1 else None
                if o is not None:
                    obs[k].nmatches, obs[k].inliers = int(o["nmatches"]), int(o["inliers"])
                    obs[k].rvec[:] = [float(x) for x in o["rvec"]]
                    obs[k].tvec[:] = [float(x) for x in o["tvec"]]
                k += 1
    rules = rules or EdgeRules()
    ne, nl = C.c_longlong(0), C.c_longlong(0)
    cap = len(flat) + 1
    ia, ib, m = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32), np.zeros((cap, 7))
    ll = np.zeros((cap, 2), dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    _check(lib().pgo_build_edges(C.c_int(n), _dp(T), ptr.ctypes.data_as(C.POINTER(C.c_longlong)), ip(idx),
                                 obs if observations is not None else None, C.byref(rules), ip(ia), ip(ib), _dp(m), C.c_longlong(cap),
                                 C.byref(ne), ip(ll), C.c_longlong(cap), C.byref(nl)))
    return ia[:ne.value], ib[:ne.value], m[:ne.value], ll[:nl.value]


class ReprojOptions(C.Structure):
    """Mirror of pgo_reproj_options (include/pgo.h); defaults = MotionEstimate.cc:71-125."""
    _fields_ = [("max_num_iterations", C.c_int), ("q_constant", C.c_int), ("t_constant", C.c_int), ("loss_kind", C.c_int),
                ("jacobi_scaling", C.c_int), ("max_num_consecutive_invalid_steps", C.c_int), ("loss_a", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double)]

    def __init__(self, **kw):
        super().__init__()
        lib().pgo_reproj_options_init(C.byref(self))
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


REPROJ_SUMMARY_DTYPE = None


def reproj_solve_batch(point_ptr, points, observations, intrinsics, q, t, options=None, return_ms=False):
    """Batched MotionEstimate solves (pgo_reproj_solve_batch).  q (n,4 xyzw) and t (n,3) are updated IN PLACE.
    Returns a structured array of per-problem summaries (and the kernel time in ms when return_ms)."""
    import numpy as np
    global REPROJ_SUMMARY_DTYPE
    if REPROJ_SUMMARY_DTYPE is None:
        REPROJ_SUMMARY_DTYPE = np.dtype([("termination_type", "i4"), ("reason", "i4"), ("num_iterations", "i4"),
                                         ("num_successful_steps", "i4"), ("num_unsuccessful_steps", "i4"), ("num_points", "i4"),
                                         ("initial_cost", "f8"), ("final_cost", "f8")])
    ptr = np.ascontiguousarray(point_ptr, dtype=np.int64)
    n = ptr.shape[0] - 1
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    ob = np.ascontiguousarray(observations, dtype=np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(intrinsics, dtype=np.float64).reshape(4)
    for a, w in ((q, 4), (t, 3)):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.shape == (n, w)):
            raise ValueError("q / t must be C-contiguous float64 arrays of shape (n_problems, 4) / (n_problems, 3)")
    o = options or ReprojOptions()
    summ = np.zeros(max(n, 1), dtype=REPROJ_SUMMARY_DTYPE)
    ms = C.c_double(0)
    _check(lib().pgo_reproj_solve_batch(C.c_int(n), ptr.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(pts), _dp(ob), _dp(K), _dp(q), _dp(t),
                                        C.byref(o), summ.ctypes.data_as(C.c_void_p), C.byref(ms)))
    return (summ[:n], ms.value) if return_ms else summ[:n]


def comm_unique_id():
    buf = (C.c_ubyte * 128)()
    _check(lib().pgo_comm_get_unique_id(buf))
    return bytes(buf)


def tuning_set(name, value=None):
    """pgo_tuning_set: a development / test knob of the library (csrc/pgo_tuning.h; until r06 these were environment variables).
    value None puts it back to its default."""
    _check(lib().pgo_tuning_set(name.encode(), C.c_double(float("nan") if value is None else float(value))))


def tuning_get(name):
    """(value or None when the knob is at its default)"""
    v, isset = C.c_double(0.0), C.c_int(0)
    _check(lib().pgo_tuning_get(name.encode(), C.byref(v), C.byref(isset)))
    return v.value if isset.value else None


def tuning_knobs():
    """{name: one line on what the knob does}"""
    out, i = {}, 0
    name, what = C.c_char_p(), C.c_char_p()
    n = lib().pgo_tuning_describe(C.c_int(0), C.byref(name), C.byref(what))
    while i < n:
        lib().pgo_tuning_describe(C.c_int(i), C.byref(name), C.byref(what))
        out[name.value.decode()] = what.value.decode()
        i += 1
    return out


class tuning:
    """with gpu.tuning(sym_repack=1, sym_rows=64): ...   — knobs set for the block, back to what they were behind it"""

    def __init__(self, **knobs):
        self.new = knobs

    def __enter__(self):
        self.old = {k: tuning_get(k) for k in self.new}
        for k, v in self.new.items():
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning_set(k, v)
        return False


def loopback_create(world):
    lib().pgo_loopback_create.restype = C.c_void_p
    return lib().pgo_loopback_create(C.c_int(world))


def loopback_destroy(group):
    lib().pgo_loopback_destroy(C.c_void_p(group))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


class SolverOptions:
    """ceres::Solver::Options (fields the path reads)."""

    def __init__(self, **kw):
        self.c = _COptions()
        lib().pgo_solver_options_init(C.byref(self.c))
        for k, v in kw.items():
            setattr(self, k, v)

    def __getattr__(self, k):
        if k != "c" and hasattr(_COptions, k):
            return getattr(self.c, k)
        raise AttributeError(k)

    def __setattr__(self, k, v):
        if k == "c":
            object.__setattr__(self, k, v)
        elif hasattr(_COptions, k):
            setattr(self.c, k, v)
        else:
            raise AttributeError("unknown solver option %r" % k)


class Summary:
    """ceres::Solver::Summary."""

    def __init__(self, c, records):
        self.c = c
        self.iterations = records

    def __getattr__(self, k):
        if k not in ("c", "iterations") and hasattr(_CSummary, k):
            v = getattr(self.c, k)
            return v.decode() if isinstance(v, bytes) else v
        raise AttributeError(k)

    def is_solution_usable(self):
        return bool(lib().pgo_summary_is_solution_usable(C.byref(self.c)))

    def full_report(self):
        n = len(self.iterations)
        rec = (_CRecord * max(n, 1)).from_buffer_copy(self.iterations.tobytes() if n else bytes(C.sizeof(_CRecord)))
        need = lib().pgo_summary_full_report(C.byref(self.c), rec, C.c_int(n), None, C.c_size_t(0))
        buf = C.create_string_buffer(need)
        lib().pgo_summary_full_report(C.byref(self.c), rec, C.c_int(n), buf, C.c_size_t(need))
        return buf.value.decode()


class Problem:
    """ceres::Problem restricted to the reference's use: SE(3) poses + between-factor residual blocks."""

    def __init__(self):
        self._h = C.c_void_p(lib().pgo_problem_create())
        if not self._h:
            raise MemoryError("pgo_problem_create failed")
        self._keep = []       # parameter memory must outlive the problem (Ceres: user owns parameters)
        self.poses = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().pgo_problem_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def add_poses(self, poses):
        """poses: (N,7) float64 C-contiguous array px py pz qx qy qz qw; updated IN PLACE by solve()."""
        if not (isinstance(poses, np.ndarray) and poses.dtype == np.float64 and poses.ndim == 2 and poses.shape[1] == 7
                and poses.flags["C_CONTIGUOUS"]):
            raise ValueError("poses must be a C-contiguous float64 (N,7) array")
        self._keep.append(poses)
        if self.poses is None:
            self.poses = poses
        return _check(lib().pgo_problem_add_poses(self._h, C.c_int(poses.shape[0]), _dp(poses), C.c_int(7)))

    def add_se3_between(self, id_begin, id_end, t_be, sqrt_information=None):
        ia = np.ascontiguousarray(id_begin, dtype=np.int32).reshape(-1)
        ib = np.ascontiguousarray(id_end, dtype=np.int32).reshape(-1)
        t = np.ascontiguousarray(t_be, dtype=np.float64).reshape(-1, 7)
        si = None if sqrt_information is None else np.ascontiguousarray(sqrt_information, dtype=np.float64).reshape(-1, 36)
        if not (len(ia) == len(ib) == len(t)) or (si is not None and len(si) != len(ia)):
            raise ValueError("edge arrays disagree in length")
        return _check(lib().pgo_problem_add_se3_between_batch(self._h, C.c_int(len(ia)), _ip(ia), _ip(ib), _dp(t), _dp(si)))

    # ---- pose / landmark problems ----
    def add_points(self, points):
        """points: (M,3) float64 C-contiguous array of 3-D points (size-3 Euclidean blocks), updated IN PLACE by solve().
        Returns the node index of the first one (points share the index space of the poses)."""
        if not (isinstance(points, np.ndarray) and points.dtype == np.float64 and points.ndim == 2 and points.shape[1] == 3
                and points.flags["C_CONTIGUOUS"]):
            raise ValueError("points must be a C-contiguous float64 (M,3) array")
        self._keep.append(points)
        return _check(lib().pgo_problem_add_points(self._h, C.c_int(points.shape[0]), _dp(points), C.c_int(3)))

    def add_point_observations(self, pose, point, z, sqrt_information3=None):
        """z[i] = the point `point[i]` in the frame of pose `pose[i]` (3 numbers); residual L3 (R(q)^T (l - p) - z)."""
        ip = np.ascontiguousarray(pose, dtype=np.int32).reshape(-1)
        il = np.ascontiguousarray(point, dtype=np.int32).reshape(-1)
        zz = np.ascontiguousarray(z, dtype=np.float64).reshape(-1, 3)
        si = None if sqrt_information3 is None else np.ascontiguousarray(sqrt_information3, dtype=np.float64).reshape(-1, 9)
        if not (len(ip) == len(il) == len(zz)) or (si is not None and len(si) != len(ip)):
            raise ValueError("observation arrays disagree in length")
        return _check(lib().pgo_problem_add_point_observation_batch(self._h, C.c_int(len(ip)), _ip(ip), _ip(il), _dp(zz), _dp(si)))

    def set_loss(self, kind, a=1.0):
        _check(lib().pgo_problem_set_loss(self._h, C.c_int(kind), C.c_double(a)))

    def set_pose_constant(self, pose, which=3):
        _check(lib().pgo_problem_set_pose_constant(self._h, C.c_int(pose), C.c_int(which)))

    @property
    def num_poses(self):
        return lib().pgo_problem_num_poses(self._h)

    @property
    def num_edges(self):
        return lib().pgo_problem_num_edges(self._h)

    # ---- evaluation (Problem::Evaluate analogue) ----
    def evaluate(self, residuals=True, jacobians=True, gradient=True):
        N, E = self.num_poses, self.num_edges
        cost = C.c_double(0)
        r = np.zeros((E, 6)) if residuals else None
        ja = np.zeros((E, 6, 6)) if jacobians else None
        jb = np.zeros((E, 6, 6)) if jacobians else None
        g = np.zeros((N, 6)) if gradient else None
        _check(lib().pgo_evaluate(self._h, C.byref(cost), _dp(r), _dp(ja), _dp(jb), _dp(g)))
        return cost.value, r, ja, jb, g

    def normal_equations(self):
        N, E = self.num_poses, self.num_edges
        diag, off, g = np.zeros((N, 6, 6)), np.zeros((E, 6, 6)), np.zeros((N, 6))
        _check(lib().pgo_normal_equations(self._h, _dp(diag), _dp(off), _dp(g)))
        return diag, off, g

    def linear_solve(self, d2, b, options=None):
        options = options or SolverOptions()
        N = self.num_poses
        d2 = np.ascontiguousarray(d2, dtype=np.float64).reshape(N * 6)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(N * 6)
        x = np.zeros(N * 6)
        it = C.c_int(0)
        _check(lib().pgo_linear_solve(self._h, C.byref(options.c), _dp(d2), _dp(b), _dp(x), C.byref(it)))
        return x, it.value

    def plus(self, delta):
        delta = np.ascontiguousarray(delta, dtype=np.float64).reshape(self.num_poses * 6)
        _check(lib().pgo_plus(self._h, _dp(delta)))

    # ---- device-resident stepping (benchmarks) ----
    def solver_begin(self, options):
        _check(lib().pgo_solver_begin(self._h, C.byref(options.c)))

    def solver_step(self, n):
        """Runs up to n LM iterations on the device-resident state. Returns (executed, done)."""
        done, ran = C.c_int(0), C.c_int(0)
        _check(lib().pgo_solver_step(self._h, C.c_int(n), C.byref(ran), C.byref(done)))
        return ran.value, bool(done.value)

    def solver_reset(self):
        _check(lib().pgo_solver_reset(self._h))

    def solver_end(self, records_capacity=4096):
        s = _CSummary()
        rec = (_CRecord * records_capacity)()
        _check(lib().pgo_solver_end(self._h, C.byref(s), rec, C.c_int(records_capacity)))
        n = min(s.num_iterations, records_capacity)
        return Summary(s, np.frombuffer(bytes(rec), dtype=RECORD_DTYPE)[:n].copy())

    # ---- one process per GPU (or virtual ranks for tests) ----
    def comm_init(self, unique_id, rank, world):
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().pgo_comm_init(self._h, buf, C.c_int(rank), C.c_int(world)))

    def comm_init_ipc(self, name, rank, world):
        """One process per rank, exchange buffers mapped through hipIpc handles (include/pgo.h pgo_comm_init_ipc)."""
        _check(lib().pgo_comm_init_ipc(self._h, name.encode(), C.c_int(rank), C.c_int(world)))

    def comm_init_loopback(self, group, rank):
        _check(lib().pgo_comm_init_loopback(self._h, C.c_void_p(group), C.c_int(rank)))

    def cg_form(self):
        """Summary::cg_form of the running session (0 standard CG, 3 fused stream, 4 resident stream; 1 / 2 several ranks)."""
        return _check(lib().pgo_solver_cg_form(self._h))

    def exchange_doubles(self):
        """Doubles one rank contributes to the per-CG-iteration collective of the running session (pgo_solver_exchange_doubles)."""
        return _check(lib().pgo_solver_exchange_doubles(self._h))

    def trace_start(self, max_launches=20000):
        """Launch trace of the fused universal stream (include/pgo.h pgo_solver_trace_start); 0 stops recording."""
        _check(lib().pgo_solver_trace_start(self._h, C.c_int(max_launches)))

    def trace_read(self, capacity=20000):
        """-> (records [n][4] int64: operation, start tick, end tick (100 MHz device clock), phase stamps; host launches; host enqueue seconds)"""
        rec = np.zeros((capacity, 4), dtype=np.int64)
        host = (C.c_double * 2)()
        n = lib().pgo_solver_trace_read(self._h, rec.ctypes.data_as(C.POINTER(C.c_longlong)), C.c_int(capacity), host)
        if n < 0:
            _check(n)
        return rec[:n].copy(), int(host[0]), float(host[1])

    def time_kernel(self, name, repeats=100):
        ms = C.c_double(0)
        _check(lib().pgo_time_kernel(self._h, name.encode(), C.c_int(repeats), C.byref(ms)))
        return ms.value


def solve(options, problem, records_capacity=4096):
    """ceres::Solve(options, &problem, &summary): runs LM on the GPU, updates the pose arrays in place."""
    s = _CSummary()
    rec = (_CRecord * records_capacity)()
    _check(lib().pgo_solve(problem._h, C.byref(options.c), C.byref(s), rec, C.c_int(records_capacity)))
    n = min(s.num_iterations, records_capacity)
    return Summary(s, np.frombuffer(bytes(rec), dtype=RECORD_DTYPE)[:n].copy())


def release_device_memory():
    """Return the device blocks pooled from destroyed problems to the driver."""
    _check(lib().pgo_release_device_memory())


def solve_batch(options, problems, records_capacity=256):
    """pgo_solve_batch: several independent problems as the components of one block-diagonal problem, every LM decision per
    problem.  Returns one Summary per problem; every problem's pose array is updated in place."""
    n = len(problems)
    handles = (C.c_void_p * n)(*[p._h.value for p in problems])
    sums = (_CSummary * n)()
    rec = (_CRecord * (n * records_capacity))()
    _check(lib().pgo_solve_batch(handles, C.c_int(n), C.byref(options.c), sums, rec, C.c_int(records_capacity)))
    allrec = np.frombuffer(rec, dtype=RECORD_DTYPE).reshape(n, records_capacity)
    out = []
    for c in range(n):
        s = _CSummary.from_buffer_copy(sums[c])
        out.append(Summary(s, allrec[c, :min(s.num_iterations, records_capacity)].copy()))
    return out


def problem_from_graph(g, loss=HUBER, loss_a=1.0, constant_first=True):
    """Mirror of BuildOptimizationProblem (finial.cpp:491-528) for array inputs; g has poses/ia/ib/meas/sqrt_info.
    Returns (problem, poses_array_updated_in_place)."""
    poses = np.array(g.poses, dtype=np.float64, order="C", copy=True)
    p = Problem()
    p.add_poses(poses)
    p.add_se3_between(g.ia, g.ib, g.meas, g.sqrt_info)
    p.set_loss(loss, loss_a)
    if constant_first:
        p.set_pose_constant(0, 3)
    return p, poses
