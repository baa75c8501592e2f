"""CPU: the trust-region rules the device and the host driver share (csrc/pgo_lm_rules.h, through the host-only hook
pgo_debug_lm_decide) against the ORACLE's own Levenberg-Marquardt loop: its committed traces (tests/golden/c2_exact_trace.npz: 248
iterations of BASELINE configs[1] with the reference's options, saw-tooth radius, 60 rejected steps; kitti00_trace.npz) are replayed
record by record — candidate cost and model change reconstructed from the logged cost change and relative decrease — and every
accept / reject decision and every trust-region radius must come out as the oracle logged them.  No GPU."""
import ctypes as C
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _replay(pkg, trace, radius0=1e4):
    L = pkg.lib()
    opt = pkg.SolverOptions(max_num_iterations=1000)
    state = (C.c_double * 4)(radius0, 2.0, float(trace[0, 1]), 1.0)
    checked = 0
    for k in range(1, len(trace)):
        it, cost, dcost, gmax, step_norm, rho, radius, lin, ok = trace[k]
        x_cost = state[2]
        if rho == 0.0 or dcost == 0.0:          # an invalid step (no model change logged): not replayable from the trace
            state[0] = radius
            continue
        cand = x_cost - dcost
        step = (C.c_double * 4)(cand, dcost / rho, step_norm * step_norm, 1e6)
        out = (C.c_double * 6)()
        assert L.pgo_debug_lm_decide(C.byref(opt.c), state, step, 0, out) == 0
        assert int(out[1]) == int(ok), (k, rho, list(out))
        assert out[0] in (4.0, 5.0)
        assert out[2] == pytest.approx(rho, rel=1e-9)
        assert state[0] == pytest.approx(radius, rel=1e-9), (k, state[0], radius)
        assert state[2] == pytest.approx(cost if ok else x_cost, rel=1e-12)
        checked += 1
    return checked


def test_rules_reproduce_the_oracles_decisions_on_c2(pkg):
    z = np.load(os.path.join(G, "c2_exact_trace.npz"))
    tr = z["trace"]
    n = _replay(pkg, tr)
    assert n >= 240 and int((tr[:, 8] == 0).sum()) >= 40         # rejected steps are part of what is replayed


def test_rules_reproduce_the_oracles_decisions_on_kitti00(pkg):
    z = np.load(os.path.join(G, "kitti00_trace.npz"))
    key = [k for k in z.files if "trace" in k][0]
    assert _replay(pkg, z[key]) >= 8


def test_termination_outcomes(pkg):
    L = pkg.lib()
    opt = pkg.SolverOptions()
    out = (C.c_double * 6)()
    # function tolerance on the candidate: |cost change| <= 1e-6 * cost, the step is not applied
    st = (C.c_double * 4)(1e4, 2.0, 100.0, 1.0)
    assert L.pgo_debug_lm_decide(C.byref(opt.c), st, (C.c_double * 4)(100.0 - 5e-5, 1e-4, 1.0, 1.0), 0, out) == 0
    assert out[0] == 3.0 and st[2] == 100.0 and st[0] == 1e4
    # parameter tolerance: |step| <= 1e-8 (|x| + 1e-8)
    st = (C.c_double * 4)(1e4, 2.0, 100.0, 1.0)
    L.pgo_debug_lm_decide(C.byref(opt.c), st, (C.c_double * 4)(50.0, 60.0, 1e-20, 4.0), 0, out)
    assert out[0] == 2.0
    # invalid step (model change <= 0, or a failed linear solve): radius halved, nothing accepted
    st = (C.c_double * 4)(1e4, 2.0, 100.0, 1.0)
    L.pgo_debug_lm_decide(C.byref(opt.c), st, (C.c_double * 4)(50.0, -1.0, 1.0, 1.0), 0, out)
    assert out[0] == 0.0 and st[0] == 5e3 and st[2] == 100.0
    st = (C.c_double * 4)(1e4, 2.0, 100.0, 1.0)
    L.pgo_debug_lm_decide(C.byref(opt.c), st, (C.c_double * 4)(50.0, 60.0, 1.0, 1.0), 2, out)
    assert out[0] == 0.0
    # rho = 1: radius / max(1/3, 1 - 1) = 3 x; rejection: radius / 2, then / 4
    st = (C.c_double * 4)(1e4, 2.0, 100.0, 1.0)
    L.pgo_debug_lm_decide(C.byref(opt.c), st, (C.c_double * 4)(40.0, 60.0, 1.0, 1.0), 0, out)
    assert out[0] == 4.0 and st[0] == 3e4 and st[2] == 40.0
    st = (C.c_double * 4)(1e4, 2.0, 100.0, 1.0)
    L.pgo_debug_lm_decide(C.byref(opt.c), st, (C.c_double * 4)(120.0, 60.0, 1.0, 1.0), 0, out)
    assert out[0] == 5.0 and st[0] == 5e3 and st[1] == 4.0
    L.pgo_debug_lm_decide(C.byref(opt.c), st, (C.c_double * 4)(120.0, 60.0, 1.0, 1.0), 0, out)
    assert st[0] == 1250.0 and st[1] == 8.0
