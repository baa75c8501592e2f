"""-m gpu: several host threads solve at once, each with its own problem (own HIP stream; device blocks, streams and pinned
blocks come from the process-wide pools).  Kernels of different problems share the CUs, so the waves of a workgroup drift apart
and the pools change hands under load: every thread must reproduce the solo run bit for bit (tools/concurrent_stress.py is
the long version; tools/batch_stress.py puts thousands of workgroups of ONE launch in flight)."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(gpu, g, kw):
    prob, poses = gpu.problem_from_graph(g)
    s = gpu.solve(gpu.SolverOptions(**kw), prob)
    return (tuple(float(c) for c in s.iterations["cost"]), tuple(int(c) for c in s.iterations["linear_solver_iterations"]),
            poses.tobytes(), s.c.factor_kind, s.cg_form)


@pytest.mark.parametrize("case", ["kitti00_exact", "manhattan_exact", "manhattan_pcg"])
def test_eight_threads_reproduce_the_solo_run(gpu, ds, case):
    if case == "kitti00_exact":
        k = np.load(os.path.join(G, "kitti00.npz"))
        g = ds.PoseGraphData(k["origin"], k["ia"], k["ib"], k["meas"], None)
        kw = dict(max_num_iterations=100, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    elif case == "manhattan_exact":
        g = ds.manhattan_se3(2000, 6000, seed=4)
        kw = dict(max_num_iterations=8, linear_solver_type=gpu.SPARSE_NORMAL_CHOLESKY)
    else:
        g = ds.manhattan_se3(10000, 40000, seed=6)      # (BASELINE configs[1] size: the library takes the resident stream where it can)
        kw = dict(max_num_iterations=10, linear_solver_type=gpu.BLOCK_JACOBI_PCG, eta=0.1, max_linear_solver_iterations=500)
    solo = _run(gpu, g, kw)
    # PCG left to the library: ONE session per device runs the resident stream (a grid barrier needs its grid on the chip), whoever asks
    # meanwhile the fused one — same algorithm, same decisions, not the same bits.  Every thread must reproduce, bit for bit, the solo run
    # of the stream it got (Summary::cg_form says which; pcg_form 2 asks for the fused stream outright).
    solo_of = {solo[4]: solo}
    if solo[4] == 4:          # (a graph whose grid the resident stream takes; smaller ones run one stream whoever asks)
        solo_of[3] = _run(gpu, g, dict(kw, pcg_form=2))
        assert solo_of[3][4] == 3 and solo_of[3][1] == solo[1] and np.allclose(solo_of[3][0], solo[0], rtol=1e-7)
    n = 8
    out, err = [None] * n, []
    bar = threading.Barrier(n)

    def work(i):
        try:
            bar.wait()
            out[i] = _run(gpu, g, kw)
        except Exception as e:  # noqa: BLE001
            err.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not err
    assert all(o == solo_of[o[4]] for o in out)
