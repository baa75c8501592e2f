"""-m gpu, needs TWO MI355X in one box: the row-sharded path over a real RCCL communicator with two processes, one per GPU
(SURVEY.md §8e).  Skipped (reported as skipped, not passed) on a one-GPU box — the development and round-end test boxes have
one GPU, so this file is the test a two-GPU maintainer runs; the same sharding logic is exercised on one GPU through the
loopback transport in tests/test_gpu_sharded.py.  Two worker processes (spawned with RANK / WORLD_SIZE / LOCAL_RANK, rendezvous
through a file holding the ncclUniqueId) solve Manhattan 20 k with cluster-Jacobi PCG; rank 0's result must reproduce the
single-rank solve: same accept/reject sequence, same CG iteration counts, costs to 1e-9, poses to 1e-7."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["PGO_ROOT"])
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
rank, world, idfile, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["PGO_IDFILE"], os.environ["PGO_OUT"]
pkg.set_device(rank)
if rank == 0:
    uid = pkg.comm_unique_id()
    with open(idfile + ".tmp", "wb") as f: f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120: raise SystemExit("no unique id")
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
g = ds.manhattan_se3(20000, 80000, seed=20260928)
prob, poses = pkg.problem_from_graph(g)
prob.comm_init(uid, rank, world)
s = pkg.solve(pkg.SolverOptions(max_num_iterations=12, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob)
np.savez(out + ".%d.npz" % rank, cost=s.iterations["cost"], ok=s.iterations["step_is_successful"],
         cg=s.iterations["linear_solver_iterations"], poses=poses)
'''


def test_two_rccl_ranks_reproduce_the_single_rank_solve(gpu, ds, tmp_path):
    if gpu.device_count() < 2:
        pytest.skip("needs two GPUs in one box (this one has %d): RCCL with more than one rank cannot be exercised here" % gpu.device_count())
    g = ds.manhattan_se3(20000, 80000, seed=20260928)
    prob, poses = gpu.problem_from_graph(g)
    ref = gpu.solve(gpu.SolverOptions(max_num_iterations=12, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PGO_ROOT=ROOT, WORLD_SIZE="2", PGO_IDFILE=str(tmp_path / "ncclid"), PGO_OUT=str(tmp_path / "out"),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.wait(timeout=max(1, 300 - (time.time() - t0)))
    assert [p.returncode for p in procs] == [0, 0]
    outs = [np.load(str(tmp_path / ("out.%d.npz" % r))) for r in range(2)]
    for o in outs:
        assert list(o["ok"]) == list(ref.iterations["step_is_successful"])
        assert list(o["cg"]) == list(ref.iterations["linear_solver_iterations"])
        assert np.allclose(o["cost"], ref.iterations["cost"], rtol=1e-9)
        assert np.abs(o["poses"] - poses).max() < 1e-7
    assert np.array_equal(outs[0]["poses"], outs[1]["poses"])          # every rank holds the identical result
