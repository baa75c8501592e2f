"""-m gpu, ONE GPU: the row-sharded path with one PROCESS per rank over the IPC transport (csrc/pgo_comm.cpp IpcComm, include/pgo.h
pgo_comm_init_ipc) — two (and three) real processes share this box's GPU, map each other's exchange buffers and flag arrays
through hipIpc memory handles, and the kernels of the owner-only CG exchange their segments THEMSELVES (DeviceGraph::peer_tab:
stores into every rank's buffer, release-store of the launch's sequence number into every rank's flag array, acquire-wait on
the own one; no host-enqueued collective per CG iteration).  What a one-GPU box can exercise of SURVEY.md section 8e beyond the
in-process virtual ranks of tests/test_gpu_sharded.py: separate address spaces, separate hardware queues, handles, the
cross-process control plane.  (RCCL refuses two ranks on one device; tests/test_gpu_rccl2.py is the two-GPU test.)

Every rank must reproduce, bit for bit, what the same number of in-process virtual ranks compute with the same device-initiated
exchange, and the single-rank solve to the tolerances of test_gpu_sharded.py."""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["PGO_ROOT"])
import numpy as np
import pgo_loader
pkg = pgo_loader.load(); ds = pgo_loader.datasets()
rank, world, name, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["PGO_IPC_NAME"], os.environ["PGO_OUT"]
pkg.set_device(0)                       # every rank on the box's one GPU
g = ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0)
prob, poses = pkg.problem_from_graph(g)
prob.comm_init_ipc(name, rank, world)
s = pkg.solve(pkg.SolverOptions(max_num_iterations=8, linear_solver_type=pkg.BLOCK_JACOBI_PCG, pcg_cluster_poses=2), prob)
np.savez(out + ".%d.npz" % rank, cost=s.iterations["cost"], ok=s.iterations["step_is_successful"],
         cg=s.iterations["linear_solver_iterations"], poses=poses, cg_form=s.cg_form, cg_exchange=s.cg_exchange, term=s.termination_type)
'''


def _virtual_ranks(gpu, g, world, opt_kw):
    group = gpu.loopback_create(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            prob, poses = gpu.problem_from_graph(g)
            prob.comm_init_loopback(group, rank)
            out[rank] = (gpu.solve(gpu.SolverOptions(**opt_kw), prob), poses)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    gpu.loopback_destroy(group)
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_processes_on_one_gpu_exchange_by_themselves(gpu, ds, tmp_path, world, monkeypatch):
    monkeypatch.setenv("PGO_BLOCK", "256")          # (work-groups of 256 slots: every pose pair fits one — what the owner-only CG needs)
    g = ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0)
    opt = dict(max_num_iterations=8, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    prob, poses1 = gpu.problem_from_graph(g)
    one = gpu.solve(gpu.SolverOptions(**opt), prob)
    monkeypatch.setenv("PGO_PEER_DIRECT", "1")      # the virtual ranks' reference run exchanges by the kernels too
    virt = _virtual_ranks(gpu, g, world, opt)
    monkeypatch.delenv("PGO_PEER_DIRECT")           # the IPC transport needs no switch: it is its normal mode
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    name = "/pgo_ipc_test_%d_%d" % (os.getpid(), world)
    env = dict(os.environ, PGO_ROOT=ROOT, WORLD_SIZE=str(world), PGO_IPC_NAME=name, PGO_OUT=str(tmp_path / "out"),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PGO_BLOCK="256")
    env.pop("PGO_PEER_DIRECT", None)
    if world == 2:
        # what a crashed run leaves behind (r06, csrc/pgo_comm.cpp IpcComm::init): a block of the same name with a VALID magic, the
        # right world size and half-used barrier words.  Rank 1 is started first and finds it; rank 0 replaces it; rank 1 has to
        # notice that the name points elsewhere now and attach to the live group.
        import struct
        with open("/dev/shm" + name, "wb") as f:
            f.write(struct.pack("<iiqiiQ", 0x50474f35, 1, 7, 0, 0, 12345) + b"\x01" * (16 * 16) + struct.pack("<i", world) + b"\0" * 8192)   # magic, arrived, generation, aborted, pad, nonce, chal / echo (stale answers), world
    order = list(range(world))[::-1] if world == 2 else list(range(world))
    procs = [None] * world
    for r in order:
        procs[r] = subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)))
        if world == 2 and r == 1:
            time.sleep(1.0)
    t0 = time.time()
    try:
        for p in procs:
            p.wait(timeout=max(1, 300 - (time.time() - t0)))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert [p.returncode for p in procs] == [0] * world
    res = [np.load(str(tmp_path / "out") + ".%d.npz" % r) for r in range(world)]
    for r, k in enumerate(res):
        assert int(k["cg_form"]) == 2 and int(k["cg_exchange"]) == 2     # the owner-only pipelined CG ran, exchanging by its kernels
        vs, vp = virt[r]
        assert np.array_equal(k["cost"], vs.iterations["cost"])           # bit for bit what in-process virtual ranks compute
        assert np.array_equal(k["cg"], vs.iterations["linear_solver_iterations"])
        assert np.array_equal(k["poses"], vp)
        assert list(k["ok"]) == list(one.iterations["step_is_successful"])        # ... and the single-rank solve
        assert list(k["cg"]) == list(one.iterations["linear_solver_iterations"])
        assert np.allclose(k["cost"], one.iterations["cost"], rtol=1e-8)
        assert np.abs(k["poses"] - poses1).max() < 1e-6
    assert all(np.array_equal(res[0]["poses"], k["poses"]) for k in res)
    assert max(one.iterations["linear_solver_iterations"]) > 20


def test_processes_with_the_staged_collective_exchange_boundary_rows(gpu, ds, tmp_path, monkeypatch):
    """The IPC transport with its device-initiated exchange switched off (PGO_PEER_DIRECT=0): the per-iteration collective is the transport's
    staged all-gather, and what it carries is the ranks' boundary rows (Summary::cg_exchange 3) — two processes on the box's one GPU
    reproduce the in-process virtual ranks with the host-enqueued exchange bit for bit."""
    world = 2
    monkeypatch.setenv("PGO_BLOCK", "256")
    g = ds.manhattan_se3(3001, 14000, seed=77, loop_radius=3.0)
    opt = dict(max_num_iterations=8, linear_solver_type=gpu.BLOCK_JACOBI_PCG, pcg_cluster_poses=2)
    monkeypatch.delenv("PGO_PEER_DIRECT", raising=False)
    virt = _virtual_ranks(gpu, g, world, opt)
    assert virt[0][0].cg_exchange == 3
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    name = "/pgo_ipc_test_staged_%d" % os.getpid()
    env = dict(os.environ, PGO_ROOT=ROOT, WORLD_SIZE=str(world), PGO_IPC_NAME=name, PGO_OUT=str(tmp_path / "out"),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PGO_BLOCK="256", PGO_PEER_DIRECT="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    t0 = time.time()
    try:
        for p in procs:
            p.wait(timeout=max(1, 300 - (time.time() - t0)))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert [p.returncode for p in procs] == [0] * world
    for r in range(world):
        k = np.load(str(tmp_path / "out") + ".%d.npz" % r)
        assert int(k["cg_form"]) == 2 and int(k["cg_exchange"]) == 3
        vs, vp = virt[r]
        assert np.array_equal(k["cost"], vs.iterations["cost"]) and np.array_equal(k["poses"], vp)
        assert np.array_equal(k["cg"], vs.iterations["linear_solver_iterations"])
