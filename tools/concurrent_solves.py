"""Aggregate throughput of several INDEPENDENT pose graphs solved concurrently on ONE GPU (one host thread and one HIP
stream per problem).  A single KITTI-scale graph is latency-bound (two dependent launches per CG iteration), so the
machine has room for more than one.  usage: python tools/concurrent_solves.py [n_threads ...]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader  # noqa: E402

pkg = pgo_loader.load()
ds = pgo_loader.datasets()
STEPS, WARMUP = 25, 5


def make(seed):
    g = ds.manhattan_se3(10000, 40000, seed=20260928)     # the same graph in every slot: equal work, so the ratio is the concurrency
    prob, poses = pkg.problem_from_graph(g)
    opt = pkg.SolverOptions(max_num_iterations=2 ** 30, linear_solver_type=pkg.BLOCK_JACOBI_PCG, eta=0.1,
                            max_linear_solver_iterations=500, function_tolerance=0.0, parameter_tolerance=0.0,
                            gradient_tolerance=0.0, pcg_cluster_poses=2)
    prob.solver_begin(opt)
    prob.solver_step(WARMUP)
    prob.solver_reset()
    return g, prob, poses


def main(counts):
    probs = [make(i) for i in range(max(counts))]
    for n in counts:
        for _, p, _ in probs[:n]:
            p.solver_reset()
        bar = threading.Barrier(n + 1)

        def work(p):
            bar.wait()
            p.solver_step(STEPS)
            bar.wait()

        ths = [threading.Thread(target=work, args=(probs[i][1],)) for i in range(n)]
        for t in ths:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in ths:
            t.join()
        edges = sum(probs[i][0].E for i in range(n))
        print("%d concurrent graphs: %.3f ms per LM iteration per graph, aggregate %.1f M edge-LM-iterations/s, %.0f LM it/s" % (
            n, 1e3 * dt / STEPS, edges * STEPS / dt / 1e6, n * STEPS / dt), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8])
