import os, sys, threading, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgo_loader
pkg = pgo_loader.load()
world = int(sys.argv[1]); iters = int(sys.argv[2]); seg = int(sys.argv[3])
group = pkg.loopback_create(world)
res = [None] * world
def run(rank):
    p = pkg.Problem()
    p.comm_init_loopback(group, rank)
    res[rank] = pkg.lib().pgo_debug_comm_stress(p._h, C.c_int(iters), C.c_int(seg))
ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
[t.start() for t in ts]; [t.join(120) for t in ts]
print("world", world, "iters", iters, "seg", seg, "mismatches per rank", res)
