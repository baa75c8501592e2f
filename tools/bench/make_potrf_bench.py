"""Regenerates tools/bench/potrf_bench.hip's copy of diag_block_wave (and its helpers) from csrc/pgo_front_kernels.hip."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "posegraph-ceres_amd", "csrc", "pgo_front_kernels.hip")).read()
bench_path = os.path.join(ROOT, "tools", "bench", "potrf_bench.hip")
bench = open(bench_path).read()
a = src.index("constexpr int LDW = FRONT_NB + 2;")
b = src.index("// One 48-column panel step (see FrontJob).")
c = bench.index("constexpr int LDW = FRONT_NB + 2;")
d = bench.index("__global__ __launch_bounds__(64) void kt(")
open(bench_path, "w").write(bench[:c] + src[a:b] + bench[d:])
print("potrf_bench.hip refreshed (%d bytes of kernel text)" % (b - a))
