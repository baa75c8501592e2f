"""CPU: the lean per-incidence algebra of the linearisation (csrc/pgo_lin_lean.h: G = 2 (Rt + (|q|^2 - 1) I) [d]x, symmetric
products, END incidence = transposed BEGIN incidence) against the general closed-form blocks written with the 3x3 helpers of
csrc/pgo_math.h, on the host: tools/lean_check_cli — random poses (unit quaternions and quaternions 1e-6 off the unit sphere),
measurements, identity / block-diagonal / diagonal information, Jacobi scales, constant blocks, every loss kind, both sides."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")


def test_lean_incidence_equals_the_general_blocks():
    subprocess.check_call(["make", "-s", "-C", TOOLS, "lean_check_cli"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(TOOLS, "lean_check_cli"), "30000", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    worst = [float(x) for x in re.findall(r"(\d\.\d+e[-+]\d+)", r.stdout)]
    assert len(worst) == 3 and max(worst[:2]) < 1e-13 and worst[2] < 1e-12, r.stdout      # blocks to 1e-13, gradient (cancellation) 1e-12
