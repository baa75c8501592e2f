"""CPU: the host side of the symmetric tile form (csrc/pgo_sym_host.cpp: row partition, slot order, positions of the row ranges)
checked without a GPU by tools/sym_check_cli, which builds the layout exactly as the library does and
runs a scalar emulation of the product kernel that consumes it against a plain sum over all incidences.  The same check run on
layouts with ONE damaged entry must fail in every case — the checker checks."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", TOOLS, "sym_check_cli"], stdout=subprocess.DEVNULL)
    return os.path.join(TOOLS, "sym_check_cli")


def _run(cli, *args):
    r = subprocess.run([cli, *map(str, args)], capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout + r.stderr[-2000:]


def test_layouts_of_random_graphs_reproduce_the_plain_product(cli):
    code, out = _run(cli, 400, 1000)
    assert code == 0, out
    words = out.split()
    assert int(words[words.index("tiles,") - 1]) > 10000 and int(words[words.index("interior") - 1]) > 100000, out   # not vacuous


def test_the_form_of_one_ranks_rows(cli):
    """r06: every rank of a row-sharded solve keeps the form of ITS rows (csrc/pgo_sym_host.h row_lo / row_hi), tiles made of whole
    2-pose clusters in consecutive lanes: the owned rows' product, one stored block per owned end of an edge (one in all when interior),
    nobody else's row in a tile, no split unit."""
    code, out = _run(cli, 400, 9000, 0, 1)
    assert code == 0, out
    words = out.split()
    assert int(words[words.index("tiles,") - 1]) > 2000 and int(words[words.index("interior") - 1]) > 20000, out


@pytest.mark.parametrize("damage", [1, 2, 3, 4])
def test_one_damaged_entry_is_noticed(cli, damage):
    code, out = _run(cli, 150, 5000, damage)
    assert code == 0 and "150 bad" in out, out
