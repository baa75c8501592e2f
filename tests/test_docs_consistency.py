"""CPU: documents that quote measured tables must quote the committed artefact.  DESIGN.md section 7 carries the latest
profiles/rNN_config_table.md verbatim between two markers (r02 verdict item 7: DESIGN said 320 iterations for the C2 exact row
where the profile said 193); README.md quotes the committed bench line's ms_per_step."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_config_table_is_the_committed_profile_table():
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_config_table.md")))[-1]
    table = open(latest).read().strip("\n")
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- config-table:begin[^>]*-->\n(.*?)\n<!-- config-table:end -->", design, re.S)
    assert m, "DESIGN.md has no config-table block"
    assert os.path.basename(latest) in design
    assert m.group(1).strip("\n") == table
    # the iteration counts quoted in prose for the C2 exact row are the table's
    row = [ln for ln in table.split("\n") if ln.startswith("| C2 Manhattan, exact request")][0]
    its = int(row.split("|")[4])
    assert ("%d with the r03 sources" % its) in design or ("%d iterations" % its) in design


def test_readme_quotes_the_committed_bench_line():
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))[-1]
    d = json.load(open(latest))
    readme = open(os.path.join(ROOT, "README.md")).read()
    assert ("%.3f" % d["ms_per_step"]) in readme, "README.md does not quote ms_per_step %.3f of %s" % (d["ms_per_step"], os.path.basename(latest))
