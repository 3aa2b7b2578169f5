"""Regenerates tools/README.md from the first paragraph of every tool's docstring / leading comment:  python tools/make_readme.py"""
import ast
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def describe(path):
    src = open(path).read()
    if path.endswith(".py"):
        try:
            doc = ast.get_docstring(ast.parse(src)) or ""
        except SyntaxError:
            doc = ""
    else:
        lines = [l[1:].strip() for l in src.splitlines()[1:] if l.startswith("#")]
        doc = "\n".join(lines)
    first = re.split(r"\n\s*\n", doc.strip())[0] if doc.strip() else ""
    return " ".join(first.split())


rows = []
for name in sorted(os.listdir(HERE)):
    p = os.path.join(HERE, name)
    if os.path.isfile(p) and name.endswith((".py", ".sh")) and name != "make_readme.py":
        rows.append("| `%s` | %s |" % (name, describe(p).replace("|", "\\|")))
ub = sorted(f for f in os.listdir(os.path.join(HERE, "ubench")) if f.endswith(".hip"))
out = """# tools/ -- probes, sweeps and profiling drivers

Everything here runs on the GPU box (`gpurun -- python tools/<name>.py`); nothing is imported by the product or the tests.  Results worth
keeping are copied to `profiles/`.  Scripts whose question is closed, and rejected kernel experiments as patches, live under
`docs/archive/` (each named in `docs/archive/DESIGN_rounds_1-3.md` or `DESIGN.md` with its numbers).  This table: `python tools/make_readme.py`.

| tool | what it measures |
|---|---|
%s

`ubench/`: %s (`valu_rate.hip`: the issue-cost table of `profiles/r03_ubench_valu_issue_rate.txt`; `semantics.hip`: `v_fract` / `v_cvt_flr`
against the oracle's text; `exec_half.hip`: instruction cost against the number of active lanes, `profiles/r05_ubench_exec_lanes.txt`; `lds_gather.hip`:
the staged march's window fill on its own and against its steps, `profiles/r05_c4_fill.txt`).
""" % ("\n".join(rows), ", ".join("`%s`" % f for f in ub))
open(os.path.join(HERE, "README.md"), "w").write(out)
print("tools/README.md: %d tools" % len(rows))
