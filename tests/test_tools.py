"""The scripts under tools/ and the root scripts at least parse, and what they import from this repository exists (a pruned helper
module once left tools/traffic_model.py importing a file that was gone)."""
import ast
import glob
import os

from conftest import ROOT


def _local_imports(path):
    tree = ast.parse(open(path).read(), path)
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
            yield node.module
        elif isinstance(node, ast.Import):
            for a in node.names:
                yield a.name


def test_scripts_parse_and_their_local_imports_exist():
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(scripts) >= 10
    for path in scripts:
        for mod in _local_imports(path):
            top = mod.split(".")[0]
            if top not in ("tools", "kinematic_icp_amd", "oracle", "tests"):
                continue
            rel = os.path.join(ROOT, *mod.split("."))
            assert os.path.exists(rel + ".py") or os.path.isdir(rel), "%s imports %s, which is not in the tree" % (os.path.relpath(path, ROOT), mod)
