"""The oracle is test infrastructure: nothing in the product package (or tools/) may import it, and the
product must not fall back to it.  bench.py may only touch it inside cpu_baseline(); __graft_entry__ only
inside smoke()."""
import ast
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _imports_oracle(path):
    tree = ast.parse(open(path).read())
    hits = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            hits += [n.name for n in node.names if n.name.split(".")[0] == "oracle"]
        elif isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
            hits.append(node.module)
    return hits


def test_product_package_never_imports_the_oracle():
    for root in ("facodec_amd", "tools"):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith(".py"):
                    assert _imports_oracle(os.path.join(dp, f)) == [], f"{dp}/{f} imports the oracle"


def _function_level_only(path, allowed):
    tree = ast.parse(open(path).read())
    for node in tree.body:                      # no module-level oracle import
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            mod = getattr(node, "module", None) or ",".join(n.name for n in node.names)
            assert not mod.startswith("oracle"), f"{path}: module-level oracle import"
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = any(isinstance(n, ast.ImportFrom) and n.module and n.module.startswith("oracle") for n in ast.walk(node))
            if uses:
                assert node.name in allowed, f"{path}: {node.name} imports the oracle"


def test_bench_and_entry_use_the_oracle_only_as_checker():
    _function_level_only(os.path.join(REPO, "bench.py"), {"cpu_baseline"})
    _function_level_only(os.path.join(REPO, "__graft_entry__.py"), {"smoke"})
