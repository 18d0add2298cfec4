#!/usr/bin/env python3
"""Extract the plug-in CONTRACT of the reference's entry script (main.py) by parsing it -- never importing or running it
(it needs torchaudio, which is not installable here) -- and store it as data: which names it imports from which module,
with which keyword arguments it calls them, which Config attributes it reads and which keys of the returned dicts it
indexes.  tests/test_main_contract_cpu.py then holds this package to it.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_main_contract.py [--reference /root/reference]
"""
import argparse
import ast
import json
from pathlib import Path

HERE = Path(__file__).resolve().parent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    tree = ast.parse((Path(args.reference) / "main.py").read_text())
    imports, calls, config_attrs, dict_keys = {}, {}, set(), {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module in ("config", "utils", "dataset", "trainer"):
            imports.setdefault(node.module, []).extend(a.name for a in node.names)
    imported = {n for names in imports.values() for n in names}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in imported | {"DataLoader"}:
            entry = calls.setdefault(node.func.id, {"keywords": [], "n_positional": 0})
            entry["keywords"] = sorted(set(entry["keywords"]) | {k.arg for k in node.keywords if k.arg})
            entry["n_positional"] = max(entry["n_positional"], len(node.args))
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "config":
            config_attrs.add(node.attr)
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id in ("history", "test_results"):
            key = node.slice
            if isinstance(key, ast.Constant) and isinstance(key.value, str):
                dict_keys.setdefault(node.value.id, set()).add(key.value)
            elif isinstance(key, ast.JoinedStr):                 # f'class_{config.LOSS_TYPE}'
                text = "".join(v.value if isinstance(v, ast.Constant) else "{}" for v in key.values)
                dict_keys.setdefault(node.value.id, set()).add(text)
    # tuple-unpacking arity of the calls whose results are destructured
    unpack = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and isinstance(node.value.func, ast.Name) \
                and isinstance(node.targets[0], ast.Tuple):
            unpack[node.value.func.id] = len(node.targets[0].elts)
    out = {"imports": {k: sorted(set(v)) for k, v in imports.items()}, "calls": calls,
           "config_attributes": sorted(config_attrs), "dict_keys": {k: sorted(v) for k, v in dict_keys.items()},
           "unpacked_results": unpack}
    (HERE / "main_contract.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
