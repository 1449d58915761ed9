#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Record the API surface the reference's entry script exercises, so that "train.py
drops in unchanged" (north_star) is a tested property of the mirror instead of a claim.

Reads /root/reference/train.py with `ast` (nothing is executed, nothing is copied) and writes tests/golden/train_surface.json:
  imports     every `from flair... import name` / `import flair.x [as y]` with its line
  attributes  for each variable of a known role (student, trainer, config, corpus, embedding, the flair / datasets / Embeddings
              modules ...) the attribute names train.py loads or stores on it, with lines
  calls       for each call through one of those names (and ColumnDataLoader / ConfigParser / trainer_func ...) the keyword
              names and the number of positional arguments, with lines
The fixture is data about the boundary (names, keywords, line numbers), not source text.

    python oracle/gen_train_surface.py
"""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_TRAIN = "/root/reference/train.py"
OUT = os.path.join(ROOT, "tests", "golden", "train_surface.json")

# variable name in train.py -> role.  `config` is a Params until train.py:90 rebinds it to the ConfigParser.
ROLES = {"student": "tagger", "trainer": "trainer", "corpus": "corpus", "embedding": "embedding", "flair": "flair_module",
         "datasets": "datasets_module", "Embeddings": "embeddings_module", "test_loader": "loader", "loader": "loader",
         "train_eval_result": "result", "subcorpus": "dataset"}
CALL_NAMES = {"ColumnDataLoader", "ConfigParser", "trainer_func", "teacher_func", "ListCorpus", "count_parameters", "Path"}


def role_of(node, config_rebound_at):
    if isinstance(node, ast.Name):
        if node.id == "config":
            return "config_parser" if node.lineno > config_rebound_at or (node.lineno == config_rebound_at) else "params"
        return ROLES.get(node.id)
    # config.config[...] -> the parsed YAML dict; trainer.corpus -> corpus
    if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name):
        if node.value.id == "trainer" and node.attr == "corpus":
            return "corpus"
        if node.value.id == "student" and node.attr == "embeddings":
            return "stacked_embeddings"
        if node.value.id == "student" and node.attr == "tag_dictionary":
            return "dictionary"
    return None


def main():
    src = open(REF_TRAIN, encoding="utf-8").read()
    tree = ast.parse(src)
    rebound = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and isinstance(node.value.func, ast.Name) \
                and node.value.func.id == "ConfigParser":
            rebound = node.lineno
    assert rebound is not None
    imports, attrs, calls = [], {}, []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "flair":
            for a in node.names:
                imports.append({"module": node.module, "name": a.name, "line": node.lineno})
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split(".")[0] == "flair":
                    imports.append({"module": a.name, "name": None, "asname": a.asname, "line": node.lineno})
        elif isinstance(node, ast.Attribute):
            r = role_of(node.value, rebound)
            if r is not None:
                # the rebinding statement itself (`config = ConfigParser(config, ...)`) reads the OLD (Params) binding
                ctx = "store" if isinstance(node.ctx, ast.Store) else "load"
                d = attrs.setdefault(r, {}).setdefault(node.attr, {"load": [], "store": []})
                d[ctx].append(node.lineno)
        elif isinstance(node, ast.Subscript):
            r = role_of(node.value, rebound)
            if r in ("params", "config_parser") and isinstance(node.ctx, ast.Load):
                attrs.setdefault(r, {}).setdefault("__getitem__", {"load": [], "store": []})["load"].append(node.lineno)
        elif isinstance(node, ast.Compare):
            for op, comp in zip(node.ops, node.comparators):
                r = role_of(comp, rebound)
                if isinstance(op, (ast.In, ast.NotIn)) and r is not None:
                    attrs.setdefault(r, {}).setdefault("__contains__", {"load": [], "store": []})["load"].append(node.lineno)
        if isinstance(node, ast.Call):
            f = node.func
            name = owner = None
            if isinstance(f, ast.Name) and f.id in CALL_NAMES:
                name, owner = f.id, None
            elif isinstance(f, ast.Attribute):
                r = role_of(f.value, rebound)
                if r is not None:
                    name, owner = f.attr, r
            elif isinstance(f, ast.Call) and isinstance(f.func, ast.Name) and f.func.id == "getattr" \
                    and isinstance(f.args[0], ast.Name) and f.args[0].id in ROLES:
                # getattr(trainer, 'train')(**train_config)
                name, owner = f.args[1].value, ROLES[f.args[0].id]
            if name is not None:
                calls.append({"owner": owner, "func": name, "line": node.lineno, "n_positional": len(node.args),
                              "keywords": sorted(k.arg for k in node.keywords if k.arg is not None),
                              "star_kwargs": sorted(ast.unparse(k.value) for k in node.keywords if k.arg is None)})
    for r in attrs.values():
        for d in r.values():
            d["load"].sort()
            d["store"].sort()
    # keyword sets the shipped YAMLs pass to the constructors / train() (config/*.yaml -> ConfigParser / train.py:127-131,412)
    import glob
    import yaml
    ykeys = {}
    for path in sorted(glob.glob("/root/reference/config/*.yaml")):
        c = yaml.load(open(path), Loader=yaml.FullLoader)
        tname = c.get("trainer", "ModelFinetuner")
        sect = {"train": c.get("train", {}), "trainer:" + tname: c.get(tname, {})}
        for k, v in c.get("model", {}).items():
            sect["model:" + k] = v
        for k, v in c.get("embeddings", {}).items():
            sect["embeddings:" + k.split("-")[0]] = v
        for k, v in sect.items():
            ykeys.setdefault(k, set()).update((v or {}).keys())
    out = {"source": "ast of /root/reference/train.py (%d lines)" % (src.count("\n") + 1), "config_rebound_at_line": rebound,
           "imports": sorted(imports, key=lambda x: (x["line"], x["name"] or "")),
           "attributes": {k: dict(sorted(v.items())) for k, v in sorted(attrs.items())},
           "calls": sorted(calls, key=lambda x: (x["line"], x["func"])),
           "yaml_keywords": {k: sorted(v) for k, v in sorted(ykeys.items())}}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=False)
        f.write("\n")
    print("wrote", OUT, "imports", len(imports), "roles", len(attrs), "calls", len(calls))


if __name__ == "__main__":
    main()
