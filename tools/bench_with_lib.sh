#!/bin/bash
# usage (GPU box, repo root): tools/bench_with_lib.sh <library.so> [bench.py flags]  -- bench.py on an experiment build of the library
# (tools/labenv.py: KBNER_LIB; the product binding itself reads no environment variable)
lib=$1; shift
KBNER_LIB=$lib python - "$@" <<'PY'
import os, sys, runpy
root = os.getcwd()
sys.path.insert(0, os.path.join(root, "tools")); sys.path.insert(0, os.path.join(root, "kb-ner_amd"))
import labenv; labenv.apply()
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
PY
