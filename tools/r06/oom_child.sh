#!/bin/bash
# runs the child scenario of tests/test_gpu_oom.py in the open (full stderr)
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_oom as t
src = t.CHILD % {"root": os.getcwd()}
exec(compile(src, "oom_child", "exec"))
PY
