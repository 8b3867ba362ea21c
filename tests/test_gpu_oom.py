"""GPU: a device allocation that cannot be satisfied must surface as ``HipLibraryError`` -- not as a memory fault of the
process -- and leave the process able to solve afterwards (parity bookkeeping).

The scenario runs in a CHILD process: most of the HBM is taken by a ballast tensor, then (a) a handle whose level-0 arrays do
not fit is created, (b) a handle that fits is asked for a cycle whose hierarchy, working copies and factor caches do not;
both must raise.  The ballast goes, and a 128^3 solve IN THE SAME PROCESS converges.  (Were a kernel launched on a missing
array, the child would die of "Memory access fault by GPU node" and the test would say so instead of taking pytest with it.)
The library retries an allocation once after releasing its own block pool (``MG::raw_alloc``); after the failure the handle
is `broken`: it launches nothing and every entry point answers hipErrorOutOfMemory until it is destroyed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
import bench
import emg3d_amd as em
from emg3d_amd import _lib
from emg3d_amd.solver import DeviceMG, MGParameters

out = {}
lib = _lib.load()
grid, model, sfield, cycle = bench.build_problem(em, "256V", 1.0)
vm = em.VolumeModel(grid, model, sfield)
var = MGParameters(verb=0, cycle=cycle, sslsolver=False, linerelaxation=True, semicoarsening=True, vnC=grid.vnC, ordering='colour')

def ballast(leave_gb):
    lib.emg3d_hip_release_cached()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    return torch.empty(int(free - leave_gb * 2 ** 30), dtype=torch.uint8, device="cuda")

# (a) level 0 itself does not fit: 3.4 GB of field and model arrays into 1 GB
b = ballast(1.0)
try:
    DeviceMG(grid, vm, np.complex128)
    out["create"] = "no error"
except _lib.HipLibraryError as exc:
    out["create"] = "HipLibraryError"
del b

# (b) the handle fits (3.4 GB into 6), its hierarchy + working copies + factor caches (32 GB) do not
b = ballast(6.0)
dev = DeviceMG(grid, vm, np.complex128)
dev.set_params(var); dev.set_sfield(sfield); dev.set_efield(None)
try:
    dev.cycles(1, [1], [4])
    out["cycle"] = "no error"
except _lib.HipLibraryError as exc:
    out["cycle"] = "HipLibraryError"
try:                                    # a broken handle keeps refusing, it does not launch on missing arrays
    dev.cycles(1, [1], [4])
    out["cycle_again"] = "no error"
except _lib.HipLibraryError as exc:
    out["cycle_again"] = "HipLibraryError"
dev.close()
# ... and the whole solve() surface on the same starved device
try:
    em.solve(grid, model, sfield, cycle=cycle, semicoarsening=True, linerelaxation=True, maxit=1, verb=0)
    out["solve"] = "no error"
except _lib.HipLibraryError as exc:
    out["solve"] = "HipLibraryError"
del b
lib.emg3d_hip_release_cached()
torch.cuda.empty_cache()

# the process is fine: BASELINE configs[1] converges
g1, m1, s1, c1 = bench.build_problem(em, "128F", 1.0)
e, info = em.solve(g1, m1, s1, cycle=c1, semicoarsening=True, linerelaxation=True, tol=1e-6, verb=0, return_info=True)
out["after"] = {"exit": int(info["exit"]), "it_mg": int(info["it_mg"]), "rel_error": float(info["rel_error"]),
                "finite": bool(np.all(np.isfinite(np.asarray(e))))}
print("OOM_RESULT " + json.dumps(out))
"""


def test_out_of_memory_surfaces_as_error_and_the_process_survives():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("OOM_RESULT ")]
    assert p.returncode == 0 and line, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    out = json.loads(line[-1][len("OOM_RESULT "):])
    assert out["create"] == "HipLibraryError", out
    assert out["cycle"] == "HipLibraryError" and out["cycle_again"] == "HipLibraryError", out
    assert out["solve"] == "HipLibraryError", out
    assert out["after"]["exit"] == 0 and out["after"]["finite"] and out["after"]["it_mg"] <= 12 and out["after"]["rel_error"] < 1e-6, out
