"""GPU: short runs of the randomised parity sweeps (tests/tools/fuzz_*.py: the HIP path against the oracle on random
shapes, models, cycle parameters, kernel variants, receivers, sources).  Fixed seeds; the long runs are started by hand."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,args", [("fuzz_parity.py", ["60", "101"]), ("fuzz_kernels.py", ["40", "102"]),
                                         ("fuzz_receivers.py", ["60", "103"])])
def test_fuzz(script, args):
    env = {k: v for k, v in os.environ.items() if not k.startswith("EMG3D_") or k == "EMG3D_POOL_GB"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", script), *args], env=env, capture_output=True,
                       text=True, timeout=1200)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-15:])
    assert p.returncode == 0, tail
    assert "0 failures" in p.stdout, tail
