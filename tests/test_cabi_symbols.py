"""CPU-only: the C-ABI library builds, loads and exports every symbol that
include/emg3d_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from emg3d_amd import _lib
    return _lib


def test_header_and_binding_agree(lib):
    header = open(os.path.join(ROOT, "include", "emg3d_hip.h")).read()
    declared = set(re.findall(r"\b(emg3d_[A-Za-z0-9_]+)\s*\(", header))
    declared.discard("emg3d_mg_t")
    bound = set(lib.SIGNATURES)
    assert declared == bound, (declared - bound, bound - declared)


def test_every_symbol_exported(lib):
    handle = lib.load()
    for name in lib.SIGNATURES:
        assert hasattr(handle, name)
    header = open(os.path.join(ROOT, "include", "emg3d_hip.h")).read()
    want = int(re.search(r"#define\s+EMG3D_HIP_ABI_VERSION\s+(\d+)", header).group(1))
    assert handle.emg3d_hip_version() == want == lib.ABI_VERSION


def test_stale_library_is_refused(lib, monkeypatch):
    """A library built from another version of the header (a stale .so) is refused at load with a clear message."""
    monkeypatch.setattr(lib, "_loaded", {})
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION + 1)
    with pytest.raises(lib.HipLibraryError, match="ABI version"):
        lib._open(lib.LIB_PATH)


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "_path", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libemg3d_hip.so")
    with pytest.raises(lib.HipLibraryError):
        lib.load()


def test_product_library_has_no_tuning_knobs(lib):
    """The product library reads six documented environment variables and nothing else; the lab build (loaded only by
    tests/test_gpu_variants.py and tools/) has one per tuning knob and exports the same C ABI."""
    import subprocess
    def names(path):
        out = subprocess.run(["strings", path], capture_output=True, text=True).stdout.splitlines()
        return sorted({w for w in out if re.fullmatch(r"EMG3D_[A-Z0-9_]+", w)})
    prod = names(lib.LIB_PATH)
    assert prod == ["EMG3D_BATCH_TUNE", "EMG3D_GRAPH", "EMG3D_LOG", "EMG3D_LOG_SETUP", "EMG3D_PLACE_TRIES", "EMG3D_POOL_GB"], prod
    assert os.path.exists(lib.LAB_PATH)
    labn = names(lib.LAB_PATH)
    assert set(prod) < set(labn) and "EMG3D_THM_LIFO" in labn and "EMG3D_THA" in labn and len(labn) > 20
    prev = lib.use(lib.LAB_PATH)
    try:
        handle = lib.load()
        for name in lib.SIGNATURES:
            assert hasattr(handle, name)
    finally:
        lib.use(prev)


def test_restrict_weights_host_only(lib):
    """restrict_weights is O(n) host work inside the library: callable on CPU.
    Hand-computed numbers of reference tests/test_core.py:422-441."""
    import numpy as np
    from emg3d_amd import core
    edges = np.array([0., 500, 1200, 2000, 3000])
    width = edges[1:] - edges[:-1]
    centr = edges[:-1] + width / 2
    c_edges = edges[::2]
    c_width = c_edges[1:] - c_edges[:-1]
    c_centr = c_edges[:-1] + c_width / 2
    wl, w0, wr = core.restrict_weights(edges, centr, width, c_edges, c_centr, c_width)
    np.testing.assert_allclose(wl, [350 / 250, 250 / 600, 400 / 900])
    np.testing.assert_allclose(w0, [1., 1., 1.])
    np.testing.assert_allclose(wr, [350 / 600, 500 / 900, 400 / 500])


def test_hot_kernels_use_no_scratch(tmp_path):
    """None of the smoother / residual / transfer kernels may touch scratch memory: a runtime index into
    a local array or a struct that is only passed through silently turns registers into stack slots (a
    global-memory round trip inside the kernel; it cost 4 % of the cycle once).  Checked on the gfx950
    code objects with the compiler's resource remarks (needs hipcc, not a GPU)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import __graft_entry__ as g
    from concurrent.futures import ThreadPoolExecutor

    def remarks(unit):      # the lab build: a superset of the product's kernels; every translation unit of the library
        name, src, extra = unit
        out = subprocess.run([hipcc] + g.FLAGS + extra + ["-c", "-DEMG3D_LAB", "-Rpass-analysis=kernel-resource-usage",
                              "-o", str(tmp_path / (name + ".o")), os.path.join(ROOT, "emg3d_amd", "csrc", src)],
                             capture_output=True, text=True, cwd=str(tmp_path))
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stderr
    with ThreadPoolExecutor(max_workers=8) as pool:
        text = "\n".join(pool.map(remarks, g.UNITS))
    name, seen, bad = None, 0, []
    for line in text.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize [bytes/lane]:" in line and name:
            size = int(line.split("ScratchSize [bytes/lane]:")[1].split()[0])
            # every line-sweep kernel (k_line_sweep_qc, _thm, _qpl, _rp, the thread-per-line k_line_sweep and the lab
            # variants _q, _qm, _th, _tw), residual, transfer and conversion kernels
            if any(k in name for k in ("k_line_sweep", "k_residual", "k_restrict", "k_prolong", "k_point_sweep",
                                       "k_transpose", "k_split0")):
                seen += 1
                if size:
                    bad.append((name, size))
    assert seen >= 60, seen
    assert not bad, bad
