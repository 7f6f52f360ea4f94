"""CPU: the C-ABI library loads and exports every symbol include/scpn_hip.h declares; no compute call is made."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "scpn_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|double|char)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_symbols_are_exported_and_bound():
    from scpn_fusion_core_amd import _lib

    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/scpn_hip.h but not exported by libscpn_hip.so"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    for legacy in ("create_solver", "set_boundary_dirichlet", "run_step", "run_step_converged", "destroy_solver", "delete_solver"):
        assert legacy in names  # hpc/solver.cpp:200-335


def test_library_reports_version_and_struct_sizes():
    from scpn_fusion_core_amd import _lib

    assert b"gfx950" in _lib.lib().scpn_version()
    # layout of the C parameter blocks (include/scpn_hip.h) as seen by ctypes
    assert ctypes.sizeof(_lib.PicardParams) % 8 == 0 and ctypes.sizeof(_lib.PicardResult) == 80


def test_no_gpu_means_loud_failure_not_fallback():
    from scpn_fusion_core_amd import _lib, multigrid_solve as mgs

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    z = np.zeros((9, 9))
    r = np.linspace(1.0, 2.0, 9)
    with pytest.raises(_lib.HipUnavailableError):
        mgs.mg_smooth(z.copy(), z, r, 0.1, 0.1, 1.0, 1)
    with pytest.raises(_lib.HipUnavailableError):
        mgs.multigrid_solve(z, z, 1.0, 2.0, -1.0, 1.0, 9, 9)
    with pytest.raises(ValueError):  # argument validation comes first, like the reference
        mgs.mg_smooth(z.copy(), z, r, 0.1, 0.1, 2.5, 1)
    assert _lib.lib().create_solver(9, 9, 1.0, 2.0, -1.0, 1.0) is None  # legacy ABI: NULL handle, never a CPU solver
    assert _lib.lib().create_solver(1, 9, 1.0, 2.0, -1.0, 1.0) is None
    assert _lib.device_info() is None and not _lib.gpu_available()


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "scpn_fusion_core_amd"
    for f in list(pkg.glob("*.py")) + [q for q in (pkg / "csrc").glob("*") if q.is_file()]:
        src = f.read_text(errors="ignore")
        assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_scpn_fusion_rs_stand_in_exports_the_gpu_tier_surface(monkeypatch):
    """integration/scpn_fusion_rs: the names benchmarks/bench_gpu_gs_solver.py:82-113 and the dispatcher's GPU tier import."""
    import importlib
    import sys
    from pathlib import Path

    monkeypatch.syspath_prepend(str(Path(__file__).resolve().parents[1] / "integration"))
    sys.modules.pop("scpn_fusion_rs", None)
    mod = importlib.import_module("scpn_fusion_rs")
    try:
        for name in ("PyGpuSolver", "py_gpu_available", "py_gpu_info", "multigrid_vcycle"):
            assert hasattr(mod, name), name
        assert isinstance(mod.py_gpu_available(), bool)
    finally:
        sys.modules.pop("scpn_fusion_rs", None)


def test_reference_bridge_loads_the_shipped_library():
    """Seam B1 through the REFERENCE's own ``HPCBridge`` (imported from the reference checkout in a child process, so nothing of it
    stays importable here): trust gate, six symbols, prototypes -- tools/check_reference_bridge.py.  Skipped where the reference
    checkout does not exist (the GPU box)."""
    import subprocess
    import sys

    import pytest

    from tests.conftest import ROOT

    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_reference_bridge.py")], capture_output=True, text=True, timeout=120)
    if r.returncode == 77:
        pytest.skip("no reference checkout")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "reference bridge check: passed" in r.stdout and "FAIL" not in r.stdout
