#!/usr/bin/env python3
"""Load the shipped libscpn_hip.so through the REFERENCE's own loader (build container only).

The drop-in claim of seam B1: ``scpn_fusion.hpc.hpc_bridge.HPCBridge`` -- the reference class, imported from
/root/reference, not this package's mirror -- pointed at this repo's library by ``SCPN_SOLVER_LIB`` with its SHA-256 in
``SCPN_SOLVER_LIB_SHA256`` must pass the trust gate (hpc/_hpc_native_trust.py:186-221), bind all six legacy symbols with the
prototypes of hpc/hpc_bridge.py:189-250 and report ``is_available()``.  No device is needed: without one ``create_solver`` returns
NULL, as hpc/solver.cpp:205-222 specifies for a failed construction, and the bridge stays uninitialised.

Exit code 0 = all checks hold; 77 = skipped (no reference checkout: the GPU box); 1 = a check failed.  Run directly or through
tests/test_abi_symbols.py::test_reference_bridge_loads_the_shipped_library.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REF = Path(os.environ.get("SCPN_REFERENCE_ROOT", "/root/reference"))
LEGACY = ("create_solver", "run_step", "run_step_converged", "set_boundary_dirichlet", "destroy_solver", "delete_solver")


def main() -> int:
    if not (REF / "src" / "scpn_fusion" / "hpc" / "hpc_bridge.py").exists():
        print("skipped: no reference checkout at", REF)
        return 77
    lib = ROOT / "scpn_fusion_core_amd" / "libscpn_hip.so"
    if not lib.exists():
        print("FAIL: build the library first (python -c 'import __graft_entry__ as g; g.build()')")
        return 1
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF / "src"))
    os.environ["SCPN_SOLVER_LIB"] = str(lib)
    os.environ["SCPN_SOLVER_LIB_SHA256"] = hashlib.sha256(lib.read_bytes()).hexdigest()
    os.environ.pop("SCPN_SOLVER_TRUST_MANIFEST", None)
    from scpn_fusion.hpc.hpc_bridge import HPCBridge  # the reference's class

    failures = []

    def expect(cond, what):
        print(("ok   " if cond else "FAIL ") + what)
        if not cond:
            failures.append(what)

    bridge = HPCBridge()
    expect(bridge.load_error is None, f"no load error (got {bridge.load_error!r})")
    expect(bridge.is_available(), "HPCBridge.is_available() with the shipped library behind the SHA-256 trust gate")
    expect(bridge.lib_sha256 == os.environ["SCPN_SOLVER_LIB_SHA256"], "the bridge verified the digest it was given")
    expect(Path(bridge.lib_path) == lib, f"the bridge loaded {lib.name}")
    L = bridge.lib
    for name in LEGACY:
        expect(L is not None and hasattr(L, name), f"symbol {name}")
    if L is not None and bridge.is_available():
        expect(L.create_solver.restype is ctypes.c_void_p and len(L.create_solver.argtypes) == 6, "create_solver(int, int, double x 4) -> void*")
        expect(len(L.run_step.argtypes) == 5, "run_step(void*, double*, double*, int, int)")
        expect(L.run_step_converged.restype is ctypes.c_int and len(L.run_step_converged.argtypes) == 8, "run_step_converged(...) -> int")
        expect(len(L.set_boundary_dirichlet.argtypes) == 2 and L.destroy_solver.argtypes == [ctypes.c_void_p], "boundary / destroy prototypes")
        expect(bridge._has_converged_api and bridge._has_boundary_api and bridge._destroy_symbol == "destroy_solver", "optional APIs detected")
    # a wrong digest must be refused by the same gate
    os.environ["SCPN_SOLVER_LIB_SHA256"] = "0" * 64
    refused = HPCBridge()
    expect(not refused.is_available() and refused.load_error, "a wrong SHA-256 is refused")
    bridge.close()
    print("reference bridge check:", "FAILED" if failures else "passed")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
