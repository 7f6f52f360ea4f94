"""Stand-in for the reference's PyO3 module ``scpn_fusion_rs`` that serves ONLY its GPU tier from libscpn_hip.so.

Put this directory's parent on ``sys.path`` (``PYTHONPATH=/path/to/this/repo/integration:/path/to/this/repo``) on a machine
where the Rust wheel is not installed and the reference's callers run unchanged:

* ``benchmarks/bench_gpu_gs_solver.py:82-113`` imports ``PyGpuSolver, py_gpu_available, py_gpu_info``;
* ``core/_multi_compat.py:101-120`` probes ``py_gpu_available`` for the GPU tier, and
  ``core/_multi_compat_providers.py:755-792`` builds its GPU providers on ``PyGpuSolver``;
* ``core/_multi_compat_providers.py:322-344`` calls ``multigrid_vcycle`` for the Rust tier -- exported too, because an importable
  ``scpn_fusion_rs`` flips that tier to "available" (SURVEY.md section 8b-B2).

Everything else the real wheel exports (transport, control, ... -- out of this repo's scope) is deliberately absent: a caller
that needs it gets the ImportError / AttributeError it would get without the wheel.
"""
from scpn_fusion_core_amd.gpu_tier import PyGpuSolver, multigrid_vcycle, py_gpu_available, py_gpu_info  # noqa: F401

__all__ = ["PyGpuSolver", "py_gpu_available", "py_gpu_info", "multigrid_vcycle"]
