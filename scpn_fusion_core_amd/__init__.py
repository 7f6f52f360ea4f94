"""scpn_fusion_core_amd -- MI355X-native Grad-Shafranov equilibrium hot path.

Hand-written gfx950 HIP kernels behind a C ABI (include/scpn_hip.h), driven from Python by ctypes.
Mirrors the reference's operator / plugin surface for this path only:

* :mod:`.multigrid_solve`  -- free functions of core/multigrid_solve.py
* :mod:`.fusion_kernel`    -- ``FusionKernel``-compatible equilibrium kernel (Picard, free boundary)
* :mod:`.ensemble`         -- batched ensemble of independent equilibria (one process per GPU)
* :mod:`.multi_compat`     -- BackendTier / register_kernel / dispatch seam + tier providers
* :mod:`.gpu_tier`         -- ``scpn_fusion_rs``-shaped GPU surface (PyGpuSolver, py_gpu_available, ...)
* :mod:`.hpc_bridge`       -- loader for the legacy libscpn_solver C ABI with the SHA-256 trust gate
"""

from ._lib import HipUnavailableError, build, device_info, gpu_available  # noqa: F401

__version__ = "0.1.0"
