"""ctypes binding of libscpn_hip.so (include/scpn_hip.h).  No PyTorch, no fallback.

The product path is HIP-only: if the shared library is missing or no MI355X is visible every
operator raises :class:`HipUnavailableError` -- nothing is ever routed to a CPU implementation.
"""

from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libscpn_hip.so"
CSRC = PKG_DIR / "csrc"
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++20", "-ldl"]

SCPN_OK, SCPN_E_ARG, SCPN_E_DEVICE, SCPN_E_DIVERGED = 0, -1, -2, -3


class HipUnavailableError(RuntimeError):
    """libscpn_hip.so is not built / cannot be loaded / no HIP device is present."""


_d, _i = C.c_double, C.c_int
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_vp = C.c_void_p


class PicardParams(C.Structure):
    _fields_ = [
        ("z_min_cfg", _d), ("r_center", _d), ("mu0", _d),
        ("method", _i), ("sor_omega", _d), ("alpha", _d), ("tol", _d), ("max_iter", _i),
        ("require_gs_residual", _i), ("gs_tol", _d),
        ("hmode", _i), ("ped_p", _d * 4), ("ped_ff", _d * 4),
        ("preserve_initial_state", _i), ("has_boundary_flux", _i), ("external_profile_mode", _i),
        ("check_every", _i), ("fail_on_diverge", _i), ("skip_seed", _i), ("anderson_m", _i), ("xpoint_saddle", _i),
    ]


class PicardResult(C.Structure):
    _fields_ = [
        ("converged", _i), ("iterations", _i), ("diverged", _i), ("reserved", _i),
        ("residual", _d), ("gs_residual", _d), ("gs_residual_best", _d),
        ("psi_axis", _d), ("psi_bnd", _d), ("x_r", _d), ("x_z", _d), ("i_current", _d),
    ]


class NewtonOpts(C.Structure):
    _fields_ = [
        ("precond", _i), ("line_search", _i), ("line_search_c", _d), ("max_backtracks", _i), ("warmup_steps", _i),
        ("newton_alpha", _d), ("gmres_restart", _i), ("gmres_maxiter", _i), ("gmres_atol", _d), ("gmres_rtol", _d),
        ("poll_every", _i), ("reserved", _i),
    ]


class NewtonResult(C.Structure):
    _fields_ = [
        ("converged", _i), ("iterations", _i), ("diverged", _i), ("n_hist", _i), ("n_gs_hist", _i),
        ("ls_attempts", _i), ("ls_accepts", _i), ("ls_rejects", _i),
        ("gmres_nonconverged", _i), ("gmres_breakdown", _i), ("gmres_last_info", _i), ("reserved", _i),
        ("gmres_inner", C.c_long), ("residual", _d), ("gs_residual", _d), ("gs_residual_best", _d),
    ]


# symbol -> (restype, argtypes); the single source of truth also checked by tests/test_abi_symbols.py
SIGNATURES = {
    # legacy libscpn_solver ABI
    "create_solver": (_vp, [_i, _i, _d, _d, _d, _d]),
    "set_boundary_dirichlet": (None, [_vp, _d]),
    "run_step": (None, [_vp, _dp, _dp, _i, _i]),
    "run_step_converged": (_i, [_vp, _dp, _dp, _i, _i, _d, _d, C.POINTER(_d)]),
    "destroy_solver": (None, [_vp]),
    "delete_solver": (None, [_vp]),
    # probes
    "scpn_device_count": (_i, []),
    "scpn_device_info": (_i, [C.c_char_p, _i]),
    "scpn_set_device": (_i, [_i]),
    "scpn_last_error": (C.c_char_p, []),
    "scpn_version": (C.c_char_p, []),
    # operator library
    "scpn_mg_smooth": (_i, [_dp, _dp, _dp, _i, _i, _i, _d, _d, _d, _i]),
    "scpn_mg_residual": (_i, [_dp, _dp, _dp, _i, _i, _i, _d, _d, _dp]),
    "scpn_residual_linf": (_i, [_dp, _dp, _dp, _i, _i, _i, _d, _d, _dp]),
    "scpn_gs_residual_rms": (_i, [_dp, _dp, _dp, _i, _i, _i, _d, _d, _dp]),
    "scpn_restrict_full_weight": (_i, [_dp, _i, _i, _i, _dp]),
    "scpn_prolongate_bilinear": (_i, [_dp, _i, _i, _i, _i, _i, _dp]),
    "scpn_multigrid_vcycle": (_i, [_dp, _dp, _dp, _i, _i, _i, _d, _d, _d, _i, _i, _i]),
    "scpn_multigrid_solve": (_i, [_dp, _dp, _dp, _dp, _i, _i, _i, _d, _i, _d, _i, _i, _i, _dp, _ip, _ip]),
    "scpn_jacobi_step": (_i, [_dp, _dp, _dp, _i, _i, _i, _d, _d, _dp]),
    "scpn_sor_step": (_i, [_dp, _dp, _dp, _i, _i, _i, _d, _d, _d, _dp]),
    "scpn_coil_flux": (_i, [_dp, _dp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _d, _i, _dp]),
    "scpn_green_points": (_i, [_i, _dp, _dp, _i, _dp, _dp, _dp]),
    "scpn_verify_division": (_i, [_dp, _dp, _i, _dp, _dp]),
    "scpn_copy_bandwidth_probe": (_i, [C.c_long, _i, C.POINTER(_d)]),
    "scpn_bandwidth_probe": (_i, [_i, C.c_long, _i, C.POINTER(_d)]),
    "scpn_np_delta_star": (_i, [_dp, _i, _i, _dp, _d, _d, _dp]),
    "scpn_np_equation_residual": (_i, [_dp, _i, _i, _dp, _dp, _d, _d, _d, _dp]),
    "scpn_tridiag_solve": (_i, [_i, _i, _vp, _dp, _vp, _dp, _dp, _vp]),
    "scpn_current_diffusion_step": (_i, [_i, _i, _dp, _dp, _d, _d, _d, _d, _dp, _dp, _d, _dp, _dp, _vp, _vp]),
    "scpn_flux_evolution": (_i, [_i, _i, _i, _d, _dp, _dp, _dp, _dp, _vp, _vp, _vp]),
    "scpn_probe_response": (_i, [_i, _dp, _dp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _dp]),
    "scpn_gs_solve_np": (_i, [_d, _d, _d, _d, _i, _i, _dp, _dp, _vp, _d, _d, _i, _i, _d, _d, _d, _dp]),
    # resident ensemble solver
    "scpn_gs_create": (_i, [C.POINTER(_vp), _i, _i, _i, _dp, _dp, _d, _d]),
    "scpn_gs_destroy": (None, [_vp]),
    "scpn_gs_set_psi": (_i, [_vp, _i, _dp]),
    "scpn_gs_get_psi": (_i, [_vp, _i, _dp]),
    "scpn_gs_get_jphi": (_i, [_vp, _i, _dp]),
    "scpn_gs_set_jphi": (_i, [_vp, _i, _dp]),
    "scpn_gs_get_source": (_i, [_vp, _i, _dp]),
    "scpn_gs_set_source": (_i, [_vp, _i, _dp]),
    "scpn_gs_set_boundary_flux": (_i, [_vp, _i, _dp]),
    "scpn_gs_get_boundary_flux": (_i, [_vp, _i, _dp]),
    "scpn_gs_set_current_target": (_i, [_vp, _i, _dp]),
    "scpn_gs_set_vacuum_coils": (_i, [_vp, _i, _vp, _vp, _vp]),
    "scpn_gs_set_seed": (_i, [_vp, _vp]),
    "scpn_gs_picard_solve": (_i, [_vp, C.POINTER(PicardParams), C.POINTER(PicardResult), _vp, _vp, _vp]),
    "scpn_gs_newton_solve": (_i, [_vp, C.POINTER(PicardParams), C.POINTER(NewtonOpts), C.POINTER(NewtonResult), _vp, _vp]),
    "scpn_gs_free_boundary_solve": (_i, [_vp, C.POINTER(PicardParams), _i, _vp, _vp, _vp, _vp, _i, _d, _vp, _vp, _vp,
                                         C.POINTER(PicardResult)]),
    "scpn_gs_free_boundary_solve_newton": (_i, [_vp, C.POINTER(PicardParams), C.POINTER(NewtonOpts), _i, _vp, _vp, _vp, _vp, _i, _d, _vp,
                                                _vp, _vp, C.POINTER(PicardResult)]),
    "scpn_gs_rb_sor": (_i, [_vp, _d, _i]),
    "scpn_gs_vcycle": (_i, [_vp, _d, _i, _i, _i]),
    "scpn_gs_compute_b_field": (_i, [_vp, _i, _dp, _dp]),
    "scpn_gs_sample_psi": (_i, [_vp, _i, _dp, _dp, _dp]),
    "scpn_gs_map_profiles": (_i, [_vp, _d, _i, _dp, _dp, _dp, _vp, _vp]),
    "scpn_gs_snapshot_psi": (_i, [_vp]),
    "scpn_gs_flux_change": (_i, [_vp, _dp]),
    "scpn_gs_fill_psi": (_i, [_vp, _d]),
    "scpn_gs_pack_psi_device": (_i, [_vp, C.POINTER(_vp)]),
    "scpn_gs_profile": (_i, [_vp, _i]),
    "scpn_gs_profile_read": (_i, [_vp, C.c_char_p, C.POINTER(_d), C.POINTER(C.c_long)]),
    "scpn_gs_profile_names": (_i, [_vp, C.c_char_p, _i]),
    "scpn_gs_sync": (_i, [_vp]),
    "scpn_gs_num_groups": (_i, [_vp]),
    "scpn_operator_pool_stats": (_i, [C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(_i)]),
    "scpn_gs_set_arith": (_i, [_vp, _i]),
    "scpn_gs_get_arith": (_i, [_vp]),
    # ensemble gather over RCCL
    "scpn_comm_unique_id": (_i, [C.c_char_p]),
    "scpn_comm_prepare": (_i, [_i, C.POINTER(_i)]),
    "scpn_comm_init": (_i, [_i, _i, C.c_char_p, _i]),
    "scpn_comm_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "scpn_comm_allgather_psi": (_i, [_vp, _i, _vp]),
    "scpn_comm_wait": (_i, []),
    "scpn_comm_allreduce_max": (_i, [C.POINTER(_d)]),
    "scpn_comm_finalize": (_i, []),
    "scpn_device_alloc": (_i, [C.c_long, C.POINTER(_vp)]),
    "scpn_device_free": (_i, [_vp]),
    "scpn_device_download": (_i, [_vp, _vp, C.c_long]),
}


def build(force: bool = False, verbose: bool = False, *, variant: str | None = None, defines: tuple[str, ...] = ()) -> Path:
    """Compile csrc/scpn_hip.hip for gfx950 into the in-tree libscpn_hip.so (hipcc cross-compiles without a GPU).

    ``variant`` / ``defines`` build an EXPERIMENT library ``libscpn_hip.<variant>.so`` with extra ``-D`` switches (another tile
    geometry, in-kernel stamps ...) next to the shipped one; it is only ever loaded through ``SCPN_HIP_LIB`` by ``tools/`` and
    by the bit-identity test of the switches.  The shipped library instantiates exactly what the default path launches."""
    target = LIB_PATH if variant is None else LIB_PATH.with_name(f"libscpn_hip.{variant}.so")
    srcs = [CSRC / "scpn_hip.hip", *sorted(CSRC.glob("*.hip.h")), PKG_DIR.parent / "include" / "scpn_hip.h"]
    if not force and target.exists() and all(target.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return target
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # build next to the target and rename: concurrent builders (one rank per GPU) never see a half-written library
    tmp = target.with_name(f".{target.name}.{os.getpid()}.tmp")
    cmd = [hipcc, *HIPCC_FLAGS, *[f"-D{d}" for d in defines], "-o", str(tmp), str(CSRC / "scpn_hip.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode:
        print(" ".join(cmd), res.stdout, res.stderr, sep="\n")
    if res.returncode:
        tmp.unlink(missing_ok=True)
        raise RuntimeError(f"hipcc failed building {target.name}:\n{res.stderr[-2000:]}")
    os.replace(tmp, target)
    return target


_LIB = None


def lib() -> C.CDLL:
    """Load libscpn_hip.so (symbols only -- no device is touched)."""
    global _LIB
    if _LIB is None:
        path = Path(os.environ.get("SCPN_HIP_LIB", LIB_PATH))
        if not path.exists():
            raise HipUnavailableError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        try:
            L = C.CDLL(str(path))
        except OSError as exc:  # pragma: no cover - depends on the host
            raise HipUnavailableError(f"cannot load {path}: {exc}") from exc
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def device_count() -> int:
    try:
        return int(lib().scpn_device_count())
    except HipUnavailableError:
        return 0


def gpu_available() -> bool:
    """True when a HIP device is visible and ``SCPN_DISABLE_GPU`` is not set (core/_multi_compat.py:101-120)."""
    if os.environ.get("SCPN_DISABLE_GPU", "").strip().lower() in ("1", "true", "yes", "on"):
        return False
    return device_count() > 0


def require_gpu() -> C.CDLL:
    L = lib()
    if L.scpn_device_count() <= 0:
        raise HipUnavailableError("no HIP device visible: the HIP path has no CPU fallback")
    return L


def device_info() -> str | None:
    try:
        L = lib()
    except HipUnavailableError:
        return None
    buf = C.create_string_buffer(256)
    if L.scpn_device_info(buf, 256) != SCPN_OK:
        return None
    return buf.value.decode()


def copy_bandwidth_probe(nbytes: int = 1 << 30, reps: int = 10, mode: int = 0) -> float:
    """GB/s of a device copy (read + written bytes; mode 0 grid-stride, 2 contiguous chunks with 8 loads in flight, 3 the same
    with non-temporal stores, 4 hipMemcpyDtoD) or of a read-only sum (mode 1) over ``nbytes`` (default 1 GiB: four times the
    Infinity Cache), best of several grid sizes."""
    out = _d(0.0)
    check(require_gpu().scpn_bandwidth_probe(mode, nbytes, reps, C.byref(out)), "scpn_bandwidth_probe")
    return float(out.value)


def bandwidth_probes(nbytes: int = 1 << 30, reps: int = 10) -> dict:
    """All probe modes: ``{"copy_grid_stride", "read", "copy_chunked", "copy_chunked_nt", "memcpy_dtod", "three_reads_one_write"}`` in GB/s."""
    names = ("copy_grid_stride", "read", "copy_chunked", "copy_chunked_nt", "memcpy_dtod", "three_reads_one_write")
    return {n: round(copy_bandwidth_probe(nbytes, reps, m), 1) for m, n in enumerate(names)}


def last_error() -> str:
    return (lib().scpn_last_error() or b"").decode()


def check(rc: int, what: str = "") -> None:
    """Map C return codes to the reference's exceptions (ValueError for argument errors)."""
    if rc == SCPN_OK:
        return
    msg = last_error() or what
    if rc == SCPN_E_ARG:
        raise ValueError(msg)
    if rc == SCPN_E_DIVERGED:
        raise RuntimeError(msg)
    raise HipUnavailableError(msg)


def as_f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)
