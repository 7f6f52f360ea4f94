"""Every tuning switch of the HIP path (XCD order, fused / one-pass kernels, lazy source, group count, defect fusion level, tile
geometry) and every experiment build must give the same bits as the default: they only change where and when the reference's arithmetic is executed."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent

SWITCHES = [
    {"SCPN_HIP_FUSED": "0"},
    {"SCPN_HIP_XCD": "0"},
    {"SCPN_HIP_LAZYSRC": "0"},
    {"SCPN_HIP_GROUPS": "2"},
    {"SCPN_HIP_DEFECT_MINLVL": "9"},
    {"SCPN_HIP_RUNAHEAD": "0"},                                 # the blocking poll every check_every iterations (rounds 1-4) instead of the run-ahead loop
    {"SCPN_HIP_RUNAHEAD": "0", "SCPN_HIP_POLL": "memcpy"},      # ... by hipMemcpyAsync + stream synchronise instead of the host-mapped word
    {"SCPN_HIP_RUNAHEAD": "1"}, {"SCPN_HIP_RUNAHEAD": "4"},     # other run-ahead depths
    {"SCPN_HIP_XCD_BLEND": "0"},
    {"SCPN_HIP_SUMFUSE": "0"},                                  # separate k_jsum_final launch instead of the sum in its first consumer
    {"SCPN_HIP_TILEGEO": "32"},                                 # tile geometries: the default of this 2-member run is 24 rows x 12 waves
    {"SCPN_HIP_TILEGEO": "48"},                                 # 48 rows x 8 waves (six rows per wave, inner sources in LDS slots): ensembles of more than 12 members
    {"SCPN_HIP_TILEGEO": "48", "SCPN_HIP_SUMFUSE": "0"},
    # the run's omega is 1.0, so its smoothers are the Gauss-Seidel instantiations (relaxation step = one fma): the general ones, each geometry
    {"SCPN_HIP_GS1": "0"},
    {"SCPN_HIP_GS1": "0", "SCPN_HIP_TILEGEO": "32"},
    {"SCPN_HIP_GS1": "0", "SCPN_HIP_TILEGEO": "48"},
    # the zero correction a coarse level starts from written by the level above and read back (default: neither), each geometry
    {"SCPN_HIP_ZEROIN": "0"},
    {"SCPN_HIP_ZEROIN": "0", "SCPN_HIP_TILEGEO": "32"},
    {"SCPN_HIP_ZEROIN": "0", "SCPN_HIP_TILEGEO": "48"},
]
# experiment builds (never shipped: `_lib.build(variant=...)` makes libscpn_hip.<variant>.so next to the product library)
# name -> (defines, tile geometry the run forces)
EXPERIMENT_BUILDS = {
    "mid40": (("TILE_MID_LZ=40", "TILE_MID_NW=10"), "32"),  # another tile geometry: 40 rows x 10 waves (the MID slot)
    "cvb0": (("CV_BLOCKS=0",), "48"),                # coarse kernel: 17^2 .. 65^2 levels as column pairs with three LDS reads per update (rounds 2-3), separate defect / prolongation passes
}


def run(extra, n=257):
    env = {k: v for k, v in os.environ.items() if not k.startswith("SCPN_HIP_") or k == "SCPN_HIP_LIB"}
    env.update(extra)
    env["PYTHONPATH"] = str(HERE.parent) + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, str(HERE / "_variant_solve.py"), str(HERE), str(n)], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_switches_are_bit_identical():
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()  # fail loudly: no fallback path exists
    base = run({})
    assert base["iterations"] == [12, 12]
    for sw in SWITCHES:
        got = run(sw)
        # fields bit-identical; the mean |dpsi| reduction groups its partial sums by tile, so it may differ in the last ulp
        assert (got["psi"], got["jphi"], got["iterations"], got["special"]) == (base["psi"], base["jphi"], base["iterations"], base["special"]), (sw, got, base)
        assert got["residual"] == pytest.approx(base["residual"], rel=1e-13), sw


def test_experiment_builds_are_bit_identical():
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()
    base = run({})
    for name, (defines, geo) in EXPERIMENT_BUILDS.items():
        lib = _lib.build(variant=name, defines=defines)
        got = run({"SCPN_HIP_LIB": str(lib), "SCPN_HIP_TILEGEO": geo})
        assert (got["psi"], got["jphi"], got["iterations"], got["special"]) == (base["psi"], base["jphi"], base["iterations"], base["special"]), (name, got, base)


@pytest.mark.parametrize("env", [{"SCPN_HIP_TILEGEO": "32"}, {"SCPN_HIP_TILEGEO": "48"}], ids=["tile_32_rows", "tile_48_rows"])
def test_operator_parity_suite_under_each_tile_geometry(env):
    """The tile geometry is chosen by the size of a stream group (a level-0 launch of at most ~220 tiles: 24 rows x 12 waves; up to
    16 members: 32 x 8; more: 48 x 8), so the default run of tests/test_gpu_parity.py pins mostly the first against the reference
    fixtures and the oracle.  Run the same suite with each of the other two forced."""
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()
    e = {k: v for k, v in os.environ.items() if not k.startswith("SCPN_HIP_") or k == "SCPN_HIP_LIB"}
    e.update(env)
    out = subprocess.run([sys.executable, "-m", "pytest", str(HERE / "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                         env=e, capture_output=True, text=True, timeout=900, cwd=str(HERE.parent))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
