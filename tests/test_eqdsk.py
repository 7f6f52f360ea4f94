"""GEQDSK in / out: reader against the reference's own parse of one of its public DIII-D files
(data fixture tests/golden/diiid_lmode_1MA.geqdsk), writer round trip, and -- on the GPU -- file -> config -> solve."""
import numpy as np
import pytest

from scpn_fusion_core_amd.eqdsk import GEqdsk, read_geqdsk, write_geqdsk
from tests.conftest import ROOT
from tests._problems import configs_from

FILE = ROOT / "tests" / "golden" / "diiid_lmode_1MA.geqdsk"


def test_reader_matches_reference_parse(golden):
    g = golden("eqdsk.npz")
    eq = read_geqdsk(FILE)
    got = [eq.nw, eq.nh, eq.rdim, eq.zdim, eq.rcentr, eq.rleft, eq.zmid, eq.rmaxis, eq.zmaxis, eq.simag, eq.sibry, eq.bcentr,
           eq.current]
    assert got == list(g["scalars"])
    for name in ("fpol", "pres", "ffprime", "pprime", "qpsi", "psirz", "rbdry", "zbdry", "rlim", "zlim"):
        assert np.array_equal(getattr(eq, name), g[name]), name
    cfg = eq.to_config("diiid_lmode_1MA")
    ref_cfg = configs_from(g)["diiid_lmode_1MA"]
    assert cfg["grid_resolution"] == ref_cfg["grid_resolution"] and cfg["dimensions"] == ref_cfg["dimensions"]
    assert cfg["physics"] == ref_cfg["physics"] and cfg["coils"] == []
    assert cfg["free_boundary"]["target_flux_points"] == ref_cfg["free_boundary"]["target_flux_points"]
    assert cfg["free_boundary"]["psi_boundary"] == ref_cfg["free_boundary"]["psi_boundary"]


def test_write_read_round_trip(tmp_path):
    eq = read_geqdsk(FILE)
    out = tmp_path / "rt.geqdsk"
    write_geqdsk(eq, out)
    eq2 = read_geqdsk(out)
    assert (eq2.nw, eq2.nh) == (eq.nw, eq.nh)
    for name in ("rdim", "zdim", "rcentr", "rleft", "zmid", "rmaxis", "zmaxis", "simag", "sibry", "bcentr", "current"):
        assert getattr(eq2, name) == pytest.approx(getattr(eq, name), rel=1e-9, abs=1e-300)
    for name in ("fpol", "pres", "ffprime", "pprime", "qpsi", "psirz", "rbdry", "zbdry", "rlim", "zlim"):
        np.testing.assert_allclose(getattr(eq2, name), getattr(eq, name), rtol=1e-9, atol=0)
    # the file we wrote is bit-stable under a second round trip
    out2 = tmp_path / "rt2.geqdsk"
    write_geqdsk(eq2, out2)
    assert out.read_text() == out2.read_text()


def test_reader_rejects_malformed(tmp_path):
    bad = tmp_path / "bad.geqdsk"
    bad.write_text("")
    with pytest.raises(ValueError):
        read_geqdsk(bad)
    bad.write_text("desc 0 1 1\n 1.0 2.0\n")
    with pytest.raises(ValueError):
        read_geqdsk(bad)
    lines = FILE.read_text().splitlines()
    bad.write_text("\n".join(lines[:40]) + "\n")  # truncated
    with pytest.raises(ValueError):
        read_geqdsk(bad)
    with pytest.raises(ValueError):
        write_geqdsk(GEqdsk(nw=1, nh=1), tmp_path / "x")


@pytest.mark.gpu
def test_geqdsk_drives_the_solver(golden):
    """File -> to_config -> FusionKernel (SOR lane) as in the reference's tests/test_gs_convergence.py:83-227."""
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("eqdsk.npz")
    cfg = read_geqdsk(FILE).to_config("diiid_lmode_1MA")
    cfg["solver"].update(solver_method="sor", sor_omega=1.6, max_iterations=400, convergence_threshold=1e-4, relaxation_factor=0.1)
    k = FusionKernel(cfg)
    res = k.solve_equilibrium()
    conv, its, resid = g["solve_stats"]
    assert res["iterations"] == int(its) and res["converged"] == bool(conv)
    assert res["residual"] == pytest.approx(resid, rel=1e-6)
    assert np.abs(k.Psi - g["solve_psi"]).max() <= 1e-10 * np.abs(g["solve_psi"]).max()
    k.close()


def test_public_sparc_source_convention_adapter(golden):
    """core/eqdsk.py:290-346, 513-537: the named-file adapter rescales ffprime / pprime of recognised public SPARC files
    (fixture tests/golden/sparc/sparc_1305.eqdsk, MIT licence alongside) by 2*pi and records provenance; other files and
    the default mode stay raw."""
    import json

    g = golden("eqdsk.npz")
    f = ROOT / "tests" / "golden" / "sparc" / "sparc_1305.eqdsk"
    raw = read_geqdsk(f)
    got = [raw.nw, raw.nh, raw.rdim, raw.zdim, raw.rcentr, raw.rleft, raw.zmid, raw.rmaxis, raw.zmaxis, raw.simag, raw.sibry, raw.bcentr,
           raw.current]
    assert got == list(g["sparc_scalars"])
    assert np.array_equal(raw.psirz, g["sparc_psirz"]) and np.array_equal(raw.ffprime, g["sparc_ffprime_raw"])
    assert np.array_equal(raw.pprime, g["sparc_pprime_raw"]) and np.array_equal(raw.rbdry, g["sparc_rbdry"])
    assert np.array_equal(raw.zlim, g["sparc_zlim"]) and raw.source_convention == "raw_canonical"
    ad = read_geqdsk(f, source_convention_mode="public_sparc_named_adapter")
    assert np.array_equal(ad.ffprime, g["sparc_ffprime_adapted"]) and np.array_equal(ad.pprime, g["sparc_pprime_adapted"])
    meta = json.loads(str(g["sparc_meta"]))
    assert [ad.source_convention, ad.source_convention_adapter, ad.source_convention_adapter_pass, ad.source_convention_metadata] == \
        meta["adapted"]
    none = read_geqdsk(FILE, source_convention_mode="public_sparc_named_adapter")
    assert [none.source_convention, none.source_convention_adapter, none.source_convention_adapter_pass,
            none.source_convention_metadata] == meta["no_match"]
    with pytest.raises(ValueError):
        read_geqdsk(f, source_convention_mode="bogus")
    assert np.array_equal(raw.psi_norm, np.linspace(0.0, 1.0, raw.nw))
