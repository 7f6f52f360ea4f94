"""The bench.py output contract, checked on the committed line of the last measured run (profiles/*_bench.json): keys and
types the driver parses, the roofline / cpu_baseline objects, and internal consistency (value = members / time,
frac = achieved / peak).  No GPU needed; bench.py itself runs only on the GPU box."""
import json

from tests.conftest import ROOT


def _latest():
    files = sorted((ROOT / "profiles").glob("r*_bench.json"))
    assert files, "no committed bench line under profiles/"
    return json.loads(files[-1].read_text())


def test_bench_line_contract():
    d = _latest()
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["dtype"] == "f64"
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    members = d["config"]["global_members"]
    assert abs(d["value"] - members / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["config"]["converged_members_rank0"] == d["config"]["members_rank0"]
    # the workload is SURVEY 8(d) cfg 5 in full: currents, I_p AND the kappa scan through the coil heights
    assert "kappa in [1.5, 1.9]" in d["config"]["workload"] and 1.5 <= d["config"]["kappa_range_rank0"][0] < d["config"]["kappa_range_rank0"][1] <= 1.9
    # which arithmetic the headline was timed in, and the other mode beside it (same handle, after the clock)
    assert d["config"]["arithmetic"] in ("exact", "contracted")
    am = d["arithmetic_modes"]
    assert am[d["config"]["arithmetic"]]["headline"] is True and am[d["config"]["arithmetic"]]["value"] == d["value"]
    other = "exact" if d["config"]["arithmetic"] == "contracted" else "contracted"
    assert am[other]["headline"] is False and am[other]["converged_members"] == d["config"]["members_rank0"] and am[other]["value"] > 0
    assert am["psi_rel_linf_between_modes"] < 1e-10 and am["members_with_equal_iteration_counts"] >= d["config"]["members_rank0"] - 2
    # one record per rank: what makes an N > 1 curve attributable (imbalance / gather / launch path)
    pr = d["per_rank"]
    assert isinstance(pr, list) and len(pr) == d["n_gpus"] and [row["rank"] for row in pr] == list(range(d["n_gpus"]))
    for row in pr:
        assert {"members", "solve_ms", "gather_wait_ms", "final_barrier_ms", "picard_iterations_max", "picard_iterations_min", "converged_members"} <= set(row)
        assert row["solve_ms"] <= d["ms_per_step"] * 1.02 and row["converged_members"] == row["members"]
    assert ("gather_ms" in d) == (d["n_gpus"] > 1)
    if d["scaling"] == "weak":
        assert members == d["config"]["members_per_gpu"] * d["n_gpus"]
        s1 = d["strong_scaling_n1"]  # the other scaling label's N = 1 value rides on the weak line
        assert s1["scaling"] == "strong" and s1["total_members"] == s1["converged_members"] == 256 and s1["n_gpus"] == 1
        assert abs(s1["value"] - 256 / (s1["ms_per_step"] * 1e-3)) <= 0.01 * s1["value"]


def test_roofline_and_cpu_baseline_objects():
    d = _latest()
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] == 8000.0
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["kernel"].startswith("smooth") and r["avg_launch_us"] > 0 and r["launches"] > 0
    # algorithmic bytes per launch = per-point figure x points x members, averaged over the levels the kernel ran on
    assert r["alg_bytes_per_launch"] > r["alg_bytes_per_point"] * 513 * 513 * r["members_per_launch"] / 3
    # the HBM fraction is a statement about bytes really moved: it can never exceed the peak
    assert r["frac"] <= 1.0 and all(v["achieved"] <= r["peak"] for v in r["per_level"].values())
    # measured bandwidth beside the nominal peak, and the full-fill launches (all members in one launch) beside the 16-member ones
    assert 1000.0 < r["measured_copy_gbs"] < r["peak"] and r["measured_read_gbs"] > 1000.0
    full = r["one_group_all_members"]
    assert full["members_per_launch"] == d["config"]["members_per_gpu"]
    for k in full["kernels"].values():
        assert 0.0 < k["frac"] <= 1.0 and abs(k["frac"] - k["achieved"] / r["peak"]) < 1e-3
    lim = r["limiter"]  # what limits the kernel: keys COMPUTED from the committed SQ counters of that kernel (no prose, no hard-coded figure)
    assert isinstance(lim, dict) and "counters" in lim and lim["evidence"] and "class" not in lim and lim["arithmetic"] == d["config"]["arithmetic"]
    assert lim["halo_redundancy"] is None or 1.0 < lim["halo_redundancy"] < 2.5
    if lim["counters"] is not None:
        assert lim["counters"]["source"].startswith("profiles/") and 0.0 < lim["counters"]["valu_busy"] <= 1.0
        assert 0.0 < lim["valu_issue_frac_of_launch"] <= 1.0 and lim["counters"]["valu_issue_us"] > 0
        # the counters are those of the kernel instantiation of the timed arithmetic (contracted instantiations end in `true>`)
        assert lim["counters"]["kernel"].rstrip().endswith("true>") == (d["config"]["arithmetic"] == "contracted")
    it = r["iteration"]  # HBM bytes of one Picard iteration (all its kernels, PMC) / iteration time
    assert it is None or (0.0 < it["frac"] <= 1.0 and abs(it["frac"] - it["achieved"] / it["peak"]) < 1e-3 and it["kernels_counted"])
    mb = r["measured_bandwidth"]
    assert {"copy_chunked", "read", "memcpy_dtod"} <= set(mb) and r["measured_copy_gbs"] == max(v for k, v in mb.items() if k.startswith("copy") or k == "memcpy_dtod")
    n = d["newton_fd_mg"]  # the Newton lane beside the metric: same ensemble, same converged equilibria
    assert n["converged_members"] == d["config"]["members_per_gpu"] and n["psi_rel_linf_vs_picard"] < 1e-7 and n["value"] > d["value"]
    cf = d["configs"]  # one figure per BASELINE config
    assert cf["config2_fixed_boundary_257"]["converged_members"] == cf["config2_fixed_boundary_257"]["members"]
    assert cf["config3_single_free_boundary_513"]["diiid_like"]["picard"]["converged"] and cf["config4_transport_coupled_loop_513"]["value"] > 0
    assert cf["smoother_200_sweeps_513"]["value"] > 0 and cf["smoother_200_sweeps_513"]["published_reference"]["numpy_s"] == 18.85
    assert r["bound_observed"] is None or isinstance(r["bound_observed"], str)  # what the committed counters say limits the kernel
    for key in ("config2_fixed_boundary_257", "config4_transport_coupled_loop_513"):
        assert "vacuum field per handle" in cf[key]["memoised_setup"]
    pg = cf["polyglot_picard_jacobi_513"]  # north_star's literal comparator: the polyglot lane at 513^2, device vs its CPU twins
    assert pg["toml_parameters"]["hip_ms"] > 0 and pg["vs_cpu"]["psi_axis_equal"] is True and pg["vs_cpu"]["x_numpy_one_core"] > 1.0
    assert cf["dispatcher_multigrid_solve_257"]["converged"] and cf["dispatcher_multigrid_solve_257"]["dispatch_host_buffers_ms"] > 0
    assert cf["arithmetic"] == d["config"]["arithmetic"]
    c = d["cpu_baseline"]
    assert c["arithmetic_of_the_compared_gpu_run"] == d["config"]["arithmetic"] and c["picard_iterations_max_abs_difference"] <= 1
    assert c["polyglot_513"]["numpy_vs_c_port_psi_rel_linf"] < 1e-12 and c["polyglot_513"]["numpy_one_core_ms"] > 0
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # every member the CPU leg solved is compared with the device's result of the timed run
    assert c["host_cores"] >= c["cores"] == c["members_compared"] == len(c["psi_rel_linf_vs_gpu_per_member"])
    assert c["picard_iterations_equal_on_all_compared_members"] is True and abs(c["psi_rel_linf_vs_gpu"] - max(c["psi_rel_linf_vs_gpu_per_member"])) <= 1e-3 * c["psi_rel_linf_vs_gpu"] < 1e-12
    p = d["parity"]  # the strict-parity gate with all five thresholds, on both lanes
    assert p["pass"] is True and set(p["thresholds"]) == {"psi_n_rmse", "axis_error_m", "boundary_max_abs_error_wb",
                                                            "current_closure_relative_error", "xpoint_psi_n_error"}
    assert len(p["free_boundary_lane"]["checks"]) == 5 and all(p["free_boundary_lane"]["checks"].values())
    for case in p["native_profile_source_lane"]["cases"].values():
        assert len(case["checks"]) == 5 and case["pass"] and case["psi_max_abs_error_wb"] <= 1e-10 and case["psi_n_rmse"] <= 1e-9


def test_pmc_traffic_file_matches_the_families_bench_reports():
    pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    assert pmc["grid"] == 513
    fam = _latest()["roofline"]["families_ms"]
    family_of = {"jsum": "source", "jsum_final": "source"}  # k_jsum + k_jsum_final are timed together as the family "source"
    for name in pmc["hbm_bytes_per_member_launch"]:
        assert family_of.get(name, name) in fam, name
