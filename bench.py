#!/usr/bin/env python3
"""bench.py -- converged free-boundary Grad-Shafranov equilibria / s on MI355X.

Metric (BASELINE.json): converged equilibria/s, 513x513 free-boundary, Picard tol 1e-8.
Workload: an ensemble of `--batch` (default 32 = 256 / 8 GPUs, BASELINE config 5) independent
ITER-like free-boundary equilibria per GPU on a 513x513 grid (validation/benchmark_vs_freegs.py:157
geometry; coil currents I_k*(1+0.1u) and plasma current Ip*(1+0.1u), u~U(-1,1), rng 2026), each solved exactly as the reference's
``solve_free_boundary(coils, max_outer_iter=20, tol=1e-4, optimize_shape=False)`` with
``solver_method="multigrid"``, sor_omega=1.0, relaxation_factor=0.12, convergence_threshold=1e-8.
One "step" = one full solve of the resident ensemble (Green's-function boundary flux, seed, ~2 outer
x ~240 Picard iterations of one V-cycle each, all on the device).  Inputs (coil table, psi=0) are
resident in HBM before the timed region; results stay in HBM (the RCCL gather for N>1 is inside).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (torchrun is only the process launcher:
    bench.py and the package import no torch; ranks meet over TCP and gather over RCCL through the C ABI)

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, HIP-event timed on the
solver's own stream) and "cpu_baseline" (the CPU oracle on the host cores, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec; the copy bandwidth measured on this very GPU is printed beside it


# Bytes per grid point one launch of a kernel family must move (fp64, fused kernels: DESIGN.md section 5) and the SURVEY 8(d)
# figure of the same work done as separate passes.  smooth_*_pre = 3 RB-SOR sweeps + defect + full-weight restriction
# (psi, src in; psi out; coarse rhs out = 8/4 -- the zero coarse correction is no longer written since round 4, and on levels
# >= 1 the zero input is not read: SMOOTH_PRE_COARSE_LEVEL_BYTES); smooth_*_post = prolong-add + 3 sweeps (coarse e in = 8/4).
FAMILY_BYTES = {"smooth_pre": (26.0, 106.0), "smooth_post": (26.0, 90.0), "smooth": (24.0, 72.0),
                "blend_fused": (32.0, 40.0), "source": (16.0, 24.0)}
# contracted arithmetic, L-mode: source on the fly (LazySrc::jfly) -- J_raw is never stored: the level-0 pre-smoother
# reads psi ONCE (it is both its input and the field the source is rebuilt from), the tail rebuilds the source from `cur`, k_jsum only reads
FAMILY_BYTES_JFLY = {"smooth_pre": (18.0, 106.0), "smooth_post": (26.0, 90.0), "smooth": (24.0, 72.0),
                     "blend_fused": (24.0, 40.0), "source": (8.0, 24.0)}
SMOOTH_PRE_COARSE_LEVEL_BYTES = 18.0  # src in, psi out, coarse rhs out
# cells a tile loads / cells it keeps (48 x 128 register tile, halo 8 / 6; the tail's 16 x 128 tile keeps 14 x 124)
HALO_REDUNDANCY = {"smooth_pre": (48 * 128) / (32 * 112), "smooth_post": (48 * 128) / (36 * 116), "blend_fused": (16 * 128) / (14 * 124)}
ARITH = {"mode": "exact"}  # set by main(): which byte table prices the launches


def family_table() -> dict:
    return FAMILY_BYTES_JFLY if ARITH["mode"] == "contracted" else FAMILY_BYTES


def family_bytes_per_point(family: str, lvl: int) -> float:
    return SMOOTH_PRE_COARSE_LEVEL_BYTES if (family == "smooth_pre" and lvl >= 1) else family_table().get(family, (16.0, 16.0))[0]


def counters_of(kernel_symbol: str) -> dict | None:
    """SQ counter summary of the dominant kernel from the committed profile the traffic figures come from
    (profiles/<tag>_sq_by_kernel.csv, tag recorded in profiles/pmc_traffic.json): never a hard-coded figure."""
    import csv

    try:
        tag = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get("tag")
        path = ROOT / "profiles" / f"{tag}_sq_by_kernel.csv"
        want = {"smooth_pre": "k_smooth_tile<3, false, true", "smooth_post": "k_smooth_tile<3, true, false",
                "smooth": "k_smooth_tile<3, false, false", "blend_fused": "k_blend_tile", "source": "k_jsum<"}.get(kernel_symbol, kernel_symbol)
        rows = [r for r in csv.DictReader(open(path)) if want in r["kernel"]]
        # the trace may hold the kernels of both arithmetic modes: the contracted instantiations carry a trailing `true` template argument
        fast = [r for r in rows if r["kernel"].rstrip().endswith("true>")]
        rows = (fast or rows) if ARITH["mode"] == "contracted" else ([r for r in rows if r not in fast] or rows)
        if not rows:
            return None
        r = max(rows, key=lambda x: float(x["grid_size"]))  # the level-0 launch shape
        pick = {k: float(r[k]) for k in ("valu_busy", "lds_busy", "waves_per_simd", "wait_frac", "stall_frac", "issue_frac", "clock_GHz", "insts_valu",
                                         "median_us") if r.get(k)}
        if "insts_valu" in pick and pick.get("clock_GHz"):
            # wave-instructions x 4 cycles each / (1 024 SIMDs x clock): the time the launch needs for pure VALU issue on the whole chip
            pick["valu_issue_us"] = round(pick["insts_valu"] * 4.0 / (1024.0 * pick["clock_GHz"] * 1e3), 2)
        pick.update({"source": f"profiles/{tag}_sq_by_kernel.csv", "kernel": r["kernel"], "grid_size": int(float(r["grid_size"]))})
        return pick
    except (OSError, ValueError, KeyError):
        return None


def roofline_of(fam: dict, grid: int, members_per_launch: float, num_groups: int) -> dict | None:
    """Roofline object of the dominant kernel.  `fam` = per-family HIP-event times ({name: {ms, launches}}); families of one
    role on different levels are ONE kernel symbol in a rocprofv3 trace (smooth_L{0,1,2}_pre = one template instance), and
    the dominant kernel is the symbol with the largest summed device time, as `rocprofv3 --kernel-trace --stats` ranks it."""
    import re

    def symbol(name):
        m = re.match(r"smooth_(?:tile|rows)_L\d+(_pre|_post)?$", name)
        return "smooth" + (m.group(1) or "") if m else name

    def level(name):
        m = re.match(r".*_L(\d+)", name)
        return int(m.group(1)) if m else 0

    groups: dict = {}
    for name, v in fam.items():
        groups.setdefault(symbol(name), []).append((name, v))
    if not groups:
        return None
    dom = max(groups, key=lambda k: sum(v["ms"] for _, v in groups[k]))
    fused_pt, survey_pt = family_table().get(dom, (16.0, 16.0))
    try:
        pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        if pmc.get("grid") != grid:
            pmc = None
    except (OSError, ValueError):
        pmc = None
    tot_ms = tot_launch = bytes_fused = bytes_survey = traffic = bytes_r3 = 0.0
    traffic_ok = pmc is not None
    per_level = {}
    for name, v in groups[dom]:
        pts = ((grid - 1) // (2 ** level(name)) + 1) ** 2 if dom.startswith("smooth") else grid * grid
        tot_ms += v["ms"]
        tot_launch += v["launches"]
        lvl_pt = family_bytes_per_point(dom, level(name))
        bytes_fused += lvl_pt * pts * members_per_launch * v["launches"]
        bytes_survey += survey_pt * pts * members_per_launch * v["launches"]
        bytes_r3 += (28.0 if dom == "smooth_pre" else fused_pt) * pts * members_per_launch * v["launches"]  # rounds 1-3: 28 B/pt on every level
        per_member = (pmc["hbm_bytes_per_member_launch"].get(name) if pmc else None)
        if per_member is None:
            traffic_ok = False
        else:
            traffic += per_member * members_per_launch * v["launches"]
        us = 1e3 * v["ms"] / v["launches"]
        per_level[name] = {"avg_launch_us": round(us, 2), "launches": v["launches"],
                           "alg_bytes_per_point": lvl_pt, "achieved": round(lvl_pt * pts * members_per_launch / (us * 1e-6) / 1e9, 1)}
    ach = bytes_fused / (tot_ms * 1e-3) / 1e9
    ctr = counters_of(dom)
    # `bound` = the roofline `frac` is taken against (the contract's enum: this kernel has no MFMA work, its ceiling is the HBM stream of
    # its compulsory bytes); `bound_observed` = what the committed SQ / traffic counters of the same kernel say holds it back
    observed = None
    if ctr is not None and "valu_busy" in ctr:
        observed = ("valu" if ctr["valu_busy"] >= 0.85 else
                    "latency (tiles in flight x tile lifetime): neither VALU nor HBM saturated" if ach / HBM_PEAK_GBS < 0.6 else "hbm")
    return {"kernel": dom, "bound": "hbm", "bound_observed": observed,
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": round(traffic / tot_launch) if traffic_ok else None,
            "alg_bytes_per_point": fused_pt, "alg_bytes_per_launch": round(bytes_fused / tot_launch),
            # continuity with rounds 1-3: the same launches priced with the bytes the pre-smoother HAD to move then (the zero coarse
            # correction written by it and read back by the level below); round 4 removed that traffic, so `frac` prices fewer bytes
            "frac_with_round3_byte_count": round(bytes_r3 / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "survey_8d_equivalent": {"bytes_per_point_unfused": survey_pt,
                                     "gbs_equivalent": round(bytes_survey / (tot_ms * 1e-3) / 1e9, 1)},
            "avg_launch_us": round(1e3 * tot_ms / tot_launch, 2), "launches": int(tot_launch),
            "groups": num_groups, "members_per_launch": int(members_per_launch), "isolated_kernel_timing": True,
            # what the committed SQ counters of this very kernel say (profiles/<tag>_sq_by_kernel.csv): VALU issue time against the launch,
            # wave occupancy, wait / stall split; and the geometric redundancy of its tile
            "limiter": {"counters": ctr, "halo_redundancy": round(HALO_REDUNDANCY[dom], 3) if dom in HALO_REDUNDANCY else None,
                        "valu_issue_frac_of_launch": (round(ctr["valu_issue_us"] / ctr["median_us"], 3) if ctr and ctr.get("valu_issue_us") and ctr.get("median_us") else None),
                        "arithmetic": ARITH["mode"], "evidence": "profiles/r06_experiments.md, DESIGN.md section 6"},
            "per_level": per_level, "families_ms": fam}


def iteration_roofline(ms_per_step: float, steps_iterations: int, members: int) -> dict | None:
    """HBM bytes of ONE Picard iteration (PMC counters, every kernel of the iteration, per member) / the measured time of an
    iteration of the whole ensemble: how much of the HBM peak the iteration as a whole uses."""
    try:
        pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        per_member = float(sum(pmc["hbm_bytes_per_member_launch"].values()))  # each family is launched once per iteration
    except (OSError, ValueError, KeyError):
        return None
    us = 1e3 * ms_per_step / max(steps_iterations, 1)
    gbs = per_member * members / (us * 1e-6) / 1e9
    return {"hbm_bytes_per_member_iteration": round(per_member), "us_per_iteration": round(us, 2), "achieved": round(gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "kernels_counted": sorted(pmc["hbm_bytes_per_member_launch"]), "source": pmc.get("source")}


def make_cfg(n: int) -> dict:
    from scpn_fusion_core_amd.cases import iter_like_cfg

    return iter_like_cfg(n, method="multigrid", omega=1.0, tol=1e-8, max_iter=1000, alpha=0.12)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="ensemble members per GPU (weak scaling: the global ensemble is batch x gpus)")
    ap.add_argument("--total-members", type=int, default=0,
                    help="STRONG scaling: a fixed global ensemble of this many members (BASELINE config 5: 256) sharded as "
                         "ceil(total/gpus) per rank; overrides --batch and reports \"scaling\": \"strong\"")
    ap.add_argument("--grid", type=int, default=513)
    ap.add_argument("--check-every", type=int, default=8)
    ap.add_argument("--arith", choices=("exact", "contracted"), default="contracted",
                    help="arithmetic of the timed run (scpn_gs_set_arith): contracted = FMA chains, psi within 1e-10 max|psi| of the CPU "
                         "reference (tests/test_gpu_fast_arith.py); exact = the reference's operation order, bit-exact operators.  The "
                         "other mode is timed after the clock and reported under \"arithmetic_modes\"")
    ap.add_argument("--no-arith-modes", action="store_true", help="skip the timing of the other arithmetic mode (profile runs: one kernel set per trace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-newton", action="store_true", help="skip the Newton-lane figure reported beside the metric")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config figures (BASELINE configs 2-4, smoother benchmark)")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` with nobody having started the ranks: this process becomes the launcher.  It has not
    #      touched a GPU (nothing of the package is imported yet); it starts N fresh children of itself, one rank per GPU, with
    #      RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, forwards rank 0's JSON line and fails if any rank fails
    #      or if the line does not say n_gpus == N.  Under `python -m torch.distributed.run` WORLD_SIZE is already set.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        from scpn_fusion_core_amd.distributed import spawn_ranks

        rc, out = spawn_ranks([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], args.gpus)
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        for ln in out.splitlines():
            if not ln.startswith("{"):
                print(ln, file=sys.stderr)
        if not lines:
            print(f"bench.py: rank 0 printed no result line (exit code {rc})", file=sys.stderr)
            return rc or 1
        print(lines[-1])
        if json.loads(lines[-1]).get("n_gpus") != args.gpus:
            print(f"bench.py: the result line does not report n_gpus == {args.gpus}", file=sys.stderr)
            return rc or 1
        return rc

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or drop WORLD_SIZE and let "
              "bench.py start them)", file=sys.stderr)
        return 2
    # two stream groups per resident handle + a second handle alive during the Newton lane: ask the HIP runtime for 8 hardware
    # queues (default 4) before this process's first device call -- the library itself never touches the environment
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if os.environ.get("SCPN_BENCH_DEVICE") is not None:  # rehearsal of the N>1 launch on a one-GPU box: every rank on this device
        local_rank = int(os.environ["SCPN_BENCH_DEVICE"])
    comm = None
    comm_error = None
    host_sync = None
    if world > 1 or os.environ.get("SCPN_BENCH_FORCE_DIST") == "1":  # (the override exercises the RCCL path on one rank)
        from scpn_fusion_core_amd.distributed import EnsembleComm

        try:
            comm = EnsembleComm.from_env({**os.environ, "LOCAL_RANK": str(local_rank)})  # TCP rendezvous of the RCCL id, then ncclCommInitRank on this rank's GPU
        except Exception as exc:  # no communicator (EVERY rank gets the same verdict: EnsembleComm exchanges it): the ranks still
            comm_error = f"{type(exc).__name__}: {exc}"  # solve their shards, the line says the gather did not run and is marked invalid

    from scpn_fusion_core_amd import _lib
    from scpn_fusion_core_amd.cases import member_cfg, shape_scan_member_table
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    if comm is None and world > 1:
        # No communicator.  The documented fall-back (every rank solves its shard, the line is marked invalid) needs every rank to
        # HAVE a device: the ranks say so to each other before anybody starts, and a job in which one rank has none ends at once
        # on every rank instead of leaving the others at a barrier the missing rank never reaches.
        from scpn_fusion_core_amd.distributed import exchange, rendezvous_port

        mine = b"" if _lib.device_count() > local_rank else f"device {local_rank} is not visible ({_lib.device_count()} device(s))".encode()
        seen = exchange(rank, world, mine, addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), port=rendezvous_port())
        missing = [f"rank {r}: {v.decode()}" for r, v in enumerate(seen) if v]
        if missing:
            if rank == 0:
                print(f"bench.py: no communicator ({comm_error}) and not every rank has a GPU -- {'; '.join(missing)}", file=sys.stderr)
            return 3
    _lib.require_gpu()
    ARITH["mode"] = args.arith
    cfg = make_cfg(args.grid)
    from scpn_fusion_core_amd.distributed import plan_ensemble

    try:  # weak: --batch members on every rank; strong (--total-members): contiguous blocks of ceil(total / world), the last may be short
        plan = plan_ensemble(rank, world, members_per_rank=args.batch, total_members=args.total_members)
    except ValueError as exc:
        print(f"bench.py: {exc}", file=sys.stderr)
        return 2
    strong, total, lo, hi = plan["scaling"] == "strong", plan["total"], plan["start"], plan["stop"]
    B, per_rank = plan["members"], plan["block"]  # members resident on THIS rank; the block every rank sends into the gather
    table = shape_scan_member_table(cfg, total)  # config-5 scan: coil currents, I_p and kappa (coil heights) per member
    currents, targets = table["currents"][lo:hi], table["targets"][lo:hi]
    coil_r, coil_z = table["coil_r"][lo:hi], table["coil_z"][lo:hi]
    es = EnsembleSolver(cfg, batch=B, coil_currents=currents, current_targets=targets, coil_r=coil_r, coil_z=coil_z, device=local_rank,
                        arithmetic=args.arith)
    n_pts = args.grid * args.grid

    if comm is None and world > 1:
        from scpn_fusion_core_amd.distributed import HostSync

        host_sync = HostSync.from_env()  # timing barrier over the rendezvous sockets only

    def barrier():
        es.sync()
        if comm is not None:
            comm.wait()     # the gather of the last step
            comm.barrier()  # RCCL all-reduce on the communicator's stream, waited for
        elif host_sync is not None:
            host_sync.barrier()

    rank_t = {"solve_s": 0.0, "gather_wait_s": 0.0}  # this rank's share of the timed region (per_rank table of the line)

    def step():
        t_a = time.perf_counter()
        es.fill_psi(0.0)
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=args.check_every)
        t_b = time.perf_counter()
        if comm is not None:  # ensemble gather of the psi blocks over RCCL/xGMI (the path's only exchange step): launched
            comm.wait()       # asynchronously, it runs while the next step solves; every gather ends inside the timed region
            comm.allgather_psi(es, total=total)
        rank_t["solve_s"] += t_b - t_a
        rank_t["gather_wait_s"] += time.perf_counter() - t_b
        return res

    for _ in range(args.warmup):
        res = step()
    barrier()
    rank_t["solve_s"] = rank_t["gather_wait_s"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    t_last = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_t["final_barrier_s"] = time.perf_counter() - t_last  # the last gather + the wait for the slowest rank
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)
    elif host_sync is not None:
        elapsed = host_sync.allreduce_max(elapsed)

    # ---- per-rank figures of the timed region (every rank sends one record over the rendezvous socket; after the clock) and, for
    #      N > 1, the duration of ONE un-overlapped ensemble gather: a sub-linear curve can then be attributed
    from scpn_fusion_core_amd.distributed import gather_rank_stats, rendezvous_port

    gather_ms = None
    if comm is not None:
        comm.wait()
        comm.barrier()
        tg = time.perf_counter()
        comm.allgather_psi(es, total=total)
        comm.wait()
        gather_ms = comm.allreduce_max(time.perf_counter() - tg) * 1e3
    its_rank = [sum(r["picard_iterations"]) for r in res]
    rank_table = gather_rank_stats(rank, world, {
        "members": B, "solve_ms": round(rank_t["solve_s"] / args.steps * 1e3, 3), "gather_wait_ms": round(rank_t["gather_wait_s"] / args.steps * 1e3, 3),
        "final_barrier_ms": round(rank_t["final_barrier_s"] * 1e3, 3), "picard_iterations_max": max(its_rank), "picard_iterations_min": min(its_rank),
        "converged_members": sum(int(r["converged"]) for r in res)},
        addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), port=rendezvous_port())

    # member 0 of the last timed solve, kept for the full-size parity figure reported with the CPU baseline
    n_cpu = max(1, min(os.cpu_count() or 1, B))  # members the CPU leg solves (one per host core, all host cores if B allows): ALL of them are compared
    psi_gpu_cmp = [es.get_psi(m) for m in range(n_cpu)] if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    picard_gpu = [list(r["picard_iterations"]) for r in res]
    n_conv = sum(int(r["converged"]) for r in res)
    picard_total = [sum(r["picard_iterations"]) for r in res]
    outer = [r["outer_iterations"] for r in res]
    value = args.steps * total / elapsed  # whole-job equilibria per second: every rank's members / the slowest rank's time

    # ---- roofline of the dominant kernel: HIP events on the solver's own stream, in a separate pass (event records would
    #      perturb the timed region); the two stream groups run one after the other while timing, so a kernel is alone on
    #      the chip.  `achieved` = the bytes the FUSED kernel must move once per launch (psi + source in,
    #      psi + coarse rhs + coarse e out -- what the PMC counters confirm, `traffic`) / its mean launch time.  The SURVEY
    #      8(d) per-point figures of the UNFUSED operator sequence (3 sweeps x 24 B + residual 24 + restriction 10 = 106 B/pt)
    #      are kept as `survey_8d_equivalent`: a statement about fusion, not about memory traffic.
    # ---- the OTHER arithmetic mode on the same handle, timed the same way but outside the headline's clock (rank 0, N = 1)
    arith_modes = None
    if rank == 0 and world == 1 and not args.no_arith_modes:
        other = "exact" if args.arith == "contracted" else "contracted"
        psi_head = es.get_psi()
        es.arithmetic = other
        step()
        es.sync()
        to = time.perf_counter()
        for _ in range(args.steps):
            ores = step()
        es.sync()
        dto = (time.perf_counter() - to) / args.steps
        psi_other = es.get_psi()
        es.arithmetic = args.arith
        o_its = [sum(r["picard_iterations"]) for r in ores]
        arith_modes = {
            args.arith: {"value": round(value, 3), "ms_per_step": round(elapsed / args.steps * 1e3, 2), "headline": True,
                         "picard_iterations_per_member": [min(picard_total), max(picard_total)]},
            other: {"value": round(B / dto, 3), "ms_per_step": round(dto * 1e3, 2), "headline": False,
                    "picard_iterations_per_member": [min(o_its), max(o_its)], "converged_members": sum(int(r["converged"]) for r in ores)},
            "psi_rel_linf_between_modes": float(np.abs(psi_other - psi_head).max() / np.abs(psi_head).max()),
            "members_with_equal_iteration_counts": int(sum(a == b for a, b in zip(o_its, picard_total))),
            "what": "exact = every expression in the reference's operation order (bit-exact linear operators); contracted = the same "
                    "iteration with FMA chains and reciprocal multiplies (scpn_gs_set_arith, include/scpn_hip.h); both checked against the CPU "
                    "reference: tests/test_gpu_parity.py / tests/test_gpu_fast_arith.py"}
        del psi_head, psi_other

    roofline = None
    if not args.no_roofline and rank == 0:
        es.fill_psi(0.0)
        es.profile(True)
        es.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=args.check_every)
        fam = es.profile_report()
        es.profile(False)
        roofline = roofline_of(fam, args.grid, B / es.num_groups, es.num_groups) or {"error": "empty profile report"}
        try:
            probes = _lib.bandwidth_probes()
            roofline["measured_bandwidth"] = {**probes, "bytes": 1 << 30,
                                              "note": "copies count read + written bytes; 1 GiB per buffer (4x the Infinity Cache)"}
            roofline["measured_copy_gbs"] = max(v for k, v in probes.items() if k.startswith("copy") or k == "memcpy_dtod")
            roofline["measured_read_gbs"] = probes["read"]
            if "achieved" in roofline:
                roofline["frac_of_measured_copy"] = round(roofline["achieved"] / roofline["measured_copy_gbs"], 4)
        except Exception as exc:  # the probe needs 2 GiB of free HBM; report, do not fail the bench line
            roofline["measured_copy_gbs"] = None
            roofline["copy_probe_error"] = str(exc)
        roofline["iteration"] = iteration_roofline(elapsed / args.steps * 1e3, max(picard_total), B)
        # the same kernels with the chip filled by ONE launch: all members of this GPU in one stream group (the production run
        # reaches this fill by overlapping its two groups).  HIP events again; the rocprofv3 durations of this configuration are
        # profiles/*_isolated_kernel_times.csv (tools/isolated_kernel_times.sh), ~6 us per launch shorter (event pair + dispatch)
        try:
            old_groups = os.environ.get("SCPN_HIP_GROUPS")
            os.environ["SCPN_HIP_GROUPS"] = "1"
            try:
                with EnsembleSolver(cfg, batch=B, coil_currents=currents, current_targets=targets, coil_r=coil_r, coil_z=coil_z, device=local_rank, arithmetic=args.arith) as one:
                    one.fill_psi(0.0)
                    one.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=args.check_every)
                    one.fill_psi(0.0)
                    one.profile(True)
                    one.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=args.check_every)
                    fam1 = one.profile_report()
                    one.profile(False)
            finally:
                if old_groups is None:
                    os.environ.pop("SCPN_HIP_GROUPS", None)
                else:
                    os.environ["SCPN_HIP_GROUPS"] = old_groups
            full = {}
            for name, (key, bpp) in {"smooth_tile_L0_pre": ("smooth_pre", family_table()["smooth_pre"][0]),
                                     "smooth_tile_L0_post": ("smooth_post", family_table()["smooth_post"][0]),
                                     "blend_fused": ("blend_fused", family_table()["blend_fused"][0]),
                                     "source": ("source", family_table()["source"][0])}.items():
                if name in fam1 and fam1[name]["launches"]:
                    us = 1e3 * fam1[name]["ms"] / fam1[name]["launches"]
                    gbs = bpp * n_pts * B / us / 1e3
                    full[name] = {"avg_launch_us": round(us, 2), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                  "frac_of_measured_copy": round(gbs / roofline["measured_copy_gbs"], 4) if roofline.get("measured_copy_gbs") else None}
            roofline["one_group_all_members"] = {"members_per_launch": B, "timing": "HIP events", "kernels": full}
        except Exception as exc:
            roofline["one_group_all_members"] = {"error": str(exc)}

    # ---- CPU baseline: the C oracle (port of the reference algorithm) on the host cores of this GPU's share: one full
    #      free-boundary solve per core, `cores` members solved concurrently (ctypes releases the GIL during the C call)
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from concurrent.futures import ThreadPoolExecutor

        from oracle import oracle as orc

        orc.lib()  # build / load once, outside the timed region
        cores = n_cpu

        def cpu_member(m):
            cm = member_cfg(cfg, currents[m], targets[m], coil_r[m], coil_z[m])
            co = cm["coils"]
            t = time.perf_counter()
            r = orc.free_boundary_solve(cm, [c["r"] for c in co], [c["z"] for c in co], [c["current"] for c in co],
                                        [1.0] * len(co), max_outer=20, tol=1e-4)
            return r, time.perf_counter() - t

        tc = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as pool:
            outs = list(pool.map(cpu_member, range(cores)))
        dt = time.perf_counter() - tc
        ref, dt0 = outs[0]
        cpu = {"value": round(cores / dt, 5), "unit": "equilibria/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
               "sample": f"{cores} full {args.grid}x{args.grid} free-boundary solves (ensemble members 0..{cores - 1}, one per host core, "
                         f"concurrently; member 0: {ref['outer_iterations']} outer, {sum(ref['picard_iterations'])} Picard iterations, "
                         f"{dt0:.1f} s on its core; wall {dt:.1f} s) by oracle/gs_oracle.c",
               "single_core_value": round(1.0 / min(o[1] for o in outs), 5),
               "picard_iterations": ref["picard_iterations"]}
        # second CPU lane: the NumPy restatement (oracle/numpy_lane.py, bit-exact vs the reference fixtures), one core,
        # bounded sample of Picard iterations on the converged state of member 0, extrapolated with that member's iteration count
        try:
            from oracle import numpy_lane as npl

            d = cfg["dimensions"]
            r1 = np.linspace(d["R_min"], d["R_max"], args.grid)
            z1 = np.linspace(d["Z_min"], d["Z_max"], args.grid)
            rr, zz = np.meshgrid(r1, z1)
            state, n_it = np.array(ref["psi"], dtype=np.float64), 8
            tn = time.perf_counter()
            for _ in range(n_it):
                state, _diff = npl.picard_iteration(state, ref["psi"], rr, zz, float(r1[1] - r1[0]), float(z1[1] - z1[0]),
                                                    i_target=float(targets[0]), mu0=cfg["physics"]["vacuum_permeability"],
                                                    z_min_cfg=d["Z_min"], alpha=cfg["solver"]["relaxation_factor"],
                                                    omega=cfg["solver"]["sor_omega"])
            per_it = (time.perf_counter() - tn) / n_it
            cpu["numpy_lane"] = {"seconds_per_picard_iteration": round(per_it, 5), "iterations_timed": n_it, "cores": 1,
                                 "equilibria_per_s_equivalent": round(1.0 / (per_it * sum(ref["picard_iterations"])), 5),
                                 "sample": f"{n_it} Picard iterations (axis/X-point, L-mode source, one V-cycle, blend) at {args.grid}x{args.grid} "
                                           "by oracle/numpy_lane.py, extrapolated to member 0's iteration count"}
        except Exception as exc:  # the baseline line must not be lost to the optional lane
            cpu["numpy_lane"] = {"error": str(exc)}
        # third CPU lane -- north_star's literal comparator: the polyglot Picard + damped-Jacobi lane (core/jax_gs_solver.py:58-167,
        # harness benchmarks/polyglot_gs_solver_comparison.py) at 513 x 513 with the parameters of validation/polyglot/gs_picard_reference.toml:
        # the NumPy restatement on one core (what the harness's Python row runs), the C port on one core and on `cores` cores (one solve
        # each, concurrently).  The device figure is configs.polyglot_picard_jacobi_513; the ratio is formed below once both exist.
        try:
            from benchmarks.configs import POLYGLOT_CASE_513 as pc
            from oracle import numpy_lane as npl

            kw = dict(mu0=pc["mu0"], n_picard=pc["n_picard"], n_jacobi=pc["n_jacobi"], alpha=pc["alpha"], omega_j=pc["omega_j"], beta_mix=pc["beta_mix"])
            pos = (pc["R_min"], pc["R_max"], pc["Z_min"], pc["Z_max"], pc["NR"], pc["NZ"], pc["Ip_target"])
            tp = time.perf_counter()
            psi_np = npl.polyglot_picard_jacobi(*pos, **kw)
            t_np = time.perf_counter() - tp
            tp = time.perf_counter()
            psi_c = orc.gs_solve_np(*pos, pc["mu0"], pc["n_picard"], pc["n_jacobi"], pc["alpha"], pc["omega_j"], pc["beta_mix"])
            t_c1 = time.perf_counter() - tp
            tp = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as pool:
                list(pool.map(lambda _: orc.gs_solve_np(*pos, pc["mu0"], pc["n_picard"], pc["n_jacobi"], pc["alpha"], pc["omega_j"], pc["beta_mix"]), range(cores)))
            t_cn = time.perf_counter() - tp
            cpu["polyglot_513"] = {"n_picard": pc["n_picard"], "n_jacobi": pc["n_jacobi"], "numpy_one_core_ms": round(t_np * 1e3, 2),
                                   "c_port_one_core_ms": round(t_c1 * 1e3, 2), "c_port_all_cores_solves_per_s": round(cores / t_cn, 2), "cores": cores,
                                   "numpy_vs_c_port_psi_rel_linf": float(np.abs(psi_np - psi_c).max() / np.abs(psi_np).max()),
                                   "psi_axis": float(psi_np.max()),
                                   "sample": "one 513x513 solve with the TOML's n_picard x n_jacobi (8 x 16 damped-Jacobi steps); the lane's defaults "
                                             "(80 x 200) cost 125 x as many steps on every lane"}
        except Exception as exc:
            cpu["polyglot_513"] = {"error": f"{type(exc).__name__}: {exc}"}
        if psi_gpu_cmp is not None:  # EVERY member the CPU leg solved against the device's result of the timed run (same inputs, full size)
            errs = [float(np.abs(psi_gpu_cmp[m] - outs[m][0]["psi"]).max() / np.abs(outs[m][0]["psi"]).max()) for m in range(cores)]
            same = [list(outs[m][0]["picard_iterations"]) == picard_gpu[m] for m in range(cores)]
            cpu["psi_rel_linf_vs_gpu"] = max(errs)  # the worst of the compared members
            cpu["members_compared"] = cores
            cpu["psi_rel_linf_vs_gpu_per_member"] = [float(f"{e:.3e}") for e in errs]
            cpu["arithmetic_of_the_compared_gpu_run"] = args.arith
            cpu["picard_iterations_equal_on_all_compared_members"] = bool(all(same))
            cpu["picard_iterations_max_abs_difference"] = int(max(max(abs(int(a) - int(b)) for a, b in zip(outs[m][0]["picard_iterations"], picard_gpu[m]))
                                                                  if len(outs[m][0]["picard_iterations"]) == len(picard_gpu[m]) else 10**6 for m in range(cores)))
            cpu["picard_iterations_gpu_member0"] = picard_gpu[0]
            if not all(same):
                cpu["picard_iterations_mismatch"] = {str(m): {"cpu": list(outs[m][0]["picard_iterations"]), "gpu": picard_gpu[m]}
                                                     for m in range(cores) if not same[m]}

    # ---- beside the Picard figure: the same ensemble with the Newton inner solver of this package (finite-difference Jacobian
    #      action, V-cycle-preconditioned GMRES, full steps).  Not the metric -- the metric is the reference's
    #      Picard iteration -- but the same converged equilibria (psi compared below) per second
    newton = None
    if rank == 0 and world == 1 and not args.no_newton:
        try:
            ncfg = {**cfg, "solver": {**cfg["solver"], "solver_method": "newton", "gmres_preconditioner": "multigrid",
                                      "newton_jacobian": "finite_difference", "newton_damping": 1.0, "max_iterations": 60,
                                      "newton_warmup_steps": 2}}
            psi_picard = es.get_psi()
            es.close()  # (done with the Picard handle; keeps this lane's two group streams on hardware queues of their own)
            with EnsembleSolver(ncfg, batch=B, coil_currents=currents, current_targets=targets, coil_r=coil_r, coil_z=coil_z, device=local_rank, arithmetic=args.arith) as nes:
                def nstep():
                    nes.fill_psi(0.0)
                    return nes.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=args.check_every)
                nres = nstep()
                nes.sync()
                tn = time.perf_counter()
                for _ in range(args.steps):
                    nres = nstep()
                nes.sync()
                dtn = (time.perf_counter() - tn) / args.steps
                psi_newton = nes.get_psi()
            inner = [sum(r["picard_iterations"]) for r in nres]
            newton = {"value": round(B / dtn, 3), "unit": "equilibria/s", "ms_per_step": round(dtn * 1e3, 2),
                      "converged_members": sum(int(r["converged"]) for r in nres), "outer_iterations": [min(o_ for o_ in [r["outer_iterations"] for r in nres]), max(r["outer_iterations"] for r in nres)],
                      "inner_iterations_per_member": [min(inner), max(inner)],
                      "psi_rel_linf_vs_picard": float(np.abs(psi_newton - psi_picard).max() / np.abs(psi_picard).max()),
                      "method": "2 Picard warm-up steps + inexact Newton steps (forcing 1e-3), J v by finite difference of the source map, GMRES(<=8) left-preconditioned by one V-cycle(3,3), full steps; inside the same device free-boundary loop (seed + 50 Jacobi steps at the start of every outer iteration, as the reference)"}
        except Exception as exc:
            newton = {"error": str(exc)}

    # ---- strict-parity gate (outside the timed region), all FIVE thresholds of
    #      validation/benchmark_freegs_public_example_reconstruction.py:491-497:
    #      (1) the harness's own native lane (:275-315: a supplied current density inside a supplied Dirichlet ring) solved on the device
    #          on both public-example grids, compared with the flux map the imported reference function produced (fixture
    #          tests/golden/freegs_native.npz, written by tests/golden/make_golden.py);
    #      (2) the HIP free-boundary solve of the DIII-D-like case at 129^2 against the reference's own NumPy-lane flux map
    #          (tests/golden/d3d129_reference_lane.npz), current closure from the device's J_phi.
    parity = None
    if rank == 0:
        try:
            from scpn_fusion_core_amd.cases import diiid_like_cfg
            from scpn_fusion_core_amd.parity_gate import (GATE_THRESHOLDS, native_profile_source_reconstruction,
                                                          same_case_profile_source_comparison, strict_parity_metrics)

            gated = ("psi_n_rmse", "axis_error_m", "boundary_max_abs_error_wb", "current_closure_relative_error", "xpoint_psi_n_error_max")
            fn = np.load(ROOT / "tests" / "golden" / "freegs_native.npz")
            native = {}
            for cid in ("freegs_01_test_tokamak_freeboundary", "freegs_16_diiid_public_example"):
                spec = json.loads(str(fn[f"{cid}_spec"]))
                ref_plasma = fn[f"{cid}_native_plasma_psi"]                     # the imported reference function's solution
                ext = ref_plasma + fn[f"{cid}_vacuum_psi"]                      # "external" map = the reference lane's own total flux
                dev = native_profile_source_reconstruction(fn[f"{cid}_r"], fn[f"{cid}_z"], fn[f"{cid}_boundary_psi"], fn[f"{cid}_jtor"])
                rec = same_case_profile_source_comparison(fn[f"{cid}_r"], fn[f"{cid}_z"], ext, ref_plasma, fn[f"{cid}_jtor"],
                                                          spec["plasma_current_a"], spec["xpoints"], native_plasma_psi=dev)
                native[cid] = {**{k: rec[k] for k in gated}, "psi_max_abs_error_wb": float(np.abs(dev - fn[f"{cid}_native_plasma_psi"]).max()),
                               "checks": rec["checks"], "pass": bool(rec["pass"] and len(rec["checks"]) == 5)}
            g = np.load(ROOT / "tests" / "golden" / "d3d129_reference_lane.npz")  # data only: the reference lane's flux map
            gcfg = diiid_like_cfg(129)
            with EnsembleSolver(gcfg, batch=2, device=local_rank, arithmetic=args.arith) as ges:
                ges.solve_free_boundary(max_outer_iter=20, tol=1e-4)
                gpsi, gj = ges.get_psi(0), ges.get_jphi(0)
            gd, gax = gcfg["dimensions"], g["axis_x"]
            m = strict_parity_metrics(gpsi, g["psi"], np.linspace(gd["R_min"], gd["R_max"], 129), np.linspace(gd["Z_min"], gd["Z_max"], 129),
                                      x_points=[(gax[2], gax[3])], jtor=gj, plasma_current_a=gcfg["physics"]["plasma_current_target"])
            parity = {"thresholds": dict(GATE_THRESHOLDS),
                      "native_profile_source_lane": {"what": "device solve of _native_profile_source_reconstruction's system (65x65, FreeGS public-example grids) vs the "
                                                             "imported reference function's flux map; synthetic J_tor 0.2 % off I_p by construction", "cases": native},
                      "free_boundary_lane": {"case": "DIII-D-like free-boundary 129x129 (validation/diiid_config.json) vs the reference NumPy lane",
                                             "psi_rmse_wb": m["psi_rmse"], **{k: m[k] for k in gated}, "checks": m["checks"], "pass": m["pass"]},
                      "pass": bool(m["pass"] and all(c["pass"] for c in native.values()) and len(m["checks"]) == 5),
                      "freegs": "live FreeGS absent; the FreeGS 0.8.2 vacuum-psi samples the reference holds are reproduced by the device Green's "
                                "function inside the artifact's own 1e-12 Wb threshold (tests/test_gpu_freegs_gate.py)"}
        except Exception as exc:  # the gate must not cost the bench line
            parity = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- one figure per BASELINE config (outside the timed region)
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        try:
            from benchmarks.configs import measure_configs

            configs = measure_configs(device=local_rank, members=B, check_every=args.check_every, arithmetic=args.arith)
        except Exception as exc:
            configs = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- the OTHER scaling label's N = 1 value (outside the timed region): a weak-scaling run also times the fixed 256-member
    #      ensemble of BASELINE config 5 on this one GPU (what `--total-members 256 --gpus 1` reports as `value`)
    strong_n1 = None
    if rank == 0 and world == 1 and not strong and not args.no_configs:
        try:
            es.close()
            t256 = shape_scan_member_table(cfg, 256)
            with EnsembleSolver(cfg, batch=256, coil_currents=t256["currents"], current_targets=t256["targets"], coil_r=t256["coil_r"],
                                coil_z=t256["coil_z"], device=local_rank, arithmetic=args.arith) as big:
                def bstep():
                    big.fill_psi(0.0)
                    return big.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=args.check_every)
                bstep()
                big.sync()
                tb = time.perf_counter()
                bres = bstep()
                big.sync()
                dtb = time.perf_counter() - tb
            its = [sum(r["picard_iterations"]) for r in bres]
            strong_n1 = {"value": round(256 / dtb, 3), "unit": "equilibria/s", "scaling": "strong", "n_gpus": 1, "total_members": 256,
                         "ms_per_step": round(dtb * 1e3, 2), "converged_members": sum(int(r["converged"]) for r in bres),
                         "picard_iterations_per_member": [min(its), max(its)],
                         "command": "python bench.py --gpus N --total-members 256"}
        except Exception as exc:
            strong_n1 = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        line = {
            "metric": "converged equilibria/s (513x513 free-boundary, Picard tol 1e-8)",
            "value": round(value, 3), "unit": "equilibria/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"fixed ensemble of {total} ITER-like free-boundary equilibria sharded over {world} GPU(s) ({per_rank} per rank), "
                                    if strong else f"ensemble of {B} ITER-like free-boundary equilibria per GPU, ")
                                   + f"{args.grid}x{args.grid}, coil/shape scan (coil currents and I_p +-10 %, kappa in [1.5, 1.9] through the coil heights), "
                                     "PF-coil Green's-function boundary, Picard+V-cycle tol 1e-8 (BASELINE configs[4] = configs[2]'s solve x 256)",
                       "grid": args.grid, "members_per_gpu": per_rank, "members_rank0": B, "global_members": total,
                       "kappa_range_rank0": [round(float(table["kappa"][lo:hi].min()), 4), round(float(table["kappa"][lo:hi].max()), 4)],
                       "arithmetic": args.arith,
                       "parallelism": f"ensemble-sharded x{world}", "rccl": comm.info() if comm is not None else ({"error": comm_error, "gather": "not run"} if comm_error else None),
                       "converged_members_rank0": n_conv, "outer_iterations": [min(outer), max(outer)],
                       "picard_iterations_per_member": [min(picard_total), max(picard_total)]},
        }
        line["per_rank"] = rank_table
        if gather_ms is not None:
            line["gather_ms"] = round(gather_ms, 3)  # one ensemble all-gather on its own (in the timed steps it overlaps the next solve)
        if arith_modes is not None:
            line["arithmetic_modes"] = arith_modes
        if roofline is not None:
            line["roofline"] = roofline
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if parity is not None:
            line["parity"] = parity
        if newton is not None:
            line["newton_fd_mg"] = newton
        if configs is not None:
            line["configs"] = configs
            try:  # north_star's literal comparison: device lane vs the CPU lanes of the same 513x513 polyglot case
                pg, pcb = configs["polyglot_picard_jacobi_513"]["toml_parameters"], cpu["polyglot_513"]
                configs["polyglot_picard_jacobi_513"]["vs_cpu"] = {
                    "hip_solves_per_s": round(1e3 / pg["hip_ms"], 2), "x_numpy_one_core": round(pcb["numpy_one_core_ms"] / pg["hip_ms"], 1),
                    "x_c_port_one_core": round(pcb["c_port_one_core_ms"] / pg["hip_ms"], 1),
                    "x_c_port_all_cores": round((1e3 / pg["hip_ms"]) / pcb["c_port_all_cores_solves_per_s"], 2), "cores": pcb["cores"],
                    "psi_axis_equal": bool(pg["psi_axis"] == pcb["psi_axis"])}
            except (KeyError, TypeError, ZeroDivisionError):
                pass
        if strong_n1 is not None:
            line["strong_scaling_n1"] = strong_n1
        if world > 1 and comm is None:  # the N>1 metric includes the ensemble gather: without it the line is not a scaling point
            line["valid"] = False
            line["n_gpus_gathered"] = 0
        print(json.dumps(line))
    es.close()
    if comm is not None:
        comm.close()
    return 3 if (world > 1 and comm is None) else 0


if __name__ == "__main__":
    sys.exit(main())
