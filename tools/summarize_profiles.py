#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the committed summaries:

  profiles/<tag>_kernel_stats_bench_default.csv      rocprofv3 --kernel-trace --stats of `python bench.py --steps 3 --warmup 1`
  profiles/<tag>_bench_under_rocprof.json            the bench line printed by that same run
  profiles/<tag>_pmc_fetch_write_by_kernel.csv       FETCH_SIZE / WRITE_SIZE per kernel and launch shape (separate PMC passes)
  profiles/pmc_traffic.json                          HBM bytes per member-launch that bench.py reports as roofline.traffic

HBM bytes = FETCH_SIZE[KB] * 1024 * 2 + WRITE_SIZE[KB] * 1024: the x2 is the gfx950 correction of MI355X_MICROARCH.md (HBM /
rocprofv3 section) for kernels whose loads are 16 B per lane (all kernels listed in pmc_traffic.json load double2).
"""
import csv
import json
import re
import statistics
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r02b"
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 513
members = int(sys.argv[3]) if len(sys.argv) > 3 else 32  # PMC passes run with SCPN_HIP_GROUPS=1: a launch covers all members
src = ROOT / "gpurun_out" / f"prof_{tag}"
out = ROOT / "profiles"


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(gs::k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


# ---- kernel stats
rows = list(csv.DictReader(open(src / "k" / "k_kernel_stats.csv")))
with open(out / f"{tag}_kernel_stats_bench_default.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "average_us", "percentage", "min_us", "max_us"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 3),
                    r["Percentage"], round(float(r["MinNs"]) / 1e3, 2), round(float(r["MaxNs"]) / 1e3, 2)])
for line in open(src / "bench_under_rocprof.log"):
    if line.startswith('{"metric'):
        (out / f"{tag}_bench_under_rocprof.json").write_text(line)

# ---- PMC
acc = defaultdict(lambda: defaultdict(list))  # (kernel, grid_size) -> counter -> values
for sub, fname in (("f", "f_counter_collection.csv"), ("w", "w_counter_collection.csv")):
    for r in csv.DictReader(open(src / sub / fname)):
        acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out / f"{tag}_pmc_fetch_write_by_kernel.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "grid_size", "dispatches", "median_KB", "max_KB", "sum_KB"])
    for (k, g), cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
        for c, vals in cs.items():
            w.writerow([c, k, g, len(vals), round(statistics.median(vals), 1), round(max(vals), 1), round(sum(vals), 1)])


def tile_grid(level: int, halo: int, lz: int = 48, nw: int = 8) -> int:  # (the geometry of groups of more than 14 members)
    n = (grid - 1) // 2**level + 1
    gx, gy = -(-n // (128 - 2 * halo)), -(-n // (lz - 2 * halo))
    return gx * gy * members * 64 * nw


def hbm_bytes(kernel_sub: str, gsize: int | None):
    for (k, g), cs in sorted(acc.items(), key=lambda kv: -kv[0][1]):  # largest launch shape first: the bench ensemble
        hit = (k == kernel_sub[:-1]) if kernel_sub.endswith("$") else (kernel_sub in k)
        if hit and (gsize is None or g == gsize) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            return (statistics.median(cs["FETCH_SIZE"]) * 2.0 + statistics.median(cs["WRITE_SIZE"])) * 1024.0 / members
    return None


traffic = {}
for lvl in (0, 1, 2):
    v = hbm_bytes("k_smooth_tile<3, false, true", tile_grid(lvl, 8))
    if v:
        traffic[f"smooth_tile_L{lvl}_pre"] = round(v, 1)
    v = hbm_bytes("k_smooth_tile<3, true, false", tile_grid(lvl, 6))
    if v:
        traffic[f"smooth_tile_L{lvl}_post"] = round(v, 1)
for fam, sub in (("blend_fused", "k_blend_tile"), ("vcycle_coarse", "k_vcycle_coarse"), ("jsum", "gs::k_jsum<"), ("picard_final", "k_picard_final"),
                 ("jsum_final", "k_jsum_final")):
    v = hbm_bytes(sub, None)
    if v:
        traffic[fam] = round(v, 1)
# ---- SQ / GRBM counters (tools/sq_summary.py)
import subprocess  # noqa: E402

if (src / "s1" / "s1_counter_collection.csv").exists():
    subprocess.run([sys.executable, str(ROOT / "tools" / "sq_summary.py"), str(src), str(out / f"{tag}_sq_by_kernel.csv")], check=True)

(out / "pmc_traffic.json").write_text(json.dumps({
    "source": f"profiles/{tag}_pmc_fetch_write_by_kernel.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
              f"SCPN_HIP_GROUPS=1, {members} members per launch)",
    "correction": "FETCH_SIZE x2 for 16-B/lane coalesced loads on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "grid": grid, "tag": tag, "hbm_bytes_per_member_launch": traffic}, indent=1))
print(json.dumps(traffic, indent=1))
