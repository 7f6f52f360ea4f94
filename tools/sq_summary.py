#!/usr/bin/env python3
"""Per-kernel summary of the SQ / GRBM counter passes of tools/collect_profiles.sh or tools/pmc_sq_quick.sh.

usage: sq_summary.py <dir with s1/ s2/> [out.csv]
Derived columns (MI355X_MICROARCH.md: SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, GRBM_GUI_ACTIVE is the
sum over the 8 XCDs): clock_GHz = GRBM_GUI_ACTIVE / 8 / duration; valu_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles);
lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles); waves_per_simd = 4 * SQ_WAVE_CYCLES / (1024 x cycles);
wait / stall / issue = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES."""
import csv
import re
import statistics
import sys
from collections import defaultdict
from pathlib import Path


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(gs::k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def load(path: Path):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    meta = {}
    if not path.exists():
        return acc, dur, meta
    seen = set()
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Workgroup_Size"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc, dur, meta


def main():
    src = Path(sys.argv[1])
    a1, d1, m1 = load(src / "s1" / "s1_counter_collection.csv")
    a2, d2, m2 = load(src / "s2" / "s2_counter_collection.csv")
    keys = sorted(d1, key=lambda k: -sum(d1[k]))
    rows = []
    for k in keys:
        if sum(d1[k]) < 0.01 * sum(sum(v) for v in d1.values()):
            continue
        med = lambda a, c: statistics.median(a[k][c]) if c in a.get(k, {}) else float("nan")  # noqa: E731
        us = statistics.median(d1[k]) / 1e3
        cyc = med(a1, "GRBM_GUI_ACTIVE") / 8.0
        wc = med(a1, "SQ_WAVE_CYCLES")
        cyc2 = med(a2, "GRBM_GUI_ACTIVE") / 8.0
        wc2 = med(a2, "SQ_WAIT_ANY") + med(a2, "SQ_WAIT_INST_ANY") + med(a2, "SQ_ACTIVE_INST_ANY")
        rows.append({
            "kernel": k[0], "grid_size": k[1], "vgpr": m1[k][0], "lds_bytes": m1[k][1], "wg": m1[k][2], "launches": len(d1[k]),
            "median_us": round(us, 2), "clock_GHz": round(cyc / (us * 1e3), 3),
            "insts_valu": med(a1, "SQ_INSTS_VALU"), "insts_lds": med(a1, "SQ_INSTS_LDS"), "insts_salu": med(a2, "SQ_INSTS_SALU"),
            "insts_vmem_rd": med(a2, "SQ_INSTS_VMEM_RD"), "insts_vmem_wr": med(a2, "SQ_INSTS_VMEM_WR"), "waves": med(a2, "SQ_WAVES"),
            "valu_busy": round(4.0 * med(a1, "SQ_ACTIVE_INST_VALU") / (1024.0 * cyc), 3),
            "lds_busy": round(med(a2, "SQ_LDS_IDX_ACTIVE") / (256.0 * cyc2), 3),
            "lds_bank_conflict_frac": round(med(a1, "SQ_LDS_BANK_CONFLICT") / max(med(a2, "SQ_LDS_IDX_ACTIVE"), 1.0), 3),
            "waves_per_simd": round(4.0 * wc / (1024.0 * cyc), 2),
            "wait_frac": round(med(a2, "SQ_WAIT_ANY") / wc2, 3), "stall_frac": round(med(a2, "SQ_WAIT_INST_ANY") / wc2, 3),
            "issue_frac": round(med(a2, "SQ_ACTIVE_INST_ANY") / wc2, 3),
            "wait_inst_lds_frac": round(med(a1, "SQ_WAIT_INST_LDS") / wc, 3),
        })
    cols = list(rows[0].keys()) if rows else []
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.DictWriter(out, fieldnames=cols)
    w.writeheader()
    for r in rows:
        w.writerow(r)


if __name__ == "__main__":
    main()
