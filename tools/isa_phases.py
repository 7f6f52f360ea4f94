#!/usr/bin/env python3
"""Static instruction mix of one kernel of the device assembly, split at s_barrier: VALU / DS / VMEM / SALU per segment.
usage: isa_phases.py <scpn.s> <kernel label substring> [--loops]
(make the assembly with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++20 -S --cuda-device-only -o scpn.s csrc/scpn_hip.hip)"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(": ; @" + l.split(":")[0] for _ in [0])) or (l.startswith("_Z") and key in l and ":" in l))
seg, segs, labels = dict(valu=0, ds=0, vmem=0, salu=0, fp64=0, branch=0), [], []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end") or t.startswith("s_endpgm"):
        break
    if not t or t.startswith((";", ".")):
        if re.match(r"\.LBB\d+_\d+:", t):
            labels.append((len(segs), t))
        continue
    op = t.split()[0]
    if op == "s_barrier":
        segs.append(seg)
        seg = dict(valu=0, ds=0, vmem=0, salu=0, fp64=0, branch=0)
        continue
    if op.startswith("v_"):
        seg["valu"] += 1
        if "_f64" in op:
            seg["fp64"] += 1
    elif op.startswith("ds_"):
        seg["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        seg["vmem"] += 1
    elif op.startswith(("s_cbranch", "s_branch")):
        seg["branch"] += 1
    elif op.startswith("s_"):
        seg["salu"] += 1
segs.append(seg)
tot = {k: sum(s[k] for s in segs) for k in seg}
print(f"{len(segs)} segments; totals {tot}")
for i, s in enumerate(segs):
    print(f"  seg {i:2d}: valu {s['valu']:5d} (fp64 {s['fp64']:4d})  ds {s['ds']:4d}  vmem {s['vmem']:4d}  salu {s['salu']:4d}  branches {s['branch']:3d}")
