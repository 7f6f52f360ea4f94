"""s_memrealtime stamps of eight sampled tile workgroups (load / six colour passes / store / defect / restriction).
usage: [STAMPS_MODE=plain|defect] [STAMPS_B=members] [STAMPS_DEFINES="A=1 B=2"] [SCPN_HIP_TILEGEO=24|32|48] python tools/tile_stamps.py     (builds the -DCV_STAMPS experiment library on first use)"""
import os, sys, ctypes as C
sys.path.insert(0, ".")
from scpn_fusion_core_amd import _lib
_mode = os.environ.get("STAMPS_MODE", "plain")
_extra = tuple(x for x in os.environ.get("STAMPS_DEFINES", "").split() if x)  # e.g. "TILE_MID_LZ=40 TILE_MID_NW=10"
_tag = "".join(c for c in "_".join(_extra) if c.isalnum() or c == "_")
os.environ["SCPN_HIP_LIB"] = str(_lib.build(variant="stamps_" + _mode + ("_" + _tag if _tag else ""),
                                            defines=("CV_STAMPS",) + (("STAMP_DEFECT=true",) if _mode == "defect" else ()) + _extra))
os.environ["SCPN_HIP_GROUPS"] = "1"
import numpy as np
from scpn_fusion_core_amd.ensemble import EnsembleSolver
from scpn_fusion_core_amd.cases import iter_like_cfg, bench_problem
for n, B in ((513, int(os.environ.get("STAMPS_B", "32"))),):
    es = EnsembleSolver(iter_like_cfg(n), batch=B)
    psi0, src, rr, r, z = bench_problem(n, n)
    es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy()); es.set_source(np.broadcast_to(src, (B, n, n)).copy())
    mode = os.environ.get("STAMPS_MODE", "plain")
    import time
    if mode == "picard":  # the Picard iteration itself (the fused tail of the level-0 post-smoother only exists inside it): fixed iteration count
        es.close()
        es = EnsembleSolver(iter_like_cfg(n, tol=1e-300, max_iter=20), batch=B)
        es.solve_equilibrium()
        es.sync(); t0 = time.perf_counter(); es.solve_equilibrium(); es.sync()
        print(f"{(time.perf_counter() - t0) / 20 * 1e6:.1f} us per Picard iteration (stamped build)")
    for _ in range(0 if mode == "picard" else 5):
        if mode == "plain": es.rb_sor(1.3, 3)
        else: es.vcycle(1.0, 3, 3, 5)
    es.sync(); t0 = time.perf_counter()
    for _ in range(0 if mode == "picard" else 20):
        if mode == "plain": es.rb_sor(1.3, 3)
        else: es.vcycle(1.0, 3, 3, 5)
    es.sync()
    if mode != "picard": print(f"{(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call (stamped build)")
    L = C.CDLL(os.environ["SCPN_HIP_LIB"])
    buf = (C.c_ulonglong * 128)()
    L.scpn_debug_tile_stamps(buf)
    names = ["start", "loaded"] + [f"pass {i}" for i in range(6)] + ["stored", "defect", "end", "drain"]
    idx = (0, 1, 11, 2, 3, 4, 5, 6, 7, 10) if os.environ.get("STAMPS_DRAIN") else ((0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10) if mode in ("defect", "picard") else (0, 1, 2, 3, 4, 5, 6, 7, 8, 10))
    print(f"--- n={n} B={B}: per-phase us (rows kernel: successive tiles of workgroup 1; tile kernel: 8 sampled workgroups); last column = start offset")
    rows = []
    for g in range(8):
        st = list(buf)[g * 16:(g + 1) * 16]
        if st[0] == 0:
            continue
        d = [(st[idx[j]] - st[idx[j - 1]]) * 0.01 for j in range(1, len(idx))]
        rows.append((g, d, (st[10] - st[0]) * 0.01, st[0]))
    t0 = min(r[3] for r in rows)
    print("wg   " + " ".join(f"{names[i]:>7s}" for i in idx[1:]) + "   total   start")
    for g, d, tot, s0 in rows:
        print(f"{g:2d}   " + " ".join(f"{x:7.2f}" for x in d) + f" {tot:7.2f} {(s0 - t0) * 0.01:7.2f}")
    es.close()
