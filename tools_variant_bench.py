"""Time the resident smoother (3 sweeps / launch) for each libscpn_hip variant under build/variants."""
import glob, os, subprocess, sys, json
if len(sys.argv) > 1:
    os.environ["SCPN_HIP_LIB"] = sys.argv[1]
    sys.path.insert(0, ".")
    import numpy as np
    from scpn_fusion_core_amd.ensemble import EnsembleSolver
    from tests._problems import iter_like_cfg, bench_problem
    n, B = 513, 32
    cfg = iter_like_cfg(n)
    es = EnsembleSolver(cfg, batch=B)
    psi0, src, rr, r, z = bench_problem(n, n)
    es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy()); es.set_source(np.broadcast_to(src, (B, n, n)).copy())
    es.rb_sor(1.3, 30)
    es.profile(True)
    es.rb_sor(1.3, 300)
    rep = es.profile_report()
    es.profile(False)
    k = [v for kk, v in rep.items() if kk.startswith("smooth_tile")][0]
    chk = float(np.abs(es.get_psi(0)).sum())
    print(json.dumps({"lib": os.path.basename(sys.argv[1]), "us_per_launch": round(1e3 * k["ms"] / k["launches"], 2), "checksum": chk}))
else:
    for lib in sorted(glob.glob("build/variants/*.so")):
        subprocess.run([sys.executable, __file__, os.path.abspath(lib)])
