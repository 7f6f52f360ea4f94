"""Time the resident smoother (3 sweeps / launch) for each libscpn_hip variant under build/variants."""
import glob, os, subprocess, sys, json
if len(sys.argv) > 1:
    os.environ["SCPN_HIP_LIB"] = sys.argv[1]
    os.environ["SCPN_HIP_GROUPS"] = "1"
    sys.path.insert(0, ".")
    import numpy as np
    from scpn_fusion_core_amd.ensemble import EnsembleSolver
    from scpn_fusion_core_amd.cases import iter_like_cfg, bench_problem
    out = {"lib": os.path.basename(sys.argv[1])}
    for (n, B) in ((513, 32), (513, 16), (257, 16), (129, 16)):
        es = EnsembleSolver(iter_like_cfg(n), batch=B)
        psi0, src, rr, r, z = bench_problem(n, n)
        es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy()); es.set_source(np.broadcast_to(src, (B, n, n)).copy())
        es.rb_sor(1.3, 30)
        es.profile(True)
        es.rb_sor(1.3, 300)
        rep = es.profile_report(); es.profile(False)
        k = [v for kk, v in rep.items() if kk.startswith("smooth_tile")][0]
        out[f"{n}/B{B}"] = round(1e3 * k["ms"] / k["launches"], 2)
        out["chk"] = float(np.abs(es.get_psi(0)).sum())
        es.close()
    print(json.dumps(out))
else:
    for lib in sorted(glob.glob("build/variants/*.so")):
        subprocess.run([sys.executable, __file__, os.path.abspath(lib)])
