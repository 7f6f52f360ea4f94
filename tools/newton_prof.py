import sys, time, json, numpy as np
sys.path.insert(0, '.')
from scpn_fusion_core_amd.ensemble import EnsembleSolver
from scpn_fusion_core_amd.cases import iter_like_cfg
n, B, mi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pre = sys.argv[4] if len(sys.argv) > 4 else "none"
cfg = iter_like_cfg(n, method="newton", omega=1.6, tol=1e-6, max_iter=mi, alpha=0.1)
cfg["solver"]["gmres_preconditioner"] = pre
ip = 15.0 * (1.0 + 0.05 * np.linspace(-1, 1, B))
with EnsembleSolver(cfg, batch=B, current_targets=ip) as es:
    es.profile(True)
    t = time.perf_counter()
    res = es.solve_newton()
    es.sync()
    dt = time.perf_counter() - t
    fam = es.profile_report()
print(f"n={n} B={B} max_iter={mi} wall {dt:.2f}s groups {es.num_groups}")
for r in res[:4]:
    print({k: r[k] for k in ("iterations", "converged", "residual", "gmres_nonconverged_count", "gmres_inner_iterations")})
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:8]:
    print(f"  {k:24s} {v['ms']:10.2f} ms {v['launches']:8d} launches {1e3*v['ms']/v['launches']:8.2f} us/launch")
