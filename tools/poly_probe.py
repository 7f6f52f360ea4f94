import sys, time
sys.path.insert(0, ".")
from benchmarks.configs import POLYGLOT_CASE_513 as pc
from scpn_fusion_core_amd.polyglot import gs_solve_hip
gs_solve_hip(**pc)
ts = []
for _ in range(6):
    a = time.perf_counter(); gs_solve_hip(**pc); ts.append(time.perf_counter() - a)
print([round(1e3 * x, 2) for x in ts], flush=True)
import ctypes as C
from scpn_fusion_core_amd import _lib
h, m, i = C.c_long(), C.c_long(), C.c_int()
_lib.lib().scpn_operator_pool_stats(C.byref(h), C.byref(m), C.byref(i))
print("pool hits", h.value, "misses", m.value, "idle", i.value)
