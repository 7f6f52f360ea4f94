"""VGPR / AGPR / scratch / occupancy of kernels from a hipcc -S device assembly: isa_regs.py file.s [name-substring]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_smooth_tile"
# the compiler prints a "; Kernel info:" style comment block after each kernel's code
for m in re.finditer(r"^; Function info:|^\s*\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S | re.M):
    if not m.group(1) or pat not in m.group(1):
        continue
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(rf"\.amdhsa_{k} (\S+)", body) or [None, "?"])[1]
    print(f"{name[:100]:100s} vgpr {g('next_free_vgpr'):>4s} accum_off {g('accum_offset'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size')}")
