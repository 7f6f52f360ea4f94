import glob, os, subprocess, sys
for lib in sorted(glob.glob("build/variants/*.so")):
    env = dict(os.environ, SCPN_HIP_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True).stdout
    print(os.path.basename(lib), out.strip().splitlines()[-1][75:120])
