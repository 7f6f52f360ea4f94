import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], "eq/s", d["ms_per_step"], "ms/step")
f = d["roofline"]["families_ms"]
tot = sum(v["ms"] for v in f.values())
for k, v in sorted(f.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:28s} {v['ms']:9.2f} ms {v['launches']:6d}  {1e3*v['ms']/v['launches']:8.1f} us/launch {100*v['ms']/tot:5.1f}%")
print("total", round(tot, 1))
