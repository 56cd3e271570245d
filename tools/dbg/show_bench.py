import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.pop("kernels")
print(json.dumps(d, indent=1)[:6000])
for r in k:
    print(r["class"], r["kernel"][:70], r["launches_per_step"], r["ms_per_step"], r.get("frac"))
