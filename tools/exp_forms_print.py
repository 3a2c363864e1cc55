import json, sys
tag = sys.argv[1]
d = json.load(open(f"/tmp/st_{tag}.json"))
print(tag, d["param_sha256"][:12], "grad_norm", d["grad_norm"], "param_sum", d["param_sum"])
