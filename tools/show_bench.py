import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","dtype")})
print("roofline", d["roofline"]["frac"], d["roofline"].get("serial_frac"), d["roofline"]["conv3x3"])
print("hbm", d["roofline_hbm"]["frac"], "step_hbm", d.get("step_hbm"))
print("secondary", {k:v for k,v in d["secondary"].items() if k in ("value","ms_per_step")}, d["secondary"]["roofline"]["frac"], d["secondary"]["roofline"]["conv3x3"])
print("cpu", d["cpu_baseline"])
