import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["workload"][:40], "frac %.4f kernel_ms %.4f ms_per_step %.4f" % (d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["ms_per_step"]))
