"""Lab: the fields of a bench.py JSON line that matter at a glance.  usage: show_bench.py file"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["config"]["stage_ms"], "roofline", d["roofline"]["frac"], d["roofline_trailing_update"]["frac"])
for k in ("extra", "extra_s", "extra_n65536"):
    e = d.get(k, {})
    print(k, e.get("seconds"), e.get("seconds_each"), e.get("stage_ms"), e.get("roofline", {}).get("frac"),
          e.get("roofline_trailing_update", {}).get("frac"), e.get("error"))
c = d.get("cpu_baseline", {})
print("cpu", c.get("value"), c.get("n"), c.get("cores"), str(c.get("sample"))[-260:])
