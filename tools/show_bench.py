"""Print the timings of a bench.py JSON line (debug aid):  python tools/show_bench.py FILE"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], {k: round(v["ms_per_step"], 3) for k, v in d.get("other_configs", {}).items()})
