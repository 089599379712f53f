"""value / ms_per_step of a bench.py JSON line on stdin (experiments):  python bench.py ... | python tools/bench_value.py [label]"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(*(sys.argv[1:] + [round(d["value"], 1), round(d["ms_per_step"], 4), d.get("parity_checked_slots")]))
