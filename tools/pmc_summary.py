"""Per-kernel means of rocprofv3 --pmc counter CSVs: python tools/pmc_summary.py <counter_collection.csv> [...]."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:28]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(28), " ".join(n.replace("SQ_", "")[:13].rjust(13) for n in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", acc[k].get("SQ_ACTIVE_INST_VALU", [0])))):
    row = [sum(acc[k][n]) / max(1, len(acc[k][n])) if n in acc[k] else float("nan") for n in names]
    print(k.ljust(28), " ".join(f"{v:13.3g}" for v in row))
