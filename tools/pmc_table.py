"""Pivot a rocprofv3 --pmc counter_collection.csv: mean counter value per launch for each kernel.

    python tools/pmc_table.py <dir> [kernel-substring]
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    if sub not in name:
        continue
    a = agg[name][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
    a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, cs in agg.items():
    any_c = next(iter(cs.values()))
    print(f"{k}: launches {any_c[0]}, avg {any_c[2] / any_c[0] / 1e3:.1f} us")
    for c, (n, v, _) in sorted(cs.items()):
        print(f"    {c:28s} {v / n:16.0f}")
