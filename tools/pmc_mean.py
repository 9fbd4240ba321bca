"""Weighted mean HBM bytes per launch over the kernels of a tools/pmc_traffic.py table whose name matches a regex.
    python tools/pmc_mean.py <table.md> '<regex>'"""
import re
import sys

rx = re.compile(sys.argv[2])
n_tot, b_tot = 0, 0.0
for line in open(sys.argv[1]):
    c = [x.strip() for x in line.split("|")]
    if len(c) < 7 or not c[2].isdigit():
        continue
    if rx.search(c[1]):
        n_tot += int(c[2])
        b_tot += int(c[2]) * float(c[5]) * 1e6
print(f"{b_tot / max(n_tot, 1) / 1e6:.1f} MB/launch over {n_tot} launches")
