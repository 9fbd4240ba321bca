"""Ordered launch list of ONE steady-state step out of a rocprofv3 --kernel-trace directory:
   python tools/step_sequence.py <trace dir> <marker kernel substring> <out.md>
(the step = everything between the last two launches of the marker kernel, e.g. the optimiser's)."""
import csv
import glob
import sys

src, marker, dst = sys.argv[1], sys.argv[2], sys.argv[3]
trace = glob.glob(f"{src}/**/*_kernel_trace.csv", recursive=True)[0]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(tr) if marker in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1] + 1
win = tr[a:b]
t0 = int(tr[a - 1]["End_Timestamp"])
with open(dst, "w") as f:
    f.write(f"# one replayed step, launch by launch ({len(win)} launches, {(int(win[-1]['End_Timestamp']) - t0) / 1e3:.1f} us)\n\n")
    f.write("| # | start us | kernel | grid | dur us | gap before us |\n|---:|---:|---|---|---:|---:|\n")
    prev = t0
    for i, r in enumerate(win):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']},{r['Grid_Size_Z']}"
        f.write(f"| {i} | {(s - t0) / 1e3:.1f} | {short(r['Kernel_Name'])} | {g} | {(e - s) / 1e3:.1f} | {(s - prev) / 1e3:.1f} |\n")
        prev = e
print("wrote", dst)
