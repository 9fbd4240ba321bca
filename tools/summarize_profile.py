"""Condense a rocprofv3 --kernel-trace --stats output directory into profiles/<name>.md
(per-kernel calls, total, average; conv launches also per grid shape)."""
import collections
import csv
import glob
import sys

src, dst, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
note = sys.argv[4] if len(sys.argv) > 4 else ""
stats = glob.glob(f"{src}/**/*_kernel_stats.csv", recursive=True)[0]
trace = glob.glob(f"{src}/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\n{note}\n\n")
    f.write(f"profiled steps (incl. warm-up): {steps}; total kernel time {tot / 1e6:.2f} ms = {tot / steps / 1e6:.3f} ms/step\n\n")
    f.write("| kernel | calls/step | ms/step | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:40]:
        f.write(f"| {short(r['Name'])} | {int(r['Calls']) / steps:.1f} | {int(r['TotalDurationNs']) / steps / 1e6:.3f} | "
                f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(trace)):
        if "conv_" not in r["Kernel_Name"] and "knn_" not in r["Kernel_Name"]:
            continue
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"],
               r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    f.write("\n## MFMA kernels by launch shape\n\n| kernel | grid (blocks x,y,z) | vgpr | agpr | lds | calls | avg us |\n|---|---|---:|---:|---:|---:|---:|\n")
    for k, (n, t) in agg.items():
        f.write(f"| {k[0]} | {k[1]},{k[2]},{k[3]} | {k[4]} | {k[5]} | {k[6]} | {n} | {t / n / 1e3:.1f} |\n")
print("wrote", dst)
