"""Condense a rocprofv3 --kernel-trace --stats output directory into profiles/<name>.md
(per-kernel calls, total, average; conv launches also per grid shape)."""
import collections
import csv
import glob
import sys

src, dst, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
note = sys.argv[4] if len(sys.argv) > 4 else ""
# optional: "marker=<kernel substring>,<K>": a kernel launched once per step (the optimiser's); the K steps that END
# with its last K launches are summarised separately -- the replayed hipGraph steps, without warm-up and capture
marker = sys.argv[5] if len(sys.argv) > 5 else ""
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
if marker.startswith("marker="):
    name, k = marker[len("marker="):].rsplit(",", 1)
    k = int(k)
    tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in tr if name in r["Kernel_Name"]]
    if len(ends) > k:
        t0, t1 = ends[-(k + 1)], ends[-1]
        win = [r for r in tr if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
        agg2 = collections.OrderedDict()
        for r in win:
            a = agg2.setdefault(short(r["Kernel_Name"]), [0, 0])
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy = sum(t for _, t in agg2.values())
        with open(dst, "a") as f:
            f.write(f"\n## Steady state: the last {k} replayed steps (window closed by `{name}` launches)\n\n"
                    f"wall {((t1 - t0) / k) / 1e6:.3f} ms/step, kernel time {busy / k / 1e6:.3f} ms/step, "
                    f"{sum(n for n, _ in agg2.values()) / k:.1f} launches/step\n\n"
                    "| kernel | calls/step | ms/step | avg us | % of kernel time |\n|---|---:|---:|---:|---:|\n")
            for kn, (n, t) in sorted(agg2.items(), key=lambda kv: -kv[1][1]):
                f.write(f"| {kn} | {n / k:.1f} | {t / k / 1e6:.3f} | {t / n / 1e3:.1f} | {100.0 * t / busy:.2f} |\n")
print("wrote", dst)
