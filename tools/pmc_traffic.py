"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.md> [steps in the pass] [note]

Corrections from MI355X_MICROARCH.md (HBM section): the counters are reported in KiB; on gfx950
FETCH_SIZE shows half of the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores and float atomics.
"""
import collections
import csv
import glob
import sys


def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


fd, wd, out = sys.argv[1:4]
steps_arg = sys.argv[4] if len(sys.argv) > 4 else "0"   # a number, or the name of a kernel launched once per step
note = sys.argv[5] if len(sys.argv) > 5 else ""
fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
steps = int(steps_arg) if steps_arg.isdigit() else max([n for k, (n, _) in fetch.items() if steps_arg in k] or [0])
rows = []
for k in fetch:
    n, fkb = fetch[k]
    wkb = write.get(k, [0, 0.0])[1]
    rows.append((k, n, 2 * fkb * 1024 / n, wkb * 1024 / max(write.get(k, [1])[0], 1)))
rows.sort(key=lambda r: -r[1] * (r[2] + r[3]))
with open(out, "w") as f:
    f.write("# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n")
    if note:
        f.write(note + "\n\n")
    f.write("FETCH_SIZE x 2 (gfx950 correction), both counters KiB -> bytes.\n\n")
    if steps:
        tot_r = sum(n * rb for _, n, rb, _ in rows) / steps
        tot_w = sum(n * wb for _, n, _, wb in rows) / steps
        f.write(f"Sum over all kernels: {tot_r / 1e9:.2f} GB read + {tot_w / 1e9:.2f} GB written = "
                f"**{(tot_r + tot_w) / 1e9:.2f} GB per step** ({steps} steps in the pass).\n\n")
    f.write("| kernel | launches | read MB/launch | write MB/launch | total MB/launch | GB/step |\n|---|---:|---:|---:|---:|---:|\n")
    for k, n, rb, wb in rows[:45]:
        per = f"{n * (rb + wb) / steps / 1e9:.3f}" if steps else "-"
        f.write(f"| {k} | {n} | {rb / 1e6:.2f} | {wb / 1e6:.2f} | {(rb + wb) / 1e6:.2f} | {per} |\n")
print("wrote", out)
