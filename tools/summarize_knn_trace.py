"""rocprofv3 kernel trace of tools/knn_pipelined_trace.py -> markdown: per call (256 batches of 64 queries on three streams) the
wall time from the first streaming kernel's start to the last selection kernel's end, per batch, as a fraction of the 8 TB/s
roofline (SURVEY 8d formula (ii): 207.9 MB per batch), and the two kernels' own durations."""
import csv
import glob
import sys

src, dst = sys.argv[1], sys.argv[2]
NB, BYTES = 256, 811457 * 128 * 2 + 64 * 128 * 2 + 64 * 8 * 8
trace = glob.glob(f"{src}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
st = [r for r in rows if "knn_stream" in r["Kernel_Name"]]
se = [r for r in rows if "knn_select" in r["Kernel_Name"]]
calls = len(st) // NB
out = ["# rocprofv3 --kernel-trace: the PIPELINED kNN mode", "",
       "`rocprofv3 --kernel-trace --stats -- python3 tools/knn_pipelined_trace.py`: 811 457 x 128 bf16 bank, 64 queries per batch, 256 batches",
       "per call round-robin on three HIP streams (`wm_knn_topk_many`: every launch of the call queued by ONE C call), 3 calls.", "",
       "| call | first stream start -> last select end (ms) | us per batch | GB/s (207.9 MB per batch) | of 8 TB/s | stream kernel avg us | select kernel avg us | kernel time summed / wall |",
       "|---|---:|---:|---:|---:|---:|---:|---:|"]
for c in range(calls):
    s, e = st[c * NB:(c + 1) * NB], se[c * NB:(c + 1) * NB]
    t0 = min(int(r["Start_Timestamp"]) for r in s)
    t1 = max(int(r["End_Timestamp"]) for r in e)
    wall = (t1 - t0) / 1e3
    ds = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s) / 1e3
    de = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in e) / 1e3
    per = wall / NB
    out.append(f"| {c} | {wall / 1e3:.3f} | {per:.1f} | {BYTES / per / 1e3:.0f} | {BYTES / per / 1e3 / 8000:.3f} | {ds / NB:.1f} | {de / NB:.1f} | {(ds + de) / wall:.2f} |")
out += ["", "Kernel time summed / wall > 1: streaming kernels of consecutive batches overlap each other's tails and the selection kernels run",
        "under the streaming kernels behind them (the 208-MB bank stays largely resident in the 256-MB Infinity Cache from batch to batch,",
        "so the figure is algorithmic bytes over time, not DRAM traffic: DESIGN.md section 9).  The first call includes the cold start."]
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out))
