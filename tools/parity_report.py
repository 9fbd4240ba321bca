"""gpurun_out/parity_errors.jsonl (tests/parity_log.py) -> a markdown table: python tools/parity_report.py <out.md> [note]"""
import json
import sys

rows = {}
for line in open("gpurun_out/parity_errors.jsonl"):
    r = json.loads(line)
    k = r["name"]
    if k not in rows:
        rows[k] = r
    else:  # several measurements under one name (parametrised tests, seeds): keep the worst
        worse = r["measured"] < rows[k]["measured"] if r["higher"] else r["measured"] > rows[k]["measured"]
        if worse:
            rows[k] = r
with open(sys.argv[1], "w") as f:
    f.write("# Measured error of every oracle comparison in the GPU tests\n\n")
    f.write((sys.argv[2] if len(sys.argv) > 2 else "") + "\n\n")
    f.write("`measured` is the worst value over the parametrisations of a test in one `pytest -m gpu` run on an MI355X; "
            "`bound` is what the test asserts (direction: `<=` unless marked `>=`).\n\n")
    f.write("| comparison | measured | bound | bound / measured | note |\n|---|---:|---:|---:|---|\n")
    for k, r in rows.items():
        m, b = r["measured"], r["bound"]
        if r["higher"]:
            ratio = (1 - b) / (1 - m) if m < 1 and b < 1 else float("nan")
            f.write(f"| {k} | {m:.6g} | >= {b:.6g} | {ratio:.2f} (of 1 - x) | {r['note']} |\n")
        else:
            ratio = b / m if m > 0 else float("inf")
            f.write(f"| {k} | {m:.3e} | {b:.3e} | {ratio:.2f} | {r['note']} |\n")
print("wrote", sys.argv[1], len(rows), "rows")
