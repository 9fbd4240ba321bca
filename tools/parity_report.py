"""gpurun_out/parity_errors.jsonl (tests/parity_log.py) -> a markdown table: python tools/parity_report.py <out.md> [note]

One row per (test function, comparison): the worst measurement over the test's parametrisation.  Two kinds of rows:
explicit `parity(name, measured, bound)` calls, and the generic record of every assert_close / allclose call (the
fraction of its tolerance the comparison used; bound / measured = 1 / fraction)."""
import json
import sys
from collections import OrderedDict

rows = OrderedDict()
counts = {}
for line in open("gpurun_out/parity_errors.jsonl"):
    r = json.loads(line)
    test = r.get("test", "")
    fn = test.split("[")[0].replace("tests/", "")
    k = (fn, r["name"])
    counts[k] = counts.get(k, 0) + 1
    if k not in rows:
        rows[k] = r
    else:
        worse = r["measured"] < rows[k]["measured"] if r["higher"] else r["measured"] > rows[k]["measured"]
        if worse:
            rows[k] = r


def fmt_ratio(r):
    m, b = r["measured"], r["bound"]
    if r["higher"]:
        return f"{(1 - b) / (1 - m):.2f} (of 1 - x)" if m < 1 and b < 1 else "-"
    return f"{b / m:.2f}" if m > 0 else "exact"


with open(sys.argv[1], "w") as f:
    f.write("# Measured error of every oracle comparison in the GPU tests\n\n")
    f.write((sys.argv[2] if len(sys.argv) > 2 else "") + "\n\n")
    f.write("`measured` is the worst value over the parametrisations of a test in one `pytest -m gpu` run on an MI355X "
            "(`n` = how many measurements the row summarises); `bound` is what the test asserts (`<=` unless marked "
            "`>=`).  Rows named `assert_close(...)` / `allclose(...)` are the generic record of those calls: `measured` "
            "is the fraction of the stated tolerance the comparison used, so `bound / measured` = 1 / fraction.  "
            "`exact` = the measured difference is 0.\n\n")
    last = None
    for (fn, name), r in rows.items():
        if fn != last:
            f.write(f"\n### {fn}\n\n| comparison | n | measured | bound | bound / measured | note |\n|---|---:|---:|---:|---:|---|\n")
            last = fn
        m, b = r["measured"], r["bound"]
        bs = f">= {b:.6g}" if r["higher"] else f"{b:.3e}"
        ms = f"{m:.6g}" if r["higher"] else f"{m:.3e}"
        f.write(f"| {name} | {counts[(fn, name)]} | {ms} | {bs} | {fmt_ratio(r)} | {r.get('note', '')} |\n")
print("wrote", sys.argv[1], len(rows), "rows")
