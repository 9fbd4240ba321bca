"""Copy the outputs of tools/final_profiles_r04.sh (and the parity log of the GPU test run) from gpurun_out/ into profiles/r04_*,
stamp the commit, and refresh profiles/traffic.json (bytes per launch of the dominant kernels + the commit and kernel-source
digest they were measured at):   python tools/collect_final_r04.py <n GPU tests passed>"""
import importlib.util
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
g, p = root / "gpurun_out", root / "profiles"
h = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
stamp = f"End of round 4, commit {h}, one MI355X (gpurun), tools/final_profiles_r04.sh."
for w in ("default", "dino_vit_tiny", "dino_vit_small", "mae_vit_small_16", "mae_vit_b_32", "knn_allpairs", "knn_allpairs_b64"):
    if (g / f"final_bench_{w}.json").exists():
        shutil.copy(g / f"final_bench_{w}.json", p / f"r04_bench_{w}.json")
for src, dst in [(f"final_trace_{w}.md", f"r04_bench_{w}_trace.md") for w in ("simclr_r18", "dino_vit_tiny", "mae_vit_small_16",
                                                                                "simclr_r18_branches", "dino_vit_tiny_branches")] + \
        [("final_trace_knn_b64.md", "r04_knn_b64_trace.md"), ("final_trace_knn_pipelined.md", "r04_knn_pipelined_trace.md")]:
    if (g / src).exists():
        lines = (g / src).read_text().split("\n")
        (p / dst).write_text("\n".join([lines[0], "", stamp] + lines[1:]))
for w in ("simclr_r18", "dino_vit_tiny", "mae_vit_small_16", "knn_b64"):
    if (g / f"final_hbm_{w}.md").exists():
        lines = (g / f"final_hbm_{w}.md").read_text().split("\n")
        (p / f"r04_hbm_traffic_{w}.md").write_text("\n".join([lines[0], "", stamp] + lines[1:]))
note = (f"One `pytest -m gpu` run on an MI355X at commit {h} ({sys.argv[1] if len(sys.argv) > 1 else '?'} passed); "
        "tests/parity_log.py records, tools/parity_report.py formats.")
if (g / "parity_errors.jsonl").exists():
    subprocess.run([sys.executable, "tools/parity_report.py", "profiles/r04_parity_errors.md", note], cwd=root, check=True)


def mean_mb(table: Path, rx: str):
    tot = n = 0
    for line in open(table):
        c = [x.strip() for x in line.split("|")]
        if len(c) < 7 or not c[2].isdigit() or not re.search(rx, c[1]):
            continue
        tot += int(c[2]) * float(c[5])
        n += int(c[2])
    return tot / max(n, 1) * 1e6, n


def per_step(table: Path):
    m = re.search(r"\*\*([\d.]+) GB per step\*\*", table.read_text())
    return m.group(1) if m else "?"


spec = importlib.util.spec_from_file_location("bench", root / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
digest = bench.source_digest()
tj = json.loads((p / "traffic.json").read_text()) if (p / "traffic.json").exists() else {}
CONV = r"^(conv_igemm|conv3x3_patch|conv_wgrad|conv_stem_patch)"
GEMM = r"^(conv_igemm|conv_wgrad|linear_panel|attn_|mlp_fused|ln_linear|ln_mlp)"
for key, rx in (("simclr_r18", CONV), ("dino_vit_tiny", GEMM), ("mae_vit_small_16", GEMM)):
    t = p / f"r04_hbm_traffic_{key}.md"
    if t.exists():
        b, n = mean_mb(t, rx)
        tj[key] = {"bytes_per_launch": round(b, 1), "commit": h, "kernel_digest": digest,
                   "source": f"profiles/r04_hbm_traffic_{key}.md (rocprofv3 --pmc, separate passes; {n} launches; {per_step(t)} GB per step over all kernels)"}
t = p / "r04_hbm_traffic_knn_b64.md"
if t.exists():
    tot = calls = 0
    for line in open(t):
        c = [x.strip() for x in line.split("|")]
        if len(c) >= 7 and c[2].isdigit() and re.search(r"^knn_(stream|select)", c[1]):
            tot += int(c[2]) * float(c[5])
            if c[1].startswith("knn_stream"):
                calls += int(c[2])
    if calls:
        tj["knn_b64"] = {"bytes_per_launch": round(tot / calls * 1e6, 1), "commit": h, "kernel_digest": digest,
                         "source": "profiles/r04_hbm_traffic_knn_b64.md (one wm_knn_topk call, 64 bf16 queries: streaming + selection kernel)"}
(p / "traffic.json").write_text(json.dumps(tj, indent=1) + "\n")
d = json.load(open(p / "r04_bench_default.json"))
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("timing"))
print("vit:", d["vit"]["imgs_per_sec"], d["vit"]["ms_per_step"], d["vit"]["roofline"]["frac"], d["vit"]["roofline"].get("timing"))
print("knn:", [(r["dtype"], r["queries"], r["us_per_batch"], r["hbm_frac"], r["pipelined_us_per_batch"], r["pipelined_hbm_frac"]) for r in d["knn"]["rows"][:1]])
print("traffic:", {k: (round(v["bytes_per_launch"] / 1e6, 1), v["commit"]) for k, v in tj.items()})
