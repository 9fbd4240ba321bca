"""Copy the outputs of tools/final_profiles.sh (+ tools/pmc_bench.sh final_hbm_simclr_r18, the parity log of the GPU test
run) from gpurun_out/ into profiles/r03_*, stamping the commit:   python tools/collect_final.py <n GPU tests passed>"""
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
g, p = root / "gpurun_out", root / "profiles"
h = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
stamp = f"End of round 3, commit {h}, one MI355X (gpurun), tools/final_profiles.sh."
for w in ("default", "dino_vit_tiny", "dino_vit_small", "mae_vit_small_16", "mae_vit_b_32", "knn_allpairs", "knn_allpairs_b64",
          "knn_allpairs_b64_sharded"):
    shutil.copy(g / f"final_bench_{w}.json", p / f"r03_bench_{w}.json")
for src, dst in [(f"final_trace_{w}.md", f"r03_bench_{w}_trace.md") for w in ("simclr_r18", "dino_vit_tiny", "dino_vit_small",
                                                                              "mae_vit_small_16")] + \
        [("final_trace_knn_b64.md", "r03_knn_b64_trace.md")]:
    lines = (g / src).read_text().split("\n")
    (p / dst).write_text("\n".join([lines[0], "", stamp] + lines[1:]))
if (g / "final_hbm_simclr_r18.md").exists():
    shutil.copy(g / "final_hbm_simclr_r18.md", p / "r03_hbm_traffic_simclr_r18.md")
old = subprocess.run(["git", "show", "HEAD:profiles/r03_parity_errors.md"], cwd=root, capture_output=True, text=True).stdout
note = old.split("\n")[2]
note = re.sub(r"commit [0-9a-f]+ \(\d+ passed\)", f"commit {h} ({sys.argv[1]} passed)", note)
subprocess.run([sys.executable, "tools/parity_report.py", "profiles/r03_parity_errors.md", note], cwd=root, check=True)
d = json.load(open(p / "r03_bench_default.json"))
print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], "mfma(model)", d["config"].get("model_mfma_frac"))
print("vit:", d["vit"]["imgs_per_sec"], d["vit"]["ms_per_step"], d["vit"]["roofline"]["frac"])
print("knn:", [(r["dtype"], r["queries"], r["us_per_batch"], r["hbm_frac"], r["pipelined_us_per_batch"], r["pipelined_hbm_frac"])
               for r in d["knn"]["rows"][:1]])
for w in ("dino_vit_tiny", "dino_vit_small", "mae_vit_small_16", "mae_vit_b_32", "knn_allpairs", "knn_allpairs_b64",
          "knn_allpairs_b64_sharded"):
    e = json.load(open(p / f"r03_bench_{w}.json"))
    print(w, e["value"], e["ms_per_step"], (e.get("roofline") or {}).get("frac"))
tot = n = 0
for line in open(p / "r03_hbm_traffic_simclr_r18.md"):
    m = re.match(r"\| (conv_igemm|conv3x3_patch|conv_wgrad|conv_stem_patch)(\S*.*?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m:
        tot += int(m.group(3)) * float(m.group(6))
        n += int(m.group(3))
print("conv launches", n, "mean MB/launch", round(tot / n, 1), open(p / "r03_hbm_traffic_simclr_r18.md").read().split("\n")[6])
