#!/bin/bash
mkdir -p gpurun_out/r02g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/t_all.log 2>&1; tail -1 gpurun_out/r02g/t_all.log
for w in dino_vit_tiny dino_vit_small; do
  timeout -k 10 300 python bench.py --workload $w > gpurun_out/r02g/$w.json 2> gpurun_out/r02g/$w.err || exit 1
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02g/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("final_loss"), d["roofline"]["frac"], d["roofline"].get("achieved"), d["config"].get("model_mfma_frac"))
    print({k:v for k,v in d["roofline"]["by_kernel"].items()})
P
