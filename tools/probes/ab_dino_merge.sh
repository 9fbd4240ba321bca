#!/bin/bash
# Same-box A/B of the DINO step with the multi-crop student merged (WM_DINO_MERGE=1, default) or run once per
# resolution as dino does (=0): bench.py --workload dino_vit_{tiny,small}, two alternations.
for rep in 1 2; do for v in 0 1; do for w in dino_vit_tiny dino_vit_small; do
WM_DINO_MERGE=$v timeout -k 10 300 python bench.py --workload $w --steps 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge=$v rep $rep $w', d['ms_per_step'], d['final_loss'])" || exit 1
done; done; done
