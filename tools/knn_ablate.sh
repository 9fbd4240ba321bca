#!/bin/bash
# kNN stream kernel timing matrix on the GPU box: WM_KNN_NT x WM_KNN_DEBUG (timing-only ablation bits)
# usage: tools/knn_ablate.sh <tag> "<nt> <dbg>" ...   -> gpurun_out/<tag>_nt<nt>_d<dbg>.md
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  set -- $v
  export WM_KNN_NT=$1 WM_KNN_DEBUG=$2
  out=$root/gpurun_out/${tag}_nt$1_d$2
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python $root/tools/knn_one.py ${BQ:-64} bf16 > /dev/null 2>&1 || exit 1
  python $root/tools/summarize_profile.py $out $out.md 5 "nt=$1 dbg=$2" > /dev/null
  echo "nt=$1 dbg=$2: $(grep -E 'knn_s' $out.md | head -2 | awk -F'|' '{print $2, $5}' | tr '\n' ';')"
done
