#!/bin/bash
# HBM traffic of one kNN call (tools/knn_one.py): two rocprofv3 --pmc passes -> gpurun_out/<name>.md
set -e
name=$1; note=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${name}_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${name}_$c -o p -- python3 tools/knn_one.py "$@" > gpurun_out/${name}_$c.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/${name}_FETCH_SIZE gpurun_out/${name}_WRITE_SIZE gpurun_out/$name.md 0 "$note"
rm -rf gpurun_out/${name}_FETCH_SIZE gpurun_out/${name}_WRITE_SIZE
