#!/bin/bash
# HBM traffic per kernel of a bench.py workload: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), eager steps
#   tools/pmc_bench.sh <name> <steps in the pass> "<note>" [bench args...]   -> gpurun_out/<name>.md
set -e
name=$1; nsteps=$2; note=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${name}_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${name}_$c -o p -- python3 bench.py --no-graph --no-cpu-baseline --no-secondary --no-kernel-timer "$@" > gpurun_out/${name}_$c.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/${name}_FETCH_SIZE gpurun_out/${name}_WRITE_SIZE gpurun_out/$name.md $nsteps "$note"
rm -rf gpurun_out/${name}_FETCH_SIZE gpurun_out/${name}_WRITE_SIZE
