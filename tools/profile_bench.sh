#!/bin/bash
# rocprofv3 kernel trace of a bench.py workload on the GPU box, summarised into gpurun_out/<name>.md
#   tools/profile_bench.sh <name> <steps> <total profiled steps> [bench args...]
# WM_PROFILE_MARKER=<kernel launched once per step> adds a steady-state section over the last <steps> replayed steps
# (give --no-kernel-timer so that no eager steps follow the timed ones)
set -e
name=$1; steps=$2; total=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf gpurun_out/$name
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -o p -- python3 bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-secondary "$@" > gpurun_out/$name.log 2>&1
python3 tools/summarize_profile.py gpurun_out/$name gpurun_out/$name.md $total "python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-secondary $* under rocprofv3 --kernel-trace --stats" ${WM_PROFILE_MARKER:+marker=$WM_PROFILE_MARKER,$steps}
rm -rf gpurun_out/$name
