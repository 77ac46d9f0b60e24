#!/bin/bash
# kernel trace of a few cold solves of configs[2] at a given batch: bash tools/trace_solve.sh BATCH   (prints the per-kernel averages)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lt$1
mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/trace_solve.py $1 > $O/out.log 2>&1 || { tail -5 $O/out.log; exit 1; }
F=$(find $O -name "*kernel_stats.csv" | head -1)
cut -d, -f1-6 $F | head -14 | cut -c1-200
