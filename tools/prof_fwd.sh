#!/bin/bash
# Kernel trace of a few C2 evaluations of the mixed mode on ONE stream (run ON the GPU box: gpurun -- 'bash tools/prof_fwd.sh r04a').
# Output: gpurun_out/<tag>/fwd_kernel_stats.csv (+ the top kernels on stdout)
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GECCO_PRECISION=${GECCO_PRECISION:-mixed} GECCO_FWD_STREAMS=${GECCO_FWD_STREAMS:-1}
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $R/tools/fwd_once.py 4 > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $F $OUT/fwd_kernel_stats.csv
python3 $R/tools/kstats.py $OUT/fwd_kernel_stats.csv 4 | head -${2:-16}
