#!/bin/bash
# Kernel trace of a few C2 evaluations of the headline mode (GECCO_PRECISION, default w2) on ONE stream (run ON the GPU box:
# gpurun -- 'bash tools/prof_fwd.sh r05a').  The tree's commit hash travels in TREE_COMMIT (written before gpurun: no .git on the box).
# Output: gpurun_out/<tag>/fwd_kernel_stats.csv (+ the top kernels on stdout)
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GECCO_PRECISION=${GECCO_PRECISION:-w2} GECCO_FWD_STREAMS=${GECCO_FWD_STREAMS:-1}
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $R/tools/fwd_once.py 4 > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
{ echo "# tree $(cat $R/TREE_COMMIT 2>/dev/null || echo unknown), GECCO_PRECISION=$GECCO_PRECISION GECCO_FWD_STREAMS=$GECCO_FWD_STREAMS, 4 evaluations: rocprofv3 --kernel-trace --stats -- python3 tools/fwd_once.py 4"; cat $F; } > $OUT/fwd_kernel_stats.csv
python3 $R/tools/kstats.py $OUT/fwd_kernel_stats.csv 4 | head -${2:-16}
