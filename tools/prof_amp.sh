#!/bin/bash
# Kernel trace of the training step under autocast(float16) + GradScaler (run ON the GPU box: gpurun -- 'bash tools/prof_amp.sh r04h')
TAG=${1:-r04h}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AMP_SKIP_GRADS=1 AMP_MODES=${AMP_MODES:-amp}
rocprofv3 --kernel-trace --stats -d $OUT/amp --output-format csv -- python3 $R/tools/debug/amp_train.py 48 6 > $OUT/amp.log 2>&1
F=$(find $OUT/amp -name '*kernel_stats.csv' | head -1)
cp $F $OUT/amp_kernel_stats.csv
grep "ms per step" $OUT/amp.log
python3 $R/tools/kstats.py $OUT/amp_kernel_stats.csv 9 ${2:-40}
