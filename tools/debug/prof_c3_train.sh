# Kernel trace of the image-conditional training step (C3: ConvNeXt-T trained inside the step) under autocast; bash tools/debug/prof_c3_train.sh <tag> [C3|C4]
TAG=${1:-r06u}; CFG=${2:-C3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/train_$CFG --output-format csv -- python3 $R/bench.py --train --amp --config $CFG --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $OUT/train_$CFG.log 2>&1
F=$(find $OUT/train_$CFG -name '*kernel_stats.csv' | head -1)
cp $F $OUT/train_${CFG}_kernel_stats.csv
tail -n 1 $OUT/train_$CFG.log | cut -c1-300
python3 $R/tools/kstats.py $OUT/train_${CFG}_kernel_stats.csv 9 45
