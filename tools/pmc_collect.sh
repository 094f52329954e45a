#!/bin/bash
# Counter passes over whole evaluations of the headline mode (GECCO_PRECISION, default w2; run ON the GPU box:
# gpurun -- 'bash tools/pmc_collect.sh r05h').  The tree's commit hash travels in TREE_COMMIT (written before gpurun: no .git on the box).
# Separate --pmc passes (TCC counters do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots"), the program directly
# after `--`.  Output: gpurun_out/<tag>/pmc/{FETCH_SIZE,WRITE_SIZE,MFMA}/... and gpurun_out/<tag>/forward_pmc_summary.txt
TAG=${1:-r03h}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT/pmc
cd /tmp && export TMPDIR=/tmp
export GECCO_PRECISION=${GECCO_PRECISION:-w2} GECCO_FWD_STREAMS=1
export GECCO_TREE=$(cat $R/TREE_COMMIT 2>/dev/null || echo unknown)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc/FETCH_SIZE --output-format csv -- python3 $R/tools/fwd_once.py 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc/WRITE_SIZE --output-format csv -- python3 $R/tools/fwd_once.py 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc/MFMA --output-format csv -- python3 $R/tools/fwd_once.py 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc/LDS --output-format csv -- python3 $R/tools/fwd_once.py 2 > /dev/null 2>&1
python3 $R/tools/pmc_to_json.py $OUT/pmc 2 $TAG ${PMC_WRITE:+--write} > $OUT/forward_pmc_summary.txt
[ -n "$PMC_WRITE" ] && cp $R/profiles/gemm_hbm_traffic.json $OUT/gemm_hbm_traffic.json
cat $OUT/forward_pmc_summary.txt | head -40
