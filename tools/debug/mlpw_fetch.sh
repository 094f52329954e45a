#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the one-launch point MLP against the number of CUs its launch takes (GECCO_MLPW_CUS): does the residual rows'
# second fetch hit L2 when an XCD's in-flight x rows fit it?  (run ON the GPU box: gpurun -- 'bash tools/debug/mlpw_fetch.sh')
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp GECCO_PRECISION=w2 GECCO_FWD_STREAMS=1
for cus in 256 192 160 128 96; do
  export GECCO_MLPW_CUS=$cus
  rm -rf /tmp/mf_$cus
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/mf_$cus --output-format csv -- python3 $R/tools/fwd_once.py 2 > /dev/null 2>&1
  python3 - <<EOF
import csv,glob
f=glob.glob('/tmp/mf_$cus/*/*counter_collection.csv')
rows=[r for r in csv.DictReader(open(f[0])) if 'mlp_fused_w' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
import statistics
v=[float(r['Counter_Value']) for r in rows]
k=glob.glob('/tmp/mf_$cus/*/*kernel_trace.csv')
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(k[0])) if 'mlp_fused_w' in r['Kernel_Name']]
print('CUs $cus: mlp_fused_w FETCH_SIZE x2 = %.1f MB per launch (%d launches), %.1f us' % (2*1024*statistics.mean(v)/1e6, len(v), statistics.mean(d)))
EOF
done
