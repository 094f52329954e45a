#!/bin/bash
# builds the variants of the w2 fused point-MLP probe (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result -DMFW_PROBE"
b() { n=$1; shift; /opt/rocm/bin/hipcc $F "$@" mlpfw_probe.hip -o mlpfw_$n -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs Spill|ScratchSize" | grep -v ": 0 \[" | sed "s/^/$n: /"; }
b BASE -DMFW_STAMPS &
b NOMFMA -DMFW_STAMPS -DMFW_DIAG_NOMFMA &
b NODMA -DMFW_STAMPS -DMFW_DIAG_NODMA &
wait
# the clock under the kernel with / without its LDS fragment reads (profiles/r05b_negative_results.txt item 3)
b NOREAD -DMFW_STAMPS -DMFW_DIAG_NOREAD &
b NOREADDMA -DMFW_STAMPS -DMFW_DIAG_NOREAD -DMFW_DIAG_NODMA &
wait
ls mlpfw_*
