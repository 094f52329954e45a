#!/bin/bash
# builds the variants of the one-wave-per-SIMD point-MLP main-loop probe (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
b() { n=$1; shift; /opt/rocm/bin/hipcc $F "$@" mlpw_probe.hip -o mlpw_$n -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Spill|ScratchSize" | sed "s/^/$n: /" | grep -v ": 0 \[" ; }
rm -f mlpw_[A-Z]*
b BASE &
b NOMFMA -DPW_NOMFMA &
b NODMA -DPW_NODMA &
b NOLDS -DPW_NOLDS &
b NOACT -DPW_NOACT &
b SERIAL -DPW_SERIALACT &
b BURST -DPW_DMABURST &
b NOSCHED -DPW_NOSCHED &
b NOPARK -DPW_NOPARK &
b S4 -DPW_SETS=4 -DPW_NS=6 &
b S12 -DPW_SETS=12 -DPW_NS=2 &
b YL -DPW_YL &
wait
ls mlpw_*
