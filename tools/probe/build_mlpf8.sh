#!/bin/bash
# builds the diagnostic variants of the fused h8 MLP probe (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DMF8_STAMPS mlpf8_probe.hip -o mlpf8_BASE &
for v in NOP1 NOP2; do /opt/rocm/bin/hipcc $F -DMF8_STAMPS -DMF8_DIAG_$v mlpf8_probe.hip -o mlpf8_$v & done
wait
ls mlpf8_*
