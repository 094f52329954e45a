#!/bin/bash
# builds the diagnostic variants of the weight-gradient kernel probe (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result -I../../gecco_amd/csrc -I../../include"
/opt/rocm/bin/hipcc $F tn_probe.hip ../../gecco_amd/csrc/common_xcd.o -o tn_BASE 2>/dev/null || /opt/rocm/bin/hipcc $F tn_probe.hip -o tn_BASE &
for v in NOLOAD NOSTORE NOREAD NOMFMA; do /opt/rocm/bin/hipcc $F -DTN_DIAG_$v tn_probe.hip -o tn_$v & done
wait
ls tn_*
