#!/bin/bash
# builds the diagnostic variants of the pool-attention probe (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result -I../../gecco_amd/csrc -I../../include"
/opt/rocm/bin/hipcc $F pool_probe.hip -o pool_BASE &
for v in NOLOAD NOSTORE NOMFMA NOSOFTMAX; do /opt/rocm/bin/hipcc $F -DPA_DIAG_$v pool_probe.hip -o pool_$v & done
wait
ls pool_*
