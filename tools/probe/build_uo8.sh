#!/bin/bash
# builds the diagnostic variants of the fused unpool + out_proj (h8) probe (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DUO8_STAMPS uo8_probe.hip -o uo8_BASE &
for v in NOATT NOMFMA NOEPI NORES; do /opt/rocm/bin/hipcc $F -DUO8_STAMPS -DUO8_DIAG_$v uo8_probe.hip -o uo8_$v & done
wait
ls uo8_*
