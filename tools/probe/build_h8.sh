#!/bin/bash
# builds the diagnostic variants of the h8 kernel probe (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DH8_STAMPS h8_probe.hip -o h8_BASE &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_PRIO h8_probe.hip -o h8_PRIO &
for v in NOMFMA NODMA NOACT NOSTORE NOEPI; do /opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_DIAG_$v h8_probe.hip -o h8_$v & done
wait
ls -la h8_*
