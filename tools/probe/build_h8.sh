#!/bin/bash
# builds the diagnostic variants of the h8 / kvq kernel probes (cross-compiles without a GPU)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
for P in h8 kvq; do
/opt/rocm/bin/hipcc $F -DH8_STAMPS ${P}_probe.hip -o ${P}_BASE &
for v in NOMFMA NODMA NOSTORE NOEPI; do /opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_DIAG_$v ${P}_probe.hip -o ${P}_$v & done
done
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_DIAG_NOACT h8_probe.hip -o h8_NOACT &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DKVQ_NS=4 kvq_probe.hip -o kvq_NS4 &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DKVQ_NS=8 kvq_probe.hip -o kvq_NS8 &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DKVQ_NS=4 -DH8_DIAG_NOSTORE kvq_probe.hip -o kvq_NS4_NOSTORE &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DKVQ_NS=8 -DH8_DIAG_NOSTORE kvq_probe.hip -o kvq_NS8_NOSTORE &
wait
ls h8_* kvq_*
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DH8_STAMPS h8areg_probe.hip -o h8areg_BASE &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_DIAG_NOMFMA h8areg_probe.hip -o h8areg_NOMFMA &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_DIAG_NOEPI h8areg_probe.hip -o h8areg_NOEPI &
wait
ls h8areg_*
