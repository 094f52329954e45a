#!/bin/bash
# Builds the fp6 probes (cross-compiles without a GPU):
#   fp6_rate        matrix-pipe rate of the scaled f8f6f4 MFMA by operand format beside the fp16 MFMA (random / zero operands)
#   fp6_mfma_probe  semantics of v_cvt_scalef32_pk32_fp6_* and of the per-lane E8M0 block scales of the MFMA (exact products)
#   h8_FP6 / h8_FP4 the shipped mlp.0 kernel with only the cross terms' instruction format swapped (timing only, garbage values)
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F fp6_rate.hip -o fp6_rate &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -DB_VIA_F16 fp6_mfma_probe.hip -o fp6_mfma_probe &
/opt/rocm/bin/hipcc $F -DH8_STAMPS h8_probe.hip -o h8_BASE &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_DIAG_CROSSFMT=2 h8_probe.hip -o h8_FP6 &
/opt/rocm/bin/hipcc $F -DH8_STAMPS -DH8_DIAG_CROSSFMT=4 h8_probe.hip -o h8_FP4 &
wait
ls -la fp6_rate fp6_mfma_probe h8_BASE h8_FP6 h8_FP4
