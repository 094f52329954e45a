#!/bin/bash
# Builds the inducer-chain probe with phase stamps: tools/probe/chain_probe <B> <two-term> <cluster>
R=$(cd "$(dirname "$0")/../.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHAIN_STAMPS -I $R/gecco_amd/csrc $R/tools/probe/chain_probe.hip -o $R/tools/probe/chain_probe
