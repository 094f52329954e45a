#!/bin/bash
# builds a variant of the library with extra compiler flags: tools/probe/build_var.sh <name> <flags...>  -> tools/probe/var/libgecco_<name>.so
# (use with GECCO_HIP_LIB=<path> for A/B runs on the GPU box)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
mkdir -p tools/probe/var/obj_$NAME
SRCS=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.SOURCES))")
for s in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c gecco_amd/csrc/$s -o tools/probe/var/obj_$NAME/${s%.hip}.o &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/probe/var/obj_$NAME/*.o -o tools/probe/var/libgecco_$NAME.so
rm -rf tools/probe/var/obj_$NAME
ls -la tools/probe/var/libgecco_$NAME.so
