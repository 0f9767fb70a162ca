#!/bin/bash
# Diagnostic build with an experiment macro (never shipped): tools/_build/libhgn_mp_$1.so   usage: exp_build.sh NAME -DMACRO...
set -e
NAME=$1; shift
cd "$(dirname "$0")/../hyper-graph-nets_amd/csrc"
mkdir -p ../../tools/_build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $@"
for f in mlp mlp6 segment wgrad features; do /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/${f}_$NAME.o & done
/opt/rocm/bin/hipcc $F -x hip -c host.cpp -o /tmp/host_$NAME.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_build/libhgn_mp_$NAME.so /tmp/mlp_$NAME.o /tmp/mlp6_$NAME.o /tmp/segment_$NAME.o /tmp/wgrad_$NAME.o /tmp/features_$NAME.o /tmp/host_$NAME.o
