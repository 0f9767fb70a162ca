#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never shipped, never timed): gpurun_out/libhgn_mp_stamp.so
set -e
cd "$(dirname "$0")/../hyper-graph-nets_amd/csrc"
mkdir -p ../../tools/_build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DHGN_STAMP"
/opt/rocm/bin/hipcc $F -c mlp.hip -o /tmp/mlp_s.o
/opt/rocm/bin/hipcc $F -c segment.hip -o /tmp/seg_s.o
/opt/rocm/bin/hipcc $F -c wgrad.hip -o /tmp/wg_s.o
/opt/rocm/bin/hipcc $F -x hip -c host.cpp -o /tmp/host_s.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_build/libhgn_mp_stamp.so /tmp/mlp_s.o /tmp/seg_s.o /tmp/wg_s.o /tmp/host_s.o
