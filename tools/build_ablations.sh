#!/bin/bash
# Diagnostic builds of libhgn_mp.so with compile-time ablations of the split-bf16 edge kernels (csrc/mlp6_device.h: HGN_ABL):
#   tools/build_ablations.sh 1 2 4 8 ...   ->  hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_abl<N>.so   (HGN_LIB=<that file> selects it)
# Results are WRONG by construction; only the timing of tools/fusedbench.py means anything with them.
set -e
cd "$(dirname "$0")/../hyper-graph-nets_amd/csrc"
make -s
mkdir -p ../hgn_amd/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include -DHGN_ABL=$n $HGN_ABL_EXTRA -c mlp6.hip -o /tmp/mlp6_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../hgn_amd/abl/libhgn_mp_abl$n.so mlp.o /tmp/mlp6_abl$n.o fused_bwd.o segment.o wgrad.o features.o host.o
  echo built abl$n
done
