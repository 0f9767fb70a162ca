#!/bin/bash
# Diagnostic builds of libhgn_mp.so with compile-time ablations of the split-product edge kernels:
#   tools/build_ablations.sh 1 2 4 8 ...        (csrc/mlp6_device.h: HGN_ABL -- forward kernels)
#   tools/build_ablations.sh f1 f2 f4 f8 f32 f64 ...  (csrc/fused_bwd3.hip: HGN_FEXP -- fused backward, product mode 3)
#   ->  hyper-graph-nets_amd/hgn_amd/abl/libhgn_mp_abl<N>.so   (HGN_LIB=<that file> selects it)
# Results are WRONG by construction; only the timing of tools/fusedbench.py means anything with them.
set -e
cd "$(dirname "$0")/../hyper-graph-nets_amd/csrc"
make -s
mkdir -p ../hgn_amd/abl
for n in "$@"; do
  if [[ $n == f* ]]; then
    v=${n#f}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include -DHGN_FEXP=$v $HGN_ABL_EXTRA -c fused_bwd3.hip -o /tmp/fused3_abl$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../hgn_amd/abl/libhgn_mp_abl$n.so mlp.o mlp6.o fused_bwd.o /tmp/fused3_abl$v.o segment.o wgrad.o features.o host.o
  else
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include -DHGN_ABL=$n $HGN_ABL_EXTRA -c mlp6.hip -o /tmp/mlp6_abl$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../hgn_amd/abl/libhgn_mp_abl$n.so mlp.o /tmp/mlp6_abl$n.o fused_bwd.o fused_bwd3.o segment.o wgrad.o features.o host.o
  fi
  echo built abl$n
done
