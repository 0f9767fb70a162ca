#!/bin/bash
# Laboratory build of the library: the product sources compiled with -DHGN_LAB=1 (kernel variants that were measured and NOT
# adopted -- 192-row / 128-row tilings everywhere, the previous weight-gradient kernel, compile-time ablation instantiations of
# the fused edge backward, LDS padding for occupancy experiments -- and their environment switches) plus the weight-stationary
# edge forward (tools/lab/ws_fwd.hip).  Output: tools/_build/libhgn_mp_lab.so; select it with HGN_LIB=<that file>.
# The shipped hyper-graph-nets_amd/hgn_amd/libhgn_mp.so contains none of this.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
SRC="$ROOT/hyper-graph-nets_amd/csrc"
OUT="$ROOT/tools/_build"
mkdir -p "$OUT/lab_obj"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I$ROOT/include -I$SRC -I$ROOT/tools/lab -DHGN_LAB=1 ${HGN_LAB_EXTRA}"
for f in mlp mlp6 segment wgrad features; do
  /opt/rocm/bin/hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/lab_obj/$f.o" &
done
# HGN_LAB_FUSED_V1=1: the first version of the fused edge backward (tools/lab/fused_bwd_v1.hip, with its HGN_FUSED_DBG ablation
# instantiations) in place of the product kernel
if [ -n "$HGN_LAB_FUSED_V1" ]; then FB="$ROOT/tools/lab/fused_bwd_v1.hip"; else FB="$SRC/fused_bwd.hip"; fi
/opt/rocm/bin/hipcc $FLAGS -c "$FB" -o "$OUT/lab_obj/fused_bwd.o" &
/opt/rocm/bin/hipcc $FLAGS -c "$ROOT/tools/lab/ws_fwd.hip" -o "$OUT/lab_obj/ws_fwd.o" &
/opt/rocm/bin/hipcc $FLAGS -x hip -c "$SRC/host.cpp" -o "$OUT/lab_obj/host.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libhgn_mp_lab.so" "$OUT"/lab_obj/*.o
echo "built $OUT/libhgn_mp_lab.so"
