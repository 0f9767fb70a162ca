#!/bin/bash
# Compile-time ablations of the fused edge backward (csrc/fused_bwd.hip, -DHGN_FEXP=<bits>: results are then WRONG, only the
# duration means something).  `build` (CPU, here): one laboratory library per bit set under tools/_build/abl/;
# `run` (GPU box): tools/fusedbench.py against each of them -> gpurun_out/<tag>/fused_bwd_ablation.log.
#   bash tools/fused_ablation.sh build ; gpurun -- 'bash tools/fused_ablation.sh run r3x'
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/hyper-graph-nets_amd/csrc"
OUT="$ROOT/tools/_build"
BITS="0 1 2 4 8 32 64 96 128 255 16 256 272"
NAMES=([0]="product schedule" [1]="no weight DMA" [2]="no z2/z1 fetch" [4]="no dz1/de stores" [8]="no chain row loads" [32]="no weight-gradient blocks"
       [64]="no chain product sweeps" [96]="no MFMA at all" [128]="no A publish" [255]="everything off"
       [16]="chain row loads as whole rows (same bytes, 8 lines per instruction instead of 64 quarter-lines)" [256]="stores as whole rows" [272]="both")
if [ "$1" = build ]; then
  bash "$ROOT/tools/lab/build_lab.sh"
  mkdir -p "$OUT/abl"
  FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I$ROOT/include -I$SRC -I$ROOT/tools/lab -DHGN_LAB=1"
  for b in $BITS; do
    ( /opt/rocm/bin/hipcc $FLAGS -DHGN_FEXP=$b -c "$SRC/fused_bwd.hip" -o "$OUT/abl/fused_$b.o" &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/abl/libhgn_fexp_$b.so" "$OUT/abl/fused_$b.o" \
        $(ls "$OUT"/lab_obj/*.o | grep -v fused_bwd.o) ) &
  done
  wait
  ls -la "$OUT/abl"/*.so
else
  TAG=${2:-abl}
  mkdir -p "$ROOT/gpurun_out/$TAG"
  LOG="$ROOT/gpurun_out/$TAG/fused_bwd_ablation.log"
  echo "# tools/fused_ablation.sh run: tools/fusedbench.py (128 flag_simple-shape graphs, 1 188 096 edge rows, 6 timed iterations) per -DHGN_FEXP build" > "$LOG"
  for b in $BITS; do
    echo "## HGN_FEXP=$b  ${NAMES[$b]}" >> "$LOG"
    HGN_LIB="$OUT/abl/libhgn_fexp_$b.so" timeout -k 10 120 python "$ROOT/tools/fusedbench.py" >> "$LOG" 2>&1 || echo "failed rc=$?" >> "$LOG"
  done
  echo "## shipped library" >> "$LOG"
  timeout -k 10 120 python "$ROOT/tools/fusedbench.py" >> "$LOG" 2>&1
  cat "$LOG"
fi
